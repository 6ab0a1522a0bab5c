#!/bin/bash
# Round-end measurement pass on the GPU box (through gpurun): bench lines, rocprofv3 kernel stats and HBM
# traffic for the fp32 headline (configs[1]) and the bf16 variant (configs[2]).  Outputs under gpurun_out/<tag>_*.
set -e
REPO=$PWD; TAG=${1:-rXX}
export TMPDIR=/tmp
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --dtype bf16 --batch 32 --frames 500 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_bf16_c3.json
python bench.py --batch 32 --frames 500 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_f32_c3.json
python bench.py --dtype bf16 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_bf16_c2.json
for shape in "1 64" "1 100" "1 200" "1 282" "1 500" "1 700" "1 3000" "8 300"; do set -- $shape; python bench.py --batch $1 --frames $2 --no-cpu-baseline --no-extras; done > gpurun_out/${TAG}_bench_other_shapes.jsonl
python tools/stream_timing.py 1024 2>/dev/null > gpurun_out/${TAG}_stream_timing.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/${TAG}_prof_f32 --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/${TAG}_bench_under_rocprof.json 2> $REPO/gpurun_out/${TAG}_prof_f32.log
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/${TAG}_prof_bf16 --output-format csv -- python3 $REPO/bench.py --dtype bf16 --batch 32 --frames 500 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/${TAG}_bench_bf16_c3_under_rocprof.json 2> $REPO/gpurun_out/${TAG}_prof_bf16.log
cd $REPO
cp $(find gpurun_out/${TAG}_prof_f32 -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
cp $(find gpurun_out/${TAG}_prof_bf16 -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats_bf16_c3.csv
tools/hbm_traffic.sh ${TAG}
PLAN=PUMMMMMMUMMMUMMMUMMMO tools/hbm_traffic.sh ${TAG}_bf16_c3 --dtype bf16 --batch 32 --frames 500    # (the C = 256 MRF steps run on the generic kernel: not recognisable by name)
ls gpurun_out/${TAG}_*.json gpurun_out/${TAG}_*.csv
