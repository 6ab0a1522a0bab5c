#!/bin/bash
# round 3, call b: full GPU suite on the reworked ABI (lazy packings, ABI guards, describe_plan), bench line with the new
# sub-records, the one-rank RCCL path, and the bf16 persistent-pair A/B at equal occupancy
set -o pipefail
O=gpurun_out/r03b
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_gpu.txt
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --force-dist --no-extras --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo "force-dist rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --force-dist --include-h2d --batch 32 --frames 1000 --steps 5 --no-extras --no-cpu-baseline > $O/bench_force_dist_h2d.json 2>> $O/bench_force_dist.err; echo "force-dist h2d rc=$?" | tee -a $O/summary.txt
timeout -k 10 900 python tools/b16_sweep.py "IRIS_B16_PAIR_PF=0" "IRIS_B16_PAIR_PF=1" "IRIS_B16_PAIR_PF=0 IRIS_B16_PAIR32=1 IRIS_B16_PAIR64=3" \
   "IRIS_B16_PAIR_PF=1 IRIS_B16_PAIR_PF64=2" "IRIS_B16_PAIR_PF=0" 2>&1 | tee $O/sweep_equal_occupancy.txt
IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_diag.so IRIS_B16_PAIR_PF=1 timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py -x -q > $O/pytest_bf16_pf.txt 2>&1; echo "pytest bf16 with the persistent pair kernel rc=$?" | tee -a $O/summary.txt
tail -2 $O/pytest_bf16_pf.txt
