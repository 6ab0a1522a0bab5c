#!/bin/bash
# Round 4, bf16 generic conv kernel: XCD grouping with the batch index folded into the item index (release) against the
# per-batch-item grouping (libiris_hifigan_nofold.so = make relvariant NAME=nofold EXTRA="-DIRIS_B16_XCD_FOLD=0").
# bf16 GPU tests, bitwise sweep, interleaved timings at configs[2] and 1 x 1000, HBM traffic of configs[2].  Through gpurun.
set -e
OUT=gpurun_out/r04_xcdfold; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q > $OUT/pytest_bf16.txt 2>&1 || { tail -40 $OUT/pytest_bf16.txt; exit 1; }
tail -2 $OUT/pytest_bf16.txt
timeout -k 10 600 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_nofold.so bf16 > $OUT/bitwise_vs_nofold.txt 2>&1 || { tail -5 $OUT/bitwise_vs_nofold.txt; exit 1; }
tail -1 $OUT/bitwise_vs_nofold.txt
for shape in "32 500" "1 1000" "8 300"; do set -- $shape
for r in 1 2 3; do
  for V in release nofold; do
    if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
    timeout -k 10 300 python bench.py --dtype bf16 --batch $1 --frames $2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$1x$2', '$V', round(d['ms_per_step'],3), {s: round(k[s]['ms_per_step'],3) for s in ('mrf_stage0_C256','mrf_stage1_C128','mrf_stage2_C64','mrf_stage3_C32','upsample')})"
  done
done
done | tee $OUT/timings.txt
unset IRIS_HIFIGAN_LIB
PLAN=PUMMMMMMUMMMUMMMUMMMO tools/hbm_traffic.sh r04_xcdfold/bf16_c3 --dtype bf16 --batch 32 --frames 500
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04_xcdfold/bf16_c3_hbm_traffic.json'))
print('mrf_traffic_bytes_per_launch', d['mrf_traffic_bytes_per_launch'])
for e in d['per_dispatch']: print(e['kernel'][:60], e['grid'], round(2*e['FETCH_SIZE_KB']/1024), round(e['WRITE_SIZE_KB']/1024))
PY
