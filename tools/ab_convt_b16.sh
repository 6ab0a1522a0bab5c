#!/bin/bash
# Round 4, bf16 upsamplers: ConvTranspose1d as one GEMM launch (release, convt_mfma_bf16.h) against the polyphase launches
# (libiris_hifigan_polyups.so = -DIRIS_CONVT_GEMM_B16_DEFAULT=0) and polyphase tile-height variants of the last upsampler.
# bf16 GPU tests, bitwise sweep, per-launch upsample times, whole-forward timings.  Through gpurun, from the repo root.
set -e
OUT=gpurun_out/${TAG:-r04_convt_b16}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q > $OUT/pytest_bf16.txt 2>&1 || { tail -40 $OUT/pytest_bf16.txt; exit 1; }
tail -2 $OUT/pytest_bf16.txt
timeout -k 10 600 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_polyups.so bf16 ${NRANDOM:-40} > $OUT/bitwise_vs_polyphase.txt 2>&1 || { tail -5 $OUT/bitwise_vs_polyphase.txt; exit 1; }
tail -1 $OUT/bitwise_vs_polyphase.txt
export DTYPE=bf16
LIBS="${LIBS:-release polyups upsmt1 upsmt2}" KIND=upsample OUT=$OUT tools/ab_launches.sh "32 500" "1 1000" "8 300" "1 100" "32 500" | tee $OUT/upsample_launch_times.txt
for shape in "32 500" "1 1000"; do set -- $shape
for r in 1 2 3; do
  for V in release polyups; do
    if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
    timeout -k 10 300 python bench.py --dtype bf16 --batch $1 --frames $2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$1x$2', '$V', round(d['ms_per_step'],3), {s: round(k[s]['ms_per_step'],3) for s in ('mrf_stage0_C256','mrf_stage1_C128','mrf_stage2_C64','mrf_stage3_C32','upsample')})"
  done
done
done | tee $OUT/timings.txt
