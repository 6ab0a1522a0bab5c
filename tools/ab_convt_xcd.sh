#!/bin/bash
# Round 4: XCD-ordered jobs of the ConvTranspose GEMM kernels (the column blocks of a row tile on one XCD), fp32 (release
# against libiris_hifigan_noxcd.so = -DIRIS_CONVT_XCD_ORDER=0) and bf16 (against the previous build, XCD order always on).
set -e
OUT=gpurun_out/r04_convt_xcd; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -m gpu -x -q -k "transpose" > $OUT/pytest_convt.txt 2>&1 || { tail -40 $OUT/pytest_convt.txt; exit 1; }
tail -2 $OUT/pytest_convt.txt
timeout -k 10 600 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_noxcd.so f32 > $OUT/bitwise_f32_vs_noxcd.txt 2>&1 || { tail -5 $OUT/bitwise_f32_vs_noxcd.txt; exit 1; }
tail -1 $OUT/bitwise_f32_vs_noxcd.txt
timeout -k 10 600 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_prev.so bf16 > $OUT/bitwise_bf16_vs_prev.txt 2>&1 || { tail -5 $OUT/bitwise_bf16_vs_prev.txt; exit 1; }
tail -1 $OUT/bitwise_bf16_vs_prev.txt
LIBS="release noxcd" KIND=upsample OUT=$OUT/f32 tools/ab_launches.sh "1 1000" "1 500" "1 100" "8 300" "32 500" "1 1000" "1 500" | tee $OUT/upsample_launch_times_f32.txt
export DTYPE=bf16
LIBS="release prev" KIND=upsample OUT=$OUT/bf16 tools/ab_launches.sh "1 1000" "1 100" "3 130" "8 300" "32 500" "1 1000" "1 100" | tee $OUT/upsample_launch_times_bf16.txt
