#!/bin/bash
# round 3, call c: persistent fp32 conv pairs (mrf_pair_f32_pf.h) -- op-level parity, whole-waveform bit identity against the
# build without them, per-stage timing A/B
set -o pipefail
O=gpurun_out/r03c
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.txt 2>&1; echo "pytest parity rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_parity.txt
timeout -k 10 600 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_nopf32.so > $O/bitwise_sweep.txt 2>&1; echo "bitwise sweep rc=$?" | tee -a $O/summary.txt
tail -2 $O/bitwise_sweep.txt
LIBS="release nopf32 release nopf32" timeout -k 10 600 bash tools/lib_ab.sh "1 1000" "1 700" "1 500" "1 282" "1 100" "32 500" 2>&1 | tee $O/lib_ab.txt
timeout -k 10 600 bash tools/ablate.sh "X=0" "IRIS_HIFIGAN_PAIR_SUM=0" "IRIS_HIFIGAN_PAIR_SUM=2" "IRIS_HIFIGAN_PAIR_PF_MODE=0" "X=0" 2>&1 | tee $O/ablate_1x1000.txt
BENCH_ARGS="--batch 32 --frames 500" timeout -k 10 600 bash tools/ablate.sh "X=0" "IRIS_HIFIGAN_PAIR_SUM=0" "IRIS_HIFIGAN_PAIR_SUM=2" "IRIS_HIFIGAN_PAIR_PF_MODE=0" 2>&1 | tee $O/ablate_32x500.txt
