#!/bin/bash
# round 3: persistent + prefetching bf16 conv pair (mrf_pair_bf16_pf.h) -- correctness, then A/B against the non-persistent kernel
set -o pipefail
O=gpurun_out/r03a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q > $O/pytest_bf16.txt 2>&1; echo "pytest bf16 rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_bf16.txt
NOPF=$PWD/iris-tts_amd/csrc/libiris_hifigan_nopf.so
REL=$PWD/iris-tts_amd/csrc/libiris_hifigan.so
timeout -k 10 900 python tools/b16_sweep.py "IRIS_HIFIGAN_LIB=$NOPF" "IRIS_HIFIGAN_LIB=$REL" "IRIS_HIFIGAN_LIB=$NOPF" "IRIS_HIFIGAN_LIB=$REL" \
   "-" "IRIS_B16_PAIR_PF=0" "IRIS_B16_PAIR_PF32=2" "IRIS_B16_PAIR_PF64=2" "IRIS_B16_PAIR_PF64=3" "IRIS_B16_PAIR_PF64=4" "-" 2>&1 | tee $O/sweep.txt
