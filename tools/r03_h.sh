#!/bin/bash
# round 3, call h: two LDS window buffers in the wide (C >= 128) persistent fp32 MRF kernel -- parity, bit identity, timing
set -o pipefail
O=gpurun_out/r03h
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.txt 2>&1; echo "pytest parity rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_parity.txt
timeout -k 10 600 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_nodbuf.so > $O/bitwise_sweep.txt 2>&1; echo "bitwise sweep rc=$?" | tee -a $O/summary.txt
tail -2 $O/bitwise_sweep.txt
LIBS="release nodbuf release nodbuf" timeout -k 10 900 bash tools/lib_ab.sh "1 1000" "1 700" "1 500" "1 282" "1 100" "32 500" 2>&1 | grep -v amdgpu.ids | tee $O/lib_ab.txt
