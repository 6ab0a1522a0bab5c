#!/bin/bash
# snake-ordered jobs also on the C <= 64 launches of the wide kernel? whole-forward launch sums, release against -DIRIS_MRF_ZDYN_MIN_CHUNKS=1
O=gpurun_out/r03za; mkdir -p $O
IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_zall.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mrf_step or mrf_job" > $O/pytest_zall.txt 2>&1; tail -2 $O/pytest_zall.txt
timeout -k 10 400 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_zall.so > $O/bitwise.txt 2>&1; tail -1 $O/bitwise.txt
VARIANTS="release zall" OUT=$O bash tools/r03_ps.sh
