#!/bin/bash
# experiment: the wide fp32 MRF kernel as 128 x 32 blocks (four waves share one weight stream) and / or three blocks per CU, half-height tiles
O=gpurun_out/r03tn; mkdir -p $O
IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_tall3.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward or bitwise or config" > $O/pytest_tall3.txt 2>&1; tail -2 $O/pytest_tall3.txt
for V in release plan1 wide3 tall2 tall3; do
  if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
  timeout -k 10 300 python tools/plan_sweep.py 1:1000 4:1000 8:1000 32:500 1:500 > $O/$V.jsonl 2>$O/$V.err || exit 1
  echo "$V done"
done
