#!/bin/bash
# round 3, call e: C = 128 fused pair on short inputs (32-row tiles), summing pair from four rounds on -- parity + short-input timing
set -o pipefail
O=gpurun_out/r03e
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.txt 2>&1; echo "pytest parity rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_parity.txt
for shape in "1 --frames 64" "1 --frames 100" "1 --frames 200" "1 --frames 282" "1 --frames 400" "2 --frames 100"; do
  echo "== batch $shape" | tee -a $O/ablate_short.txt
  BENCH_ARGS="--batch $shape" timeout -k 10 300 bash tools/ablate.sh "IRIS_HIFIGAN_PAIR128_MAX=0" "X=0" "IRIS_HIFIGAN_PAIR128_MAX=0" "X=0" 2>&1 | grep -v amdgpu.ids | tee -a $O/ablate_short.txt
done
