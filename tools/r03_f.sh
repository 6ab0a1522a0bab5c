#!/bin/bash
# round 3, call f: chunk prefetch in the generic fp32 conv kernel (upsamplers, PostNet), sub-batch passes, summing pair from four
# rounds on -- full GPU suite, then A/B of the prefetch
set -o pipefail
O=gpurun_out/r03f
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_gpu.txt
for shape in "1 --frames 1000" "1 --frames 100" "32 --frames 500"; do
  echo "== batch $shape" | tee -a $O/ablate_conv_prefetch.txt
  BENCH_ARGS="--batch $shape" timeout -k 10 400 bash tools/ablate.sh "IRIS_HIFIGAN_ABLATE=64" "X=0" "IRIS_HIFIGAN_ABLATE=64" "X=0" 2>&1 | grep -v amdgpu.ids | tee -a $O/ablate_conv_prefetch.txt
done
timeout -k 10 300 python tools/per_launch.py 1 1000 2>/dev/null | grep -v amdgpu > $O/per_launch_1x1000.txt
timeout -k 10 300 python tools/stream_timing.py 1024 2>/dev/null | grep -v amdgpu > $O/stream_timing.txt
