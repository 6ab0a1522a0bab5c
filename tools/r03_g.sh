#!/bin/bash
# round 3, call g: start-up stagger of the persistent fp32 MRF kernel's second residency generation; last upsampler at four blocks per CU
set -o pipefail
O=gpurun_out/r03g
mkdir -p $O
timeout -k 10 600 bash tools/ablate.sh "X=0" "IRIS_HIFIGAN_STAGGER=2" "IRIS_HIFIGAN_STAGGER=4" "IRIS_HIFIGAN_STAGGER=8" "IRIS_HIFIGAN_STAGGER=16" "IRIS_HIFIGAN_STAGGER=32" "X=0" "IRIS_HIFIGAN_CONV_MT=2" 2>&1 | grep -v amdgpu.ids | tee $O/ablate_stagger_1x1000.txt
BENCH_ARGS="--batch 32 --frames 500" timeout -k 10 600 bash tools/ablate.sh "X=0" "IRIS_HIFIGAN_STAGGER=4" "IRIS_HIFIGAN_STAGGER=16" "X=0" 2>&1 | grep -v amdgpu.ids | tee $O/ablate_stagger_32x500.txt
