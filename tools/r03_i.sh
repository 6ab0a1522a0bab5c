#!/bin/bash
# round 3, call i: does drawing tiles from a counter (finer, dynamic units) shorten the tail of the wide MRF launches? batch 1 x 2000 frames
set -o pipefail
O=gpurun_out/r03i
mkdir -p $O
BENCH_ARGS="--batch 1 --frames 2000" timeout -k 10 600 bash tools/ablate.sh "X=0" "IRIS_HIFIGAN_DYNTILES=0" "IRIS_HIFIGAN_MRFPLAN=1" "IRIS_HIFIGAN_MRFPLAN=1 IRIS_HIFIGAN_DYNTILES=0" "IRIS_HIFIGAN_MRFPLAN=2" "X=0" "IRIS_HIFIGAN_DYNTILES=0" 2>&1 | grep -v amdgpu.ids | tee $O/ablate_dyn_1x2000.txt
BENCH_ARGS="--batch 4 --frames 1000" timeout -k 10 600 bash tools/ablate.sh "X=0" "IRIS_HIFIGAN_DYNTILES=0" "IRIS_HIFIGAN_MRFPLAN=1" "X=0" 2>&1 | grep -v amdgpu.ids | tee $O/ablate_dyn_4x1000.txt
