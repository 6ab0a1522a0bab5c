#!/bin/bash
# round 3, call m: does the wide persistent fp32 kernel gain from a third block per CU at EQUAL tile height? (half-height tiles, 2 vs 3 blocks)
O=gpurun_out/r03m; mkdir -p $O
export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_mw3.so
bash tools/ablate.sh "IRIS_HIFIGAN_MRFPLAN=1 IRIS_HIFIGAN_PERCU=2" "IRIS_HIFIGAN_MRFPLAN=1 IRIS_HIFIGAN_PERCU=3" "IRIS_HIFIGAN_MRFPLAN=1 IRIS_HIFIGAN_PERCU=2" "IRIS_HIFIGAN_MRFPLAN=1 IRIS_HIFIGAN_PERCU=3" 2>&1 | grep -v amdgpu.ids | tee $O/ablate_1x1000.txt
BENCH_ARGS="--batch 8 --frames 1000" bash tools/ablate.sh "IRIS_HIFIGAN_MRFPLAN=1 IRIS_HIFIGAN_PERCU=2" "IRIS_HIFIGAN_MRFPLAN=1 IRIS_HIFIGAN_PERCU=3" 2>&1 | grep -v amdgpu.ids | tee $O/ablate_8x1000.txt
