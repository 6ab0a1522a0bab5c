#!/bin/bash
# round 3, call j: per-CU timeline of the persistent fp32 MRF launches (block start / end stamps, blocks per CU)
set -o pipefail
O=gpurun_out/r03j
mkdir -p $O
export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_blocklog.so IRIS_HIFIGAN_BLOCKLOG=1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-profile --steps 2 --warmup 2 > $O/bench_1x1000.json 2> $O/blocklog_1x1000.txt; echo "rc=$?" | tee -a $O/summary.txt
grep blocklog $O/blocklog_1x1000.txt | tail -16
timeout -k 10 300 python bench.py --batch 32 --frames 500 --no-cpu-baseline --no-extras --no-profile --steps 1 --warmup 1 > $O/bench_32x500.json 2> $O/blocklog_32x500.txt; echo "rc=$?" | tee -a $O/summary.txt
grep blocklog $O/blocklog_32x500.txt | tail -16
