#!/bin/bash
# One GPU call, several steps: TAG=<dir under gpurun_out> STEPS="tests bench launches" tools/gpu_steps.sh   (through gpurun, from the repo root)
set -e
OUT=gpurun_out/${TAG:-r04x}; mkdir -p $OUT
for step in ${STEPS:-tests bench launches}; do
  case $step in
    tests)    timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }; tail -3 $OUT/pytest.txt ;;
    bench)    timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; } ;;
    launches) for t in 100 282 500 1000; do timeout -k 10 120 python tools/per_launch.py 1 $t > $OUT/launches_1x$t.txt 2>&1; done ;;
  esac
done
