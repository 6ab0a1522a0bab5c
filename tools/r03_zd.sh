#!/bin/bash
# round 3: snake-ordered (tile, branch) jobs in the wide fp32 MRF kernel + the pair-model planner -- parity, bitwise sweep
# against the library built from the previous commit, plan sweep, bench
O=gpurun_out/r03zd2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" > $O/summary.txt
tail -3 $O/pytest_gpu.txt
grep -q "rc=0" $O/summary.txt || exit 1
timeout -k 10 600 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_head.so > $O/bitwise.txt 2>&1; echo "bitwise rc=$?" >> $O/summary.txt
tail -2 $O/bitwise.txt
VARIANTS="release head" OUT=$O bash tools/r03_ps.sh
for V in release head; do
  if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
  for T in 100 282 500 700 1000 100 282 500 700 1000; do
    echo "$V T=$T $(timeout -k 10 120 python bench.py --frames $T --steps 50 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')" >> $O/bench_ab.txt
  done
done
cat $O/bench_ab.txt
