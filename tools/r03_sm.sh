#!/bin/bash
# small-problem kernel: weight ring carried across chunk boundaries -- parity, bitwise against the previous build, timing at short shapes
O=gpurun_out/r03sm; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mrf_step or small or short or forward" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout -k 10 400 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_head.so > $O/bitwise.txt 2>&1; tail -1 $O/bitwise.txt
for V in release head release head; do
  if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
  timeout -k 10 300 python tools/plan_sweep.py 1:40 1:64 1:100 1:120 1:150 2:64 3:50 >> $O/$V.jsonl 2>$O/$V.err || exit 1
  for T in 64 100 150; do echo "$V T=$T $(timeout -k 10 120 python bench.py --frames $T --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4))')" >> $O/bench.txt; done
done
cat $O/bench.txt
