#!/bin/bash
O=gpurun_out/r03g; mkdir -p $O
for T in 100 282 700 1000; do for G in "" "--graph"; do
  echo "T=$T $G $(timeout -k 10 120 python bench.py --frames $T --steps 100 --warmup 20 --no-cpu-baseline --no-extras $G 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4))')" >> $O/graph.txt
done; done
cat $O/graph.txt
