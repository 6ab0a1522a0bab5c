#!/bin/bash
O=gpurun_out/r03bs2
mkdir -p $O
for V in release sumA nosum release sumA nosum; do
  if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
  for S in "32 500" "8 1000"; do
    set -- $S
    timeout -k 10 200 python bench.py --dtype bf16 --batch $1 --frames $2 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c '
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get("kernels",{})
print(sys.argv[1], sys.argv[2], "ms", round(d["ms_per_step"],3), {n: round(v["ms_per_step"],3) for n,v in k.items() if n in ("upsample","conv_post","mrf_stage2_C64","mrf_stage3_C32")})' $V "$S" >> $O/bench_ab.txt
  done
done
cat $O/bench_ab.txt
