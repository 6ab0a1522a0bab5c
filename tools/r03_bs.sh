#!/bin/bash
# round 3: bf16 summing pair -- parity, bitwise against the three-tensor build, configs[2] A/B
O=gpurun_out/r03bs
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu > $O/pytest_bf16.txt 2>&1; echo "pytest rc=$?" > $O/summary.txt
tail -5 $O/pytest_bf16.txt
grep -q "rc=0" $O/summary.txt || exit 1
timeout -k 10 600 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_nosum.so bf16 > $O/bitwise_bf16.txt 2>&1; echo "bitwise rc=$?" >> $O/summary.txt
tail -2 $O/bitwise_bf16.txt
for V in release nosum release nosum; do
  if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
  for S in "32 500" "1 1000" "8 1000"; do
    set -- $S
    timeout -k 10 200 python bench.py --dtype bf16 --batch $1 --frames $2 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c '
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get("kernels",{})
print(sys.argv[1], sys.argv[2], "ms", round(d["ms_per_step"],3), {n: round(v["ms_per_step"],3) for n,v in k.items()})' $V "$S" >> $O/bench_ab.txt
  done
done
cat $O/bench_ab.txt
