#!/bin/bash
# Release-vs-release A/B on ONE box, interleaved: per-shape median step time of the wide fp32 MRF launches (tools/plan_sweep.py)
# for each library build.  usage: LIBS="release notall" SHAPES="1:1000 32:500" ROUNDS=2 OUT=gpurun_out/x tools/ab_sweep.sh
# ("release" = csrc/libiris_hifigan.so, any other name = csrc/libiris_hifigan_<name>.so from `make relvariant NAME=<name>`)
set -e
OUT=${OUT:-gpurun_out/ab}; mkdir -p $OUT
for r in $(seq 1 ${ROUNDS:-2}); do
  for V in ${LIBS:-release}; do
    if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
    timeout -k 10 300 python tools/plan_sweep.py ${SHAPES:-1:1000} 2>/dev/null | sed "s/^{/{\"round\": $r, \"build\": \"$V\", /" >> $OUT/sweep.jsonl
  done
done
python3 - $OUT/sweep.jsonl <<'PY'
import json,sys,collections
rows=[json.loads(l) for l in open(sys.argv[1])]
by=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    for k in ("s0_step","s0_sum","s1_step","s1_sum","all_us"):
        if k in r: by[(r["B"],r["T"],k)][r["build"]].append(r[k])
builds=sorted({r["build"] for r in rows})
print("shape key " + " ".join(f"{b:>12s}" for b in builds))
for (B,T,k),d in sorted(by.items()):
    print(f"{B}x{T} {k:8s} " + " ".join(f"{min(d[b]):12.1f}" if b in d else " "*12 for b in builds))
PY
