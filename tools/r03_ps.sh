#!/bin/bash
mkdir -p ${OUT:-gpurun_out/r03ps}
O=${OUT:-gpurun_out/r03ps}
SH=""
for T in 100 150 200 250 282 300 350 400 450 500 550 600 650 700 750 800 850 900 950 1000 1100 1200 1400 1600 2000; do SH="$SH 1:$T"; done
SH="$SH 2:282 2:500 2:700 3:500 4:400 4:1000 8:500 32:500"
for V in ${VARIANTS:-plan0 plan1 plan2 plan5 plan6 nozdyn release}; do
  if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
  timeout -k 10 400 python tools/plan_sweep.py $SH > $O/$V.jsonl 2>$O/$V.err || exit 1
  echo "$V done"
done
