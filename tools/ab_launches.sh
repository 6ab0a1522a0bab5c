#!/bin/bash
# Per-launch times (tools/per_launch.py) of one forward for several library builds, interleaved on one box.
# usage: LIBS="release polyphase" KIND=upsample OUT=gpurun_out/x tools/ab_launches.sh "1 1000" "1 100" ...
OUT=${OUT:-gpurun_out/ab}; mkdir -p $OUT
for shape in "$@"; do
  for V in ${LIBS:-release}; do
    if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
    f=$OUT/launches_${V}_$(echo $shape | tr ' ' x).txt
    timeout -k 10 180 python tools/per_launch.py $shape 2>/dev/null > $f
    echo "B x T = $shape  $V: $(awk -v kind=${KIND:-upsample} '{tot+=$7} $2 ~ kind {printf "%s ", $7; s+=$7} END {printf "| sum %.1f us of all launches %.0f us", s, tot}' $f)"
  done
done
