#!/bin/bash
# diagnostics: per-stage MRF time of one forward (sum of the stage's launches) for several builds of the library
# usage: tools/lib_ab.sh "B T" "B T" ... ; LIBS="nopair pt0 ..." names libiris_hifigan_<name>.so under csrc/ ("" = the release library)
LIBS=${LIBS:-"release"}
for shape in "$@"; do
for V in $LIBS; do
  if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
  echo "B x T = $shape  $V: $(python tools/per_launch.py $shape | awk '{tot+=$7} /mrf/ {s[$4]+=$7} END {printf "all launches %.0f us | MRF stage 0 %.0f  1 %.0f  2 %.0f  3 %.0f", tot, s[0], s[1], s[2], s[3]}')"
done; done
