export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_diag.so
for shape in "1 282" "1 400" "1 600" "1 1000" "4 500"; do
for cfg in "IRIS_HIFIGAN_PAIR=0" "IRIS_HIFIGAN_PAIR_FORCE=1" "IRIS_HIFIGAN_PAIR_FORCE=1 IRIS_HIFIGAN_PAIR_ZMAJOR=1"; do
  echo "B x T = $shape  $cfg: $(env $cfg python tools/per_launch.py $shape | awk '{tot+=$7} /mrf/ {s[$4]+=$7} END {printf "all %.0f us | MRF stage 2 %.0f  3 %.0f", tot, s[2], s[3]}')"
done; done
