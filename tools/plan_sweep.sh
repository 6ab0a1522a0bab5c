#!/bin/bash
# diagnostics: per-stage MRF time of one forward (and the launches it took: fused conv pairs count once) under each forced launch plan (IRIS_HIFIGAN_MRFPLAN of the diagnostic build)
# NOTE: the diagnostic build's persistent kernel carries runtime ablation switches and runs ~10 % slower than the release
# build's; the small-problem kernel (plan 4) does not -- compare plan 4 with the others in RELEASE builds only.
# usage: tools/plan_sweep.sh "B T" "B T" ...     (run through gpurun from the repo root)
export IRIS_HIFIGAN_LIB=${IRIS_HIFIGAN_LIB:-$PWD/iris-tts_amd/csrc/libiris_hifigan_diag.so}
PLANS=${PLANS:--1 0 1 2 4}
for shape in "$@"; do
for P in $PLANS; do
  echo "== B x T = $shape plan=$P"
  IRIS_HIFIGAN_MRFPLAN=$P python tools/per_launch.py $shape | awk '/mrf/ {s[$4]+=$7; n[$4]++} END {for (k in s) printf "  stage %s: %.1f us in %d launches\n", k, s[k], n[k]}' | sort
done; done
