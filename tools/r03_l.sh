#!/bin/bash
# round 3, call l: the generic conv kernel after removing the chunk prefetch; last upsampler at MT = 1 vs MT = 2 (release-flag builds)
O=gpurun_out/r03l; mkdir -p $O
for i in 1 2; do
for V in release upsmt2; do
  if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
  echo "== $V" | tee -a $O/upsamplers.txt
  python tools/per_launch.py 1 1000 2>/dev/null | grep -E "upsample|conv_pre|conv_post" | tee -a $O/upsamplers.txt
done; done
unset IRIS_HIFIGAN_LIB
python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})" | tee -a $O/upsamplers.txt
