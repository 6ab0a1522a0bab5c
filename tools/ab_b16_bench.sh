set -e
timeout -k 10 300 python tools/bitwise_sweep.py iris-tts_amd/csrc/libiris_hifigan_researly.so bf16 > gpurun_out/researly_bitwise.txt 2>&1 || true
tail -1 gpurun_out/researly_bitwise.txt
for r in 1 2 3; do
  for V in release researly; do
    if [ "$V" = release ]; then unset IRIS_HIFIGAN_LIB; else export IRIS_HIFIGAN_LIB=$PWD/iris-tts_amd/csrc/libiris_hifigan_$V.so; fi
    timeout -k 10 300 python bench.py --dtype bf16 --batch 32 --frames 500 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$V', round(d['ms_per_step'],3), {s: round(k[s]['ms_per_step'],3) for s in ('mrf_stage0_C256','mrf_stage1_C128','mrf_stage2_C64','mrf_stage3_C32','upsample')})"
  done
done
