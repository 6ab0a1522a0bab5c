#!/bin/bash
# diagnostics: per-stage MRF TFLOP/s for kernel variants / ablations (env-var switches of the library).
# usage: tools/ablate.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each argument = one configuration)
# The switches exist only in the diagnostic build of the library (make -C iris-tts_amd/csrc diag): the release
# library reads no environment variable.
export IRIS_HIFIGAN_LIB=${IRIS_HIFIGAN_LIB:-$PWD/iris-tts_amd/csrc/libiris_hifigan_diag.so}
[ -f "$IRIS_HIFIGAN_LIB" ] || make -C iris-tts_amd/csrc diag > /dev/null
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 5 $BENCH_ARGS > gpurun_out/ab_$i.json 2>gpurun_out/ab_err.txt || tail -5 gpurun_out/ab_err.txt
  python - "$cfg" gpurun_out/ab_$i.json <<'PY'
import json, sys
d=json.load(open(sys.argv[2]))
print("%-44s"%sys.argv[1], "ms/step %.3f"%d["ms_per_step"], {k[4:]:round(v["tflops"],1) for k,v in d["kernels"].items() if k.startswith("mrf_stage")}, "ups %.1f"%d["kernels"]["upsample"]["tflops"])
PY
done
