"""Runs bench.py --dtype bf16 under a list of environment settings and prints per-stage times (diagnostic).
usage: python tools/b16_sweep.py "IRIS_B16_TILE32=1" "IRIS_B16_TILE32=2 IRIS_B16_TILE64=1" ...   ("-" = no override)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the switches exist only in the diagnostic build (make -C iris-tts_amd/csrc diag); the release library reads none
os.environ.setdefault("IRIS_HIFIGAN_LIB", os.path.join(ROOT, "iris-tts_amd", "csrc", "libiris_hifigan_diag.so"))
shape = os.environ.get("SWEEP_SHAPE", "32x500").split("x")
for spec in sys.argv[1:]:
    env = dict(os.environ)
    if spec != "-":
        for kv in spec.split():
            k, v = kv.split("=")
            env[k] = v
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dtype", "bf16", "--batch", shape[0], "--frames", shape[1],
                          "--no-cpu-baseline", "--no-extras", "--steps", "10", "--warmup", "2"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print(spec, "FAILED", out.stderr[-400:], flush=True)
        continue
    k = d["kernels"]
    print("%-44s step %.2f ms | up %.2f | s0 %.2f s1 %.2f s2 %.2f s3 %.2f | pre %.3f post %.3f" % (
        spec, d["ms_per_step"], k["upsample"]["ms_per_step"], k["mrf_stage0_C256"]["ms_per_step"], k["mrf_stage1_C128"]["ms_per_step"],
        k["mrf_stage2_C64"]["ms_per_step"], k["mrf_stage3_C32"]["ms_per_step"], k["conv_pre"]["ms_per_step"], k["conv_post"]["ms_per_step"]), flush=True)
