"""Diagnostics: median time of the wide MRF launches (stages 0 and 1; steps 0-4 = three-output steps, step 5 = the
summing step) of one forward, for a list of B x T shapes, in whatever library IRIS_HIFIGAN_LIB names.  Used to calibrate
the launch-plan estimates of mrf_conv_mfma_f32.h against builds that force one plan (make relvariant
EXTRA=-DIRIS_MRF_FORCE_PLAN=n).  usage: plan_sweep.py B:T B:T ..."""
import json, os, sys, torch
sys.path.insert(0, "iris-tts_amd")
from iris._engine import GeneratorEngine
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
cfg = GeneratorConfig(); eng = GeneratorEngine(cfg, seeded_state_dict(cfg), torch.device("cuda", 0))
REP = 7
for shape in sys.argv[1:]:
    B, T = (int(v) for v in shape.split(":"))
    mel = torch.from_numpy(seeded_mel(1, B, T)).cuda()
    eng.set_profiling(0)
    for _ in range(3): eng.forward(mel)
    torch.cuda.synchronize(); eng.set_profiling(1)
    for _ in range(REP): eng.forward(mel)
    torch.cuda.synchronize(); recs = eng.read_profile(); n = len(recs) // REP
    out = {"lib": os.path.basename(os.environ.get("IRIS_HIFIGAN_LIB", "release")), "B": B, "T": T}
    total = 0.0
    for i in range(n):
        ms = sorted(recs[i + k * n]["ms"] for k in range(REP))[REP // 2]; r = recs[i]
        total += ms
        if r["kind"].startswith("mrf") and r["stage"] in (0, 1):
            key = f"s{r['stage']}_{'sum' if r['step'] == 5 else 'step'}"
            out.setdefault(key, []).append(round(ms * 1e3, 1))
    for k in list(out):
        if k.endswith("_step"): out[k] = round(sorted(out[k])[len(out[k]) // 2], 1)
        elif k.endswith("_sum"): out[k] = out[k][0]
    out["all_us"] = round(total * 1e3, 1)
    print(json.dumps(out), flush=True)
