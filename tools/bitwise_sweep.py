"""Compares the waveforms of two builds of the library bitwise over 63 (batch, frames) shapes (diagnostics: every launch
plan and fused kernel must produce the same bits).  usage: python tools/bitwise_sweep.py <other libiris_hifigan_*.so> [f32|bf16] [n_random_shapes]"""
import os, subprocess, sys, json
import numpy as np
shapes = [(1, t) for t in (1, 2, 3, 5, 7, 13, 31, 40, 50, 63, 64, 65, 77, 99, 100, 117, 118, 119, 127, 128, 129, 150, 199, 230, 257, 282, 301, 333, 390, 391, 450)] + \
         [(2, 50), (2, 117), (3, 33), (3, 100), (4, 64), (5, 21), (7, 13), (8, 40), (12, 40), (16, 9), (2, 282), (3, 200), (1, 700), (1, 1000), (4, 500)] + \
         [(1, t) for t in (350, 400, 500, 550, 600, 650, 750, 800, 850, 900, 950, 1100, 1400, 1600)] + [(2, 700), (3, 500), (4, 400)] + \
         [(32, 500), (8, 300), (9, 130), (24, 57)]   # snake-ordered job plans; batches whose tiles qualify for XCD grouping only together
if len(sys.argv) > 3:          # optional third argument: that many extra random shapes (seeded): 1 <= B <= 5, 1 <= T <= 1600
    rng = np.random.default_rng(20260)
    shapes = shapes + [(int(rng.integers(1, 6)), int(rng.integers(1, 1601))) for _ in range(int(sys.argv[3]))]
code = r'''
import sys, json, hashlib, torch
sys.path.insert(0, "iris-tts_amd")
from iris._engine import GeneratorEngine
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
shapes = json.loads(sys.argv[1]); dtype = sys.argv[2]
cfg = GeneratorConfig(); eng = GeneratorEngine(cfg, seeded_state_dict(cfg, seed=11, gain=1.15, post_gain=12.0), torch.device("cuda", 0))
out = {}
for B, T in shapes:
    y = eng.forward(torch.from_numpy(seeded_mel(100 + T, B, T)).cuda(), dtype=dtype)
    out[f"{B}x{T}"] = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()
print(json.dumps(out))
'''
res = {}
for name, lib in (("release", None), ("ref", os.path.abspath(sys.argv[1]))):
    env = dict(os.environ)
    if lib: env["IRIS_HIFIGAN_LIB"] = lib
    else: env.pop("IRIS_HIFIGAN_LIB", None)
    p = subprocess.run([sys.executable, "-c", code, json.dumps(shapes), sys.argv[2] if len(sys.argv) > 2 else "f32"], env=env, capture_output=True, text=True)
    if p.returncode: print(p.stderr[-2000:]); sys.exit(1)
    res[name] = json.loads(p.stdout.strip().splitlines()[-1])
bad = [k for k in res["release"] if res["release"][k] != res["ref"][k]]
print(f"{len(res['release'])} shapes compared bitwise; differing: {bad}")
sys.exit(1 if bad else 0)
