"""Prints the per-launch HIP-event records of one forward (diagnostics).  usage: [DTYPE=bf16] python tools/per_launch.py B T"""
import os, sys, torch
sys.path.insert(0, "iris-tts_amd")
from iris._engine import GeneratorEngine
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cfg = GeneratorConfig(); eng = GeneratorEngine(cfg, seeded_state_dict(cfg), torch.device("cuda", 0))
mel = torch.from_numpy(seeded_mel(1, B, T)).cuda()
dtype = os.environ.get("DTYPE", "f32")
for _ in range(3): eng.forward(mel, dtype=dtype)
torch.cuda.synchronize(); eng.set_profiling(1)
for _ in range(5): eng.forward(mel, dtype=dtype)
torch.cuda.synchronize(); recs = eng.read_profile(); n = len(recs) // 5
for i in range(n):
    ms = sorted(recs[i + k * n]["ms"] for k in range(5))[2]; r = recs[i]
    print(f"{i:2d} {r['kind']:18s} stage {r['stage']:2d} step {r['step']} {ms * 1e3:8.1f} us  {r['flops'] / ms / 1e9:7.1f} TFLOP/s  {r['bytes'] / ms / 1e6:7.0f} GB/s")
