"""Wall time per forward of GeneratorEngine.forward (no profiling events), median of R rounds of N back-to-back forwards (diagnostics).
usage: [IRIS_HIFIGAN_LIB=...] python tools/forward_time.py "B T" ["B T" ...] [--dtype f32|bf16]"""
import sys, time, torch
sys.path.insert(0, "iris-tts_amd")
from iris._engine import GeneratorEngine
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
args = [a for a in sys.argv[1:] if not a.startswith("--")]
dtype = "bf16" if "--dtype=bf16" in sys.argv or ("--dtype" in sys.argv and sys.argv[sys.argv.index("--dtype") + 1] == "bf16") else "f32"
args = [a for a in args if a not in ("f32", "bf16")]
cfg = GeneratorConfig(); eng = GeneratorEngine(cfg, seeded_state_dict(cfg), torch.device("cuda", 0))
for shape in args:
    B, T = (int(v) for v in shape.split())
    mel = torch.from_numpy(seeded_mel(1, B, T)).cuda()
    for _ in range(5): eng.forward(mel, dtype=dtype)
    N = max(5, min(200, int(400000 / (B * T)))); ts = []
    for r in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(N): eng.forward(mel, dtype=dtype)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3 / N)
    ts.sort()
    print(f"{B} x {T} {dtype}: {ts[3]:.4f} ms per forward (min {ts[0]:.4f})")
