"""Whole waveforms of the fp32 path against the ORACLE (oracle/hifigan_oracle.py: generator_forward_torch, the CPU restatement
of HiFiGANModel.forward, reference src/iris/hifigan_pretrained.py:123-143) over N seeded random (batch, frames) shapes -- a
one-off companion to tests/test_planner_sweep.py (diagnostics; it imports oracle/ exactly as the tests do, as the checker).
usage: python tools/oracle_random_sweep.py [N=120] [max_frames=1600] [f32|bf16]   ->  one line per shape + a summary; exit 1 above 1e-4.
bf16: the bf16-storage variant against its CPU restatement (generator_forward_bf16) and the fp32 oracle, bars of tests/test_gpu_bf16.py
(max 6e-2, mean 5e-3: unpinned by the reference, which has no bf16 path)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "iris-tts_amd")); sys.path.insert(0, ROOT)
from iris._engine import GeneratorEngine
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
from oracle import hifigan_oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
TMAX = int(sys.argv[2]) if len(sys.argv) > 2 else 1600
DTYPE = sys.argv[3] if len(sys.argv) > 3 else "f32"
rng = np.random.default_rng(4242)
shapes = [(int(rng.integers(1, 6)), int(rng.integers(1, TMAX + 1))) for _ in range(N)]
dev = torch.device("cuda", 0); cfg = GeneratorConfig()
sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)             # the amplified set: tanh reaches +-0.99
eng = GeneratorEngine(cfg, sd, dev, graph_max_frames=0); folded = orc.to_torch_folded(sd)
torch.set_num_threads(min(32, os.cpu_count() or 8))
worst, t0, mean_worst = (0.0, None), time.time(), 0.0
for n, (B, T) in enumerate(shapes):
    mel = seeded_mel(9000 + n, B, T, log_mel=bool(n & 1))
    got = eng.forward(torch.from_numpy(mel).to(dev), dtype=DTYPE).cpu().numpy()
    idx = sorted({0, B - 1})
    want = orc.generator_forward_torch(folded, mel[idx]).numpy()[:, 0, :]
    err = float(np.abs(got[idx] - want).max())
    if DTYPE == "bf16":
        emu = orc.generator_forward_bf16(folded, mel[idx]).numpy()[:, 0, :]
        d = np.abs(got[idx] - emu)
        mean_worst = max(mean_worst, float(d.mean()), float(np.abs(got[idx] - want).mean()))
        print(f"{n:3d} B={B} T={T:4d} max|wav|={float(np.abs(got).max()):.4f} vs restatement max {float(d.max()):.3e} mean {float(d.mean()):.3e}; vs fp32 oracle max {err:.3e}", flush=True)
        err = max(err, float(d.max()))
    else:
        print(f"{n:3d} B={B} T={T:4d} max|wav|={float(np.abs(got).max()):.4f} max-abs err {err:.3e}", flush=True)
    if err > worst[0]: worst = (err, (B, T))
bar = 1e-4 if DTYPE == "f32" else 6e-2
print(f"{N} shapes ({DTYPE}), worst max-abs error {worst[0]:.3e} at B x T = {worst[1]}; bar {bar:g}" + (f"; worst mean {mean_worst:.3e} (bar 5e-3)" if DTYPE == "bf16" else "") + f"; {time.time() - t0:.0f} s")
sys.exit(1 if worst[0] > bar or mean_worst > 5e-3 else 0)
