"""Worst-case distance of the split-product mode to the fp32 oracle over seeds / weight gains (diagnostic)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "iris-tts_amd"), str(ROOT)]
from iris._engine import GeneratorEngine  # noqa: E402
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict  # noqa: E402
from oracle import hifigan_oracle as orc  # noqa: E402

dev = torch.device("cuda", 0)
cfg = GeneratorConfig()
worst = 0.0
for wseed, gain, post in [(1, 1.0, 1.0), (2, 1.18, 20.0), (3, 1.25, 30.0), (4, 1.3, 10.0), (5, 1.1, 50.0)]:
    sd = seeded_state_dict(cfg, seed=wseed, gain=gain, post_gain=post)
    folded = orc.to_torch_folded(sd)
    eng = GeneratorEngine(cfg, sd, dev)
    for mseed, log_mel in [(11, False), (12, True)]:
        mel = seeded_mel(mseed, 2, 150, log_mel=log_mel)
        md = torch.from_numpy(mel).to(dev)
        ref = orc.generator_forward_torch(folded, mel).numpy()[:, 0, :]
        es = float(np.abs(eng.forward(md, dtype="f32s").cpu().numpy() - ref).max())
        ef = float(np.abs(eng.forward(md, dtype="f32").cpu().numpy() - ref).max())
        worst = max(worst, es)
        print(f"weights seed {wseed} gain {gain} post {post} mel {mseed} log={log_mel}: rms {np.sqrt((ref**2).mean()):.3f} max|wav| {np.abs(ref).max():.3f}  f32s err {es:.3e}  f32 err {ef:.3e}", flush=True)
    eng.close()
print("worst f32s error", worst)
