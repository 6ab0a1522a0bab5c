"""Split-product mode (dtype "f32s") against the fp32 oracle and the exact fp32 path (diagnostic)."""
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
sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)
folded = orc.to_torch_folded(sd)
eng = GeneratorEngine(cfg, sd, dev)
for (B, T, seed, log_mel) in [(1, 1, 9, False), (1, 100, 1001, False), (3, 57, 5, True), (2, 300, 4, True), (1, 1000, 1002, False)]:
    mel = seeded_mel(seed, B, T, log_mel=log_mel)
    md = torch.from_numpy(mel).to(dev)
    s = eng.forward(md, dtype="f32s").cpu().numpy()
    f = eng.forward(md).cpu().numpy()
    ref = orc.generator_forward_torch(folded, mel).numpy()[:, 0, :]
    print(f"B={B} T={T}: |f32s-ref| max {np.abs(s - ref).max():.3e} mean {np.abs(s - ref).mean():.3e}   |f32-ref| max {np.abs(f - ref).max():.3e}", flush=True)
