"""Distance of the bf16-storage path to its CPU restatement and to the fp32 path (diagnostic)."""
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
for (B, T, seed, log_mel) in [(1, 1, 9, False), (1, 100, 1001, False), (3, 57, 5, True), (5, 2, 10, False), (2, 300, 4, True)]:
    mel = seeded_mel(seed, B, T, log_mel=log_mel)
    md = torch.from_numpy(mel).to(dev)
    got16 = eng.forward(md, dtype="bf16").cpu().numpy()
    got32 = eng.forward(md).cpu().numpy()
    emu = orc.generator_forward_bf16(folded, mel).numpy()[:, 0, :]
    ref = orc.generator_forward_torch(folded, mel).numpy()[:, 0, :]
    print(f"B={B} T={T}: |hip16-emu|max={np.abs(got16 - emu).max():.3e} mean={np.abs(got16 - emu).mean():.3e}  "
          f"|hip16-fp32|max={np.abs(got16 - got32).max():.3e} mean={np.abs(got16 - got32).mean():.3e}  "
          f"|emu-ref|max={np.abs(emu - ref).max():.3e}  |hip32-ref|max={np.abs(got32 - ref).max():.3e} "
          f"finite={np.isfinite(got16).all()} rms={np.sqrt((ref ** 2).mean()):.3f}", flush=True)
