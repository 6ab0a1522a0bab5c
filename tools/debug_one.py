"""One forward at a given shape, compared with the oracle (diagnostic)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "iris-tts_amd"), str(ROOT)]
from iris._engine import GeneratorEngine  # noqa: E402
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict  # noqa: E402
from oracle import hifigan_oracle as orc  # noqa: E402

B, T = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
cfg = GeneratorConfig()
sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)
eng = GeneratorEngine(cfg, sd, dev)
mel = seeded_mel(5, B, T)
print("forward", B, T, flush=True)
got = eng.forward(torch.from_numpy(mel).to(dev))
torch.cuda.synchronize()
print("done", flush=True)
want = orc.generator_forward_torch(orc.to_torch_folded(sd), mel).numpy()[:, 0, :]
print("max err", float(np.abs(got.cpu().numpy() - want).max()), flush=True)
