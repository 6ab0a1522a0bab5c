"""Time to first chunk and total time of chunked vocoding, plain vs grouped (diagnostic; configs[4] shape)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "iris-tts_amd"), str(ROOT)]
from iris._engine import GeneratorEngine  # noqa: E402
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict  # noqa: E402
from iris.streaming import StreamingVocoder  # noqa: E402

dev = torch.device("cuda", 0)
cfg = GeneratorConfig()
eng = GeneratorEngine(cfg, seeded_state_dict(cfg, seed=3), dev)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
mel = torch.from_numpy(seeded_mel(9, 1, T)).to(dev)
for dtype in ("f32", "bf16"):
    fwd = lambda m: eng.forward(m, dtype=dtype)
    for g in (1, 2, 4, 8):
        sv = StreamingVocoder(fwd, group_chunks=g)
        for _ in range(2):
            list(sv.stream(mel))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first = None
        for i, c in enumerate(sv.stream(mel)):
            if i == 0:
                torch.cuda.synchronize()
                first = time.perf_counter() - t0
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        print(f"{dtype} T={T} group_chunks={g}: first chunk {first * 1e3:.2f} ms, all {total * 1e3:.2f} ms "
              f"({T * 256 / total / 1e6:.1f} M samples/s)", flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); eng.forward(mel); torch.cuda.synchronize()
print(f"one-shot f32: {(time.perf_counter() - t0) * 1e3:.2f} ms")
