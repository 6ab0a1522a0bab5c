"""Host-inclusive rate of the drop-in entry point (numpy in -> numpy out, PCIe both ways, synchronous),
for DESIGN.md.  Never the bench `value`."""
import sys, time, tempfile, numpy as np, torch
from pathlib import Path
sys.path.insert(0, "iris-tts_amd")
from iris import hifigan_pretrained as hp
from iris._weights import seeded_mel, seeded_state_dict
with tempfile.TemporaryDirectory() as td:
    ck = Path(td) / "generator.ckpt"
    torch.save({k: torch.from_numpy(v) for k, v in seeded_state_dict().items()}, ck)
    for B, T in ((1, 1000), (1, 100), (32, 500)):
        mel = seeded_mel(1, B, T)
        hp.infer_hifigan(mel, checkpoint_path=ck)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); out = hp.infer_hifigan(mel, checkpoint_path=ck); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[len(ts) // 2]
        print(f"host-inclusive infer_hifigan B={B} T={T}: {t * 1e3:.2f} ms  {B * T * 256 / t / 1e6:.1f} M samples/s  out {out.shape} {out.dtype}")
