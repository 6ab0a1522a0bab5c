"""Debug helper: compare the stage-3 MRF branch outputs left in the workspace with the oracle."""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, "iris-tts_amd"); sys.path.insert(0, ".")
from iris._engine import GeneratorEngine
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
from oracle import hifigan_oracle as orc
T = int(sys.argv[1]) if len(sys.argv) > 1 else 600
cfg = GeneratorConfig(); sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)
mel = seeded_mel(1002, 1, T)
eng = GeneratorEngine(cfg, sd, torch.device("cuda", 0))
got = eng.forward(torch.from_numpy(mel).cuda()); torch.cuda.synchronize()
ws = eng._workspace.view(torch.float32)
al = lambda n: (n + 63) // 64 * 64
frames = T
off = al(frames * 512); up_off = off; off += al(frames * 8192)
bufs = {}
for j in range(3):
    bufs[f"y{j}"] = off; off += al(frames * 8192)
    bufs[f"xt{j}"] = off; off += al(frames * 8192)
folded = orc.to_torch_folded(sd)
taps = {}
orc.generator_forward_torch(folded, mel, taps=taps)
x = taps["ups.3"]                                    # [1,32,L]
L = x.shape[2]
for j in range(3):
    p = f"resblocks.{9 + j}"
    r = x
    for m, d in enumerate((1, 3, 5)):
        w1, b1 = folded[f"{p}.convs1.{m}.weight"], folded[f"{p}.convs1.{m}.bias"]
        w2, b2 = folded[f"{p}.convs2.{m}.weight"], folded[f"{p}.convs2.{m}.bias"]
        k = w1.shape[-1]
        xt = F.conv1d(F.leaky_relu(r, 0.1), w1, b1, dilation=d, padding=(k * d - d) // 2)
        if m == 2:
            xt_last = xt
        r = F.conv1d(F.leaky_relu(xt, 0.1), w2, b2, padding=(k - 1) // 2) + r
    for name, ref in ((f"xt{j}", xt_last), (f"y{j}", r)):
        g = ws[bufs[name]: bufs[name] + L * 32].view(L, 32).cpu().numpy()
        e = np.abs(g - ref[0].numpy().T)
        bad = np.argwhere(e > 1e-4)
        print(name, "max err %.3g" % e.max(), "n bad", len(bad))
        if len(bad):
            rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
            print("   rows %d..%d (n=%d) first rows %s" % (rows.min(), rows.max(), len(rows), rows[:24]))
            print("   rows mod 256 hist (top):", np.bincount(rows % 256, minlength=256).nonzero()[0][:40])
            print("   cols:", cols)
            r0 = rows[0]
            print("   sample row", r0, "got", g[r0, :8], "want", ref[0].numpy().T[r0, :8])
