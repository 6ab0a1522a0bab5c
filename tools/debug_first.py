"""Debug helper: stop the forward after the first MRF step of stage 3 and diff xt_j with the oracle."""
import os, sys, numpy as np, torch, torch.nn.functional as F
os.environ["IRIS_HIFIGAN_STOP_AFTER_MRF"] = "300"
sys.path.insert(0, "iris-tts_amd"); sys.path.insert(0, ".")
from iris._engine import GeneratorEngine
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
from oracle import hifigan_oracle as orc
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = GeneratorConfig(); sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)
mel = seeded_mel(1002, 1, T)
eng = GeneratorEngine(cfg, sd, torch.device("cuda", 0))
eng.forward(torch.from_numpy(mel).cuda()); torch.cuda.synchronize()
ws = eng._workspace.view(torch.float32)
al = lambda n: (n + 63) // 64 * 64
off = al(T * 512); off += al(T * 8192)
bufs = {}
for j in range(3):
    bufs[f"y{j}"] = off; off += al(T * 8192)
    bufs[f"xt{j}"] = off; off += al(T * 8192)
folded = orc.to_torch_folded(sd); taps = {}
orc.generator_forward_torch(folded, mel, taps=taps)
x = taps["ups.3"]; L = x.shape[2]
for j in range(3):
    p = f"resblocks.{9 + j}"; w1, b1 = folded[f"{p}.convs1.0.weight"], folded[f"{p}.convs1.0.bias"]; k = w1.shape[-1]
    ref = F.conv1d(F.leaky_relu(x, 0.1), w1, b1, dilation=1, padding=(k - 1) // 2)[0].numpy().T
    g = ws[bufs[f"xt{j}"]: bufs[f"xt{j}"] + L * 32].view(L, 32).cpu().numpy()
    e = np.abs(g - ref); bad = np.argwhere(e > 1e-5)
    print(f"xt{j} (k={k}) max err %.3g n bad %d of %d" % (e.max(), len(bad), e.size))
    if len(bad):
        rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
        print("   bad rows:", rows[:40], "... n", len(rows)); print("   rows mod 64:", np.unique(rows % 64)[:64]); print("   bad cols:", cols)
        r0, c0 = bad[0]; print("   e.g. row", r0, "got", g[r0, :8], "want", ref[r0, :8])
    if len(bad) and j == 0:
        np.set_printoptions(linewidth=200, precision=5, suppress=True)
        for r in (12, 13, 28, 44):
            print("   row", r, "diff:", (g[r] - ref[r]))
        print("   bias:", b1.numpy())
        # what would the value be without the last tap / without bias?
        xin = F.leaky_relu(x, 0.1)[0].numpy()      # [32, L]
        w = w1.numpy()                               # [co, ci, k]
        r = 12
        for kk in range(k):
            contrib = np.array([sum(w[co, ci, kk] * (xin[ci, r + kk - (k - 1) // 2] if 0 <= r + kk - (k - 1) // 2 < L else 0) for ci in range(32)) for co in range(32)])
            print("   tap", kk, "contribution to row 12:", contrib[[1, 5, 9, 13]])
