import sys, numpy as np, torch
sys.path.insert(0, "iris-tts_amd"); sys.path.insert(0, ".")
from iris._engine import GeneratorEngine
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
from oracle import hifigan_oracle as orc
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cfg = GeneratorConfig(); sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)
mel = seeded_mel(1002, 1, T)
eng = GeneratorEngine(cfg, sd, torch.device("cuda", 0))
got = eng.forward(torch.from_numpy(mel).cuda()).cpu().numpy()[0]
want = orc.generator_forward_torch(orc.to_torch_folded(sd), mel).numpy()[0, 0]
err = np.abs(got - want)
bad = np.nonzero(err > 1e-4)[0]
print("T", T, "max err", err.max(), "n bad", bad.size)
if bad.size:
    # cluster bad positions
    gaps = np.nonzero(np.diff(bad) > 2000)[0]
    starts = np.concatenate([[bad[0]], bad[gaps + 1]]); ends = np.concatenate([bad[gaps], [bad[-1]]])
    for s, e in list(zip(starts, ends))[:40]:
        print("bad region", s, e, "centre/256 = %.2f frames, centre sample %d" % ((s + e) / 2 / 256, (s + e) // 2), "max", err[s:e + 1].max())
    print("regions:", len(starts))
