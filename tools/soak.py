"""Soak run (diagnostics): for S seconds, forwards of seeded random (batch, frames, dtype) shapes through ONE engine, alternating
with a second engine on a second stream; every (shape, dtype, mel seed) is repeated now and then and must reproduce its first
waveform BITWISE (the launches have no atomics or order-dependent sums: any difference is a race or a stale buffer); all samples
finite and |wav| <= 1; device memory in use must not grow.  usage: python tools/soak.py [seconds=240]"""
import hashlib, os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "iris-tts_amd"))
from iris._engine import GeneratorEngine
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
S = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
dev = torch.device("cuda", 0); cfg = GeneratorConfig(); sd = seeded_state_dict(cfg, seed=77, gain=1.15, post_gain=12.0)
eng_a = GeneratorEngine(cfg, sd, dev); eng_b = GeneratorEngine(cfg, sd, dev)
stream_b = torch.cuda.Stream(device=dev)
rng = np.random.default_rng(99)
pool = [(int(rng.integers(1, 5)), int(rng.integers(1, 1301)), str(rng.choice(["f32", "f32", "bf16", "f32s"]))) for _ in range(60)]
seen, n_fw, n_rep, t0 = {}, 0, 0, time.time()
free0 = None
while time.time() - t0 < S:
    B, T, dt = pool[int(rng.integers(0, len(pool)))]
    mel = torch.from_numpy(seeded_mel(500 + B * 10000 + T, B, T)).to(dev)
    use_b = bool(rng.integers(0, 2))
    if use_b:
        stream_b.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream_b):
            y = eng_b.forward(mel, dtype=dt)
        stream_b.synchronize()
    else:
        y = eng_a.forward(mel, dtype=dt)
    h = y.cpu().numpy()
    assert np.isfinite(h).all() and np.abs(h).max() <= 1.0, (B, T, dt)
    key = (B, T, dt); dig = hashlib.sha256(h.tobytes()).hexdigest()
    if key in seen:
        assert seen[key] == dig, ("not reproducible", key, "engine b" if use_b else "engine a")
        n_rep += 1
    else:
        seen[key] = dig
    n_fw += 1
    if n_fw == 200: free0 = torch.cuda.mem_get_info(dev)[0]
free1 = torch.cuda.mem_get_info(dev)[0]
print(f"{n_fw} forwards in {time.time() - t0:.0f} s over {len(seen)} (batch, frames, dtype) shapes, {n_rep} repeats all bitwise equal to their first run; "
      f"free device memory after 200 forwards {free0 / 2**20 if free0 else -1:.0f} MiB, at the end {free1 / 2**20:.0f} MiB")
assert free0 is None or free1 >= free0 - (64 << 20), "device memory in use grew"
