"""How much of a forward's launch ramps and tails can independent work on a second stream fill?  (diagnostics)

Two engines (two handles, two workspaces) vocode the same (batch, frames) mel: N forwards each, (a) back to back on ONE stream,
(b) concurrently on TWO streams.  (b) / (a) < 1 is the share of a forward that a neighbour's blocks can use -- the upper bound of what
running the MRF branches of one forward as concurrent streams could win.  usage: python tools/two_stream_overlap.py [B T [dtype]]"""
import sys, time, torch
sys.path.insert(0, "iris-tts_amd")
from iris._engine import GeneratorEngine
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
cfg = GeneratorConfig(); dev = torch.device("cuda", 0)
sd = seeded_state_dict(cfg)
e1, e2 = GeneratorEngine(cfg, sd, dev), GeneratorEngine(cfg, sd, dev)
mel = torch.from_numpy(seeded_mel(1, B, T)).cuda()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 20
def serial():
    with torch.cuda.stream(s1):
        for _ in range(N): e1.forward(mel, dtype=dtype); e2.forward(mel, dtype=dtype)
def concurrent():
    for _ in range(N):
        with torch.cuda.stream(s1): e1.forward(mel, dtype=dtype)
        with torch.cuda.stream(s2): e2.forward(mel, dtype=dtype)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / (2 * N)
for f in (serial, concurrent): f()
torch.cuda.synchronize()
for r in range(4):
    a, b = timed(serial), timed(concurrent)
    print(f"{B} x {T} {dtype}: one stream {a:.3f} ms per forward, two streams {b:.3f} ms per forward, ratio {b / a:.3f}")
