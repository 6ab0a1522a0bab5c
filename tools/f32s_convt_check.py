import ctypes, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "iris-tts_amd"), str(ROOT)]
from iris import _native
from oracle import hifigan_oracle as orc
lib = _native.load()
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
for (B, L, Ci, Co, k, u) in [(1, 40, 64, 32, 4, 2), (1, 40, 128, 64, 4, 2), (1, 40, 256, 128, 16, 8), (1, 40, 512, 256, 16, 8)]:
    rng = np.random.default_rng(L + Ci + k)
    x = rng.standard_normal((B, Ci, L)).astype(np.float32)
    w = (rng.standard_normal((Ci, Co, k)) / np.sqrt(Ci * k / u)).astype(np.float32)
    b = rng.standard_normal(Co).astype(np.float32)
    want = orc.conv_transpose1d_np(orc.lrelu_np(x, 0.1), w, b, u, (k - u) // 2)
    xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda()
    yd = torch.full((B, L * u, Co), float("nan"), device="cuda")
    rc = lib.iris_hifigan_op_conv_transpose1d_f32s(xd.data_ptr(), fp(w), fp(b), yd.data_ptr(), B, L, Ci, Co, k, u, 0.1, None)
    got = yd.cpu().numpy().transpose(0, 2, 1)
    err = np.abs(got - want)
    bad = np.argwhere(~(err <= 1e-3))
    print(f"Ci={Ci} Co={Co} k={k} u={u}: rc {rc} max err {np.nanmax(err):.3e} nan {np.isnan(got).sum()} bad {len(bad)} of {got.size}; first {bad[:5].tolist()}; bad rows mod u: {sorted(set((bad[:,2] % u).tolist()))[:8]}; bad co range {bad[:,1].min() if len(bad) else None}-{bad[:,1].max() if len(bad) else None}", flush=True)
