"""Single split-product conv against the numpy oracle (diagnostic)."""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "iris-tts_amd"), str(ROOT)]
from iris import _native  # noqa: E402
from oracle import hifigan_oracle as orc  # noqa: E402

lib = _native.load()
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
for (B, L, C, k, d, use_res) in [(1, 64, 32, 3, 1, False), (1, 300, 32, 11, 5, True), (1, 200, 64, 7, 3, True), (1, 130, 128, 3, 1, False), (1, 70, 256, 11, 3, True)]:
    rng = np.random.default_rng(L + C + k)
    x = rng.standard_normal((B, C, L)).astype(np.float32)
    w = (rng.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    res = rng.standard_normal((B, C, L)).astype(np.float32) if use_res else None
    want = orc.conv1d_np(orc.lrelu_np(x, 0.1), w, b, d)
    if use_res:
        want = want + res
    xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda()
    rd = torch.from_numpy(np.ascontiguousarray(res.transpose(0, 2, 1))).cuda() if use_res else None
    yd = torch.full((B, L, C), float("nan"), device="cuda")
    _native.check("op", lib.iris_hifigan_op_conv1d_f32s(xd.data_ptr(), fp(w), fp(b), rd.data_ptr() if use_res else None, yd.data_ptr(), B, L, C, k, d, 0.1, None))
    got = yd.cpu().numpy().transpose(0, 2, 1)
    err = np.abs(got - want)
    bad = np.argwhere(err > 1e-3)
    print(f"L={L} C={C} k={k} d={d} res={use_res}: max err {np.nanmax(err):.3e}, nan {np.isnan(got).sum()}, bad {len(bad)} first {bad[:4].tolist()}", flush=True)
