import ctypes, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "iris-tts_amd"), str(ROOT)]
from iris import _native
lib = _native.load()
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
B, L, C, k = 1, 64, 32, 1
x = (np.arange(L)[None, None, :] * 1.0 + np.arange(C)[None, :, None] * 100.0).astype(np.float32)   # x[c, t] = 100 c + t
w = np.zeros((C, C, k), np.float32); w[np.arange(C), np.arange(C), 0] = 1.0
b = np.zeros(C, np.float32)
xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda()
yd = torch.full((B, L, C), float("nan"), device="cuda")
_native.check("op", lib.iris_hifigan_op_conv1d_f32s(xd.data_ptr(), fp(w), fp(b), None, yd.data_ptr(), B, L, C, k, 1, 0.1, None))
got = yd.cpu().numpy()[0]          # [t, c]
print("want y[t,c] = 100 c + t")
print(np.round(got[:6, :10], 2))
print(np.round(got[30:34, 28:32], 2))
