"""One grouped fp32 MRF step (iris_hifigan_op_mrf_step) under each forced launch plan, RELEASE library, for reading the kernel
durations out of a `rocprofv3 --kernel-trace --stats` run of this script (the entry point packs its weights on every call, so
only the kernel's own duration means anything).  usage: rocprofv3 --kernel-trace --stats -d <dir> --output-format csv -- python3 tools/forced_plan_probe.py L C [B] [reps]
plans: 0 the library's choice, 1 full-height tiles, 2 half-height, 3 one branch per block, 5 / 6 snake jobs at half / full height."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "iris-tts_amd"))
from iris import _native
L, C = int(sys.argv[1]), int(sys.argv[2]); B = int(sys.argv[3]) if len(sys.argv) > 3 else 1; reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
lib = _native.load()
rng = np.random.default_rng(5)
ks, dils = (3, 7, 11), (3, 3, 3)
xs = [torch.from_numpy(rng.standard_normal((B, L, C)).astype(np.float32)).cuda() for _ in ks]
rs = [torch.from_numpy(rng.standard_normal((B, L, C)).astype(np.float32)).cuda() for _ in ks]
ys = [torch.empty((B, L, C), device="cuda") for _ in ks]
ws = [np.ascontiguousarray((rng.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32)) for k in ks]
bs = [rng.standard_normal(C).astype(np.float32) for _ in ks]
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
vp3, fp3 = ctypes.c_void_p * 3, ctypes.POINTER(ctypes.c_float) * 3
for plan in (0, 1, 2, 3, 5, 6):
    for _ in range(reps):
        st = lib.iris_hifigan_op_mrf_step(vp3(*[t.data_ptr() for t in xs]), fp3(*[fp(w) for w in ws]), fp3(*[fp(b) for b in bs]),
                                          vp3(*[t.data_ptr() for t in rs]), vp3(*[t.data_ptr() for t in ys]), None, B, L, C,
                                          (ctypes.c_int32 * 3)(*ks), (ctypes.c_int32 * 3)(*dils), 0.1, plan, None)
        torch.cuda.synchronize()
    print("plan", plan, "status", st, flush=True)
