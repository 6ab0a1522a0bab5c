"""ctypes loader for the plain-C oracle (oracle/hifigan_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_MAX = 8


class OrcConfig(ctypes.Structure):
    _fields_ = [
        ("in_channels", ctypes.c_int32), ("upsample_initial_channel", ctypes.c_int32),
        ("num_upsamples", ctypes.c_int32),
        ("upsample_rates", ctypes.c_int32 * _MAX), ("upsample_kernel_sizes", ctypes.c_int32 * _MAX),
        ("num_kernels", ctypes.c_int32), ("resblock_kernel_sizes", ctypes.c_int32 * _MAX),
        ("num_dilations", ctypes.c_int32 * _MAX),
        ("resblock_dilations", (ctypes.c_int32 * _MAX) * _MAX),
        ("pre_kernel_size", ctypes.c_int32), ("post_kernel_size", ctypes.c_int32),
        ("lrelu_slope", ctypes.c_float),
    ]


def load():
    path = _HERE / "libhifigan_oracle.so"
    if not path.exists():
        raise FileNotFoundError(f"{path} not built: run `make -C {_HERE}`")
    lib = ctypes.CDLL(str(path))
    lib.orc_generator_forward.restype = ctypes.c_int
    lib.orc_generator_forward.argtypes = [ctypes.POINTER(OrcConfig), ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


def generator_forward_c(cfg, blob: np.ndarray, mel: np.ndarray) -> np.ndarray:
    """cfg: any object with the GeneratorConfig attributes; blob: folded weights in C-ABI order."""
    lib = load()
    c = OrcConfig()
    c.in_channels, c.upsample_initial_channel = cfg.in_channels, cfg.upsample_initial_channel
    c.num_upsamples = len(cfg.upsample_rates)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        c.upsample_rates[i], c.upsample_kernel_sizes[i] = u, k
    c.num_kernels = len(cfg.resblock_kernel_sizes)
    for j, (k, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
        c.resblock_kernel_sizes[j], c.num_dilations[j] = k, len(dils)
        for m, d in enumerate(dils):
            c.resblock_dilations[j][m] = d
    c.pre_kernel_size, c.post_kernel_size, c.lrelu_slope = 7, 7, 0.1
    mel = np.ascontiguousarray(mel, dtype=np.float32)
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    B, _, T = mel.shape
    hop = int(np.prod(cfg.upsample_rates))
    out = np.empty((B, hop * T), dtype=np.float32)
    rc = lib.orc_generator_forward(ctypes.byref(c), blob.ctypes.data, mel.ctypes.data, B, T, out.ctypes.data)
    if rc != 0:
        raise MemoryError("orc_generator_forward failed")
    return out
