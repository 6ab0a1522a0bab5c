"""ctypes binding of the C-ABI declared in ``include/iris_hifigan.h``.

The shared library is built in-tree by ``iris-tts_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is no fallback: if the library is missing or does not load, every entry point raises
``NativeLibraryError`` -- the product path never computes on the CPU.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path
from typing import Optional

MAX_STAGES = 8
MAX_KERNELS = 8
MAX_DILATIONS = 8
ABI_VERSION = 4
MAX_PLAN_LAUNCHES = 96

DTYPE_F32 = 0
DTYPE_BF16 = 1
DTYPE_F32_SPLIT = 2

STATUS_WORKSPACE_TOO_SMALL = 5
STATUS_NOT_PREPARED = 6

LIB_NAME = "libiris_hifigan.so"
CSRC_DIR = Path(__file__).resolve().parent.parent / "csrc"


class NativeLibraryError(RuntimeError):
    """The HIP extension is missing, stale or failed to load."""


class NativeCallError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, fn: str, status: int, message: str):
        super().__init__(f"{fn} failed with status {status}: {message}")
        self.status = status


class Config(ctypes.Structure):
    """``iris_hifigan_config``"""

    _fields_ = [
        ("in_channels", ctypes.c_int32),
        ("upsample_initial_channel", ctypes.c_int32),
        ("num_upsamples", ctypes.c_int32),
        ("upsample_rates", ctypes.c_int32 * MAX_STAGES),
        ("upsample_kernel_sizes", ctypes.c_int32 * MAX_STAGES),
        ("num_kernels", ctypes.c_int32),
        ("resblock_kernel_sizes", ctypes.c_int32 * MAX_KERNELS),
        ("num_dilations", ctypes.c_int32 * MAX_KERNELS),
        ("resblock_dilations", (ctypes.c_int32 * MAX_DILATIONS) * MAX_KERNELS),
        ("pre_kernel_size", ctypes.c_int32),
        ("post_kernel_size", ctypes.c_int32),
        ("lrelu_slope", ctypes.c_float),
    ]


class LaunchRecord(ctypes.Structure):
    """``iris_hifigan_launch_record``"""

    _fields_ = [
        ("kind", ctypes.c_int32),
        ("stage", ctypes.c_int32),
        ("step", ctypes.c_int32),
        ("launches", ctypes.c_int32),
        ("flops", ctypes.c_double),
        ("bytes", ctypes.c_double),
        ("ms", ctypes.c_float),
        ("reserved2", ctypes.c_float),
    ]


class WorkspaceMap(ctypes.Structure):
    """``iris_hifigan_workspace_map``"""

    _fields_ = [
        ("pre_offset", ctypes.c_uint64),
        ("up_offset", ctypes.c_uint64),
        ("y_offset", ctypes.c_uint64 * MAX_KERNELS),
        ("xt_offset", ctypes.c_uint64 * MAX_KERNELS),
        ("total_bytes", ctypes.c_uint64),
        ("element_bytes", ctypes.c_int32),
        ("launches", ctypes.c_int32),
    ]


class PlanLaunch(ctypes.Structure):
    """``iris_hifigan_plan_launch``"""

    _fields_ = [
        ("kernel", ctypes.c_char * 80),
        ("grid", ctypes.c_uint32 * 3),
        ("block", ctypes.c_uint32),
        ("lds_bytes", ctypes.c_uint64),
    ]


class Plan(ctypes.Structure):
    """``iris_hifigan_plan``"""

    _fields_ = [
        ("workspace_bytes", ctypes.c_uint64),
        ("n_launches", ctypes.c_int32),
        ("cu_count", ctypes.c_int32),
        ("passes", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("launches", PlanLaunch * MAX_PLAN_LAUNCHES),
    ]


# name -> (restype, argtypes); must list every symbol of include/iris_hifigan.h
_c = ctypes
_vp, _i32, _u64, _f = _c.c_void_p, _c.c_int32, _c.c_uint64, _c.c_float
_fp = _c.POINTER(_c.c_float)
SYMBOLS = {
    "iris_hifigan_abi_version": (_i32, []),
    "iris_hifigan_last_error": (_c.c_char_p, []),
    "iris_hifigan_weight_count": (_i32, [_c.POINTER(Config), _c.POINTER(_u64)]),
    "iris_hifigan_create": (_i32, [_c.POINTER(Config), _fp, _u64, _c.POINTER(_vp)]),
    "iris_hifigan_destroy": (_i32, [_vp]),
    "iris_hifigan_prepare": (_i32, [_vp, _i32]),
    "iris_hifigan_release_host_weights": (_i32, [_vp]),
    "iris_hifigan_pause_profiling": (_i32, [_vp, _i32]),
    "iris_hifigan_describe_plan": (_i32, [_c.POINTER(Config), _i32, _i32, _i32, _i32, _c.POINTER(Plan)]),
    "iris_hifigan_workspace_bytes": (_i32, [_vp, _i32, _i32, _i32, _c.POINTER(_u64)]),
    "iris_hifigan_forward": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _u64, _i32, _vp]),
    "iris_hifigan_workspace_layout": (_i32, [_vp, _i32, _i32, _i32, _c.POINTER(WorkspaceMap)]),
    "iris_hifigan_forward_until": (_i32, [_vp, _vp, _i32, _i32, _vp, _u64, _i32, _i32, _i32, _c.POINTER(_i32), _vp]),
    "iris_hifigan_hop_length": (_i32, [_vp, _c.POINTER(_i32)]),
    "iris_hifigan_set_profiling": (_i32, [_vp, _i32]),
    "iris_hifigan_read_profile": (_i32, [_vp, _c.POINTER(LaunchRecord), _i32, _c.POINTER(_i32)]),
    "iris_hifigan_op_conv1d": (_i32, [_vp, _fp, _fp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f, _i32, _vp]),
    "iris_hifigan_op_conv_transpose1d": (_i32, [_vp, _fp, _fp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f, _vp]),
    "iris_hifigan_op_conv_post": (_i32, [_vp, _vp, _vp, _fp, _fp, _vp, _i32, _i32, _i32, _i32, _f, _vp]),
    "iris_hifigan_op_mrf_step": (_i32, [_c.POINTER(_vp), _c.POINTER(_fp), _c.POINTER(_fp), _c.POINTER(_vp), _c.POINTER(_vp), _vp,
                                       _i32, _i32, _i32, _c.POINTER(_i32), _c.POINTER(_i32), _f, _i32, _vp]),
    "iris_hifigan_op_mrf_pair": (_i32, [_c.POINTER(_vp), _c.POINTER(_fp), _c.POINTER(_fp), _c.POINTER(_fp), _c.POINTER(_fp),
                                       _c.POINTER(_vp), _vp, _i32, _i32, _i32, _c.POINTER(_i32), _c.POINTER(_i32), _f, _i32, _vp]),
    "iris_hifigan_op_mrf_pair_bf16": (_i32, [_c.POINTER(_vp), _c.POINTER(_fp), _c.POINTER(_fp), _c.POINTER(_fp), _c.POINTER(_fp),
                                            _c.POINTER(_vp), _i32, _i32, _i32, _i32, _c.POINTER(_i32), _c.POINTER(_i32), _f, _vp]),
    "iris_hifigan_op_mrf_pair_mean_bf16": (_i32, [_c.POINTER(_vp), _c.POINTER(_fp), _c.POINTER(_fp), _c.POINTER(_fp), _c.POINTER(_fp),
                                                 _vp, _i32, _i32, _i32, _i32, _c.POINTER(_i32), _c.POINTER(_i32), _f, _vp]),
    "iris_hifigan_op_conv1d_bf16": (_i32, [_vp, _fp, _fp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f, _vp]),
    "iris_hifigan_op_conv_transpose1d_bf16": (_i32, [_vp, _fp, _fp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f, _vp]),
    "iris_hifigan_op_conv1d_f32s": (_i32, [_vp, _fp, _fp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f, _vp]),
    "iris_hifigan_op_conv_transpose1d_f32s": (_i32, [_vp, _fp, _fp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f, _vp]),
    "iris_postnet_create": (_i32, [_i32, _i32, _i32, _i32, _fp, _u64, _c.POINTER(_vp)]),
    "iris_postnet_destroy": (_i32, [_vp]),
    "iris_postnet_workspace_bytes": (_i32, [_vp, _i32, _i32, _c.POINTER(_u64)]),
    "iris_postnet_forward": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _u64, _vp]),
}

_lib: Optional[ctypes.CDLL] = None


def library_path() -> Path:
    override = os.environ.get("IRIS_HIFIGAN_LIB")
    return Path(override) if override else CSRC_DIR / LIB_NAME


def load() -> ctypes.CDLL:
    """Loads (once) and returns the library with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise NativeLibraryError(
            f"HIP extension not built: {path} is missing. Build it with "
            f"`make -C {CSRC_DIR}` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for the vocoder path.")
    try:
        # torch must own the process's HIP runtime first so that both see the same libamdhip64
        import torch  # noqa: F401
        lib = ctypes.CDLL(str(path), mode=ctypes.RTLD_GLOBAL)
    except OSError as exc:
        raise NativeLibraryError(f"could not load {path}: {exc}") from exc
    for name, (restype, argtypes) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise NativeLibraryError(f"{path} does not export {name}; rebuild the extension") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    version = lib.iris_hifigan_abi_version()
    if version != ABI_VERSION:
        raise NativeLibraryError(f"{path} has ABI version {version}, binding expects {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(fn_name: str, status: int) -> None:
    if status != 0:
        msg = load().iris_hifigan_last_error()
        raise NativeCallError(fn_name, status, msg.decode("utf-8", "replace") if msg else "")


def describe_plan(cfg, batch: int, frames: int, dtype_code: int = DTYPE_F32, cu_count: int = 0) -> dict:
    """Launch plan of one forward, computed on the host alone (``iris_hifigan_describe_plan``: no device needed,
    nothing is launched): workspace bytes and, per launch, kernel name, grid, block and dynamic LDS."""
    lib = load()
    plan = Plan()
    ccfg = make_config(cfg)
    check("iris_hifigan_describe_plan", lib.iris_hifigan_describe_plan(ctypes.byref(ccfg), batch, frames, dtype_code, cu_count,
                                                                       ctypes.byref(plan)))
    n = min(plan.n_launches, MAX_PLAN_LAUNCHES)
    return {"workspace_bytes": int(plan.workspace_bytes), "n_launches": int(plan.n_launches), "cu_count": int(plan.cu_count),
            "passes": int(plan.passes),
            "launches": [{"kernel": plan.launches[i].kernel.decode(), "grid": tuple(plan.launches[i].grid),
                          "block": int(plan.launches[i].block), "lds_bytes": int(plan.launches[i].lds_bytes)} for i in range(n)]}


def make_config(cfg) -> Config:
    """``GeneratorConfig`` -> ``iris_hifigan_config`` (validates the fixed-size limits)."""
    if cfg.num_upsamples > MAX_STAGES:
        raise ValueError(f"at most {MAX_STAGES} upsample stages are supported, got {cfg.num_upsamples}")
    if cfg.num_kernels > MAX_KERNELS:
        raise ValueError(f"at most {MAX_KERNELS} MRF kernels are supported, got {cfg.num_kernels}")
    c = Config()
    c.in_channels = cfg.in_channels
    c.upsample_initial_channel = cfg.upsample_initial_channel
    c.num_upsamples = cfg.num_upsamples
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        c.upsample_rates[i] = u
        c.upsample_kernel_sizes[i] = k
    c.num_kernels = cfg.num_kernels
    for j, (k, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
        if len(dils) > MAX_DILATIONS:
            raise ValueError(f"at most {MAX_DILATIONS} dilations per ResBlock are supported")
        c.resblock_kernel_sizes[j] = k
        c.num_dilations[j] = len(dils)
        for m, d in enumerate(dils):
            c.resblock_dilations[j][m] = d
    c.pre_kernel_size = cfg.pre_kernel_size
    c.post_kernel_size = cfg.post_kernel_size
    c.lrelu_slope = cfg.lrelu_slope
    return c
