"""AddressSanitizer + UBSan over the HOST side of the C-ABI.

``make -C iris-tts_amd/csrc asan`` builds the library's host pass with ``-fsanitize=address,undefined`` (device code and
its registration are stubbed: the build loads on a box without a GPU and can launch nothing).  A child process --
the sanitizer runtime has to be preloaded into the interpreter -- then drives
  * ``iris_hifigan_describe_plan`` over a shape sweep: the forward's own argument checks, workspace layout and every
    launch plan (mrf_plan, pair_f32_plan, the bf16 pair plans, conv / conv_post grids) with caller-controlled B and T
    at the limits the ABI admits (B up to 65535, T*hop up to 2^30);
  * ``iris_hifigan_create`` up to its first HIP call: validation and the weight packers (fp32 MFMA fragments) on the V1
    configuration and on a narrow one with ragged channel counts -- the upload then fails (no device) and the
    half-built generator is torn down, which is also checked for leaks of the error path.
GPU AddressSanitizer is not available on this pool (and not attempted)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "iris-tts_amd" / "csrc"
ASAN_LIB = CSRC / "libiris_hifigan_asan.so"

CHILD = r'''
import ctypes, sys
import numpy as np
sys.path[:0] = [sys.argv[2]]
from iris import _native
from iris._weights import GeneratorConfig, expected_weight_count

lib = ctypes.CDLL(sys.argv[1])
for name, (restype, argtypes) in _native.SYMBOLS.items():
    fn = getattr(lib, name); fn.restype = restype; fn.argtypes = argtypes
assert lib.iris_hifigan_abi_version() == _native.ABI_VERSION

def err():
    return (lib.iris_hifigan_last_error() or b"").decode()

n_ok = n_refused = 0
configs = [GeneratorConfig(), GeneratorConfig(upsample_initial_channel=32),
           GeneratorConfig(upsample_initial_channel=128, upsample_rates=(4, 4), upsample_kernel_sizes=(8, 8))]
for cfg in configs:
    hop = cfg.hop_length
    ccfg = _native.make_config(cfg)
    frames = [1, 2, 13] + list(range(117, 130)) + [1000, (1 << 22) // hop, (1 << 30) // hop]
    for dtype in (0, 1, 2):
        for B in (1, 7, 65535):
            for T in frames:
                plan = _native.Plan()
                rc = lib.iris_hifigan_describe_plan(ctypes.byref(ccfg), B, T, dtype, 256, ctypes.byref(plan))
                if rc == 0:
                    n_ok += 1
                    assert plan.n_launches >= 2 + 2 * cfg.num_upsamples, (B, T, dtype, plan.n_launches)
                    assert plan.workspace_bytes > 0
                    for i in range(min(plan.n_launches, _native.MAX_PLAN_LAUNCHES)):
                        l = plan.launches[i]
                        assert l.block == 256 and l.grid[0] >= 1 and 1 <= l.grid[1] <= 65535 and l.lds_bytes <= 160 * 1024, (B, T, dtype, i, l.kernel)
                else:
                    n_refused += 1              # a documented refusal (e.g. a dtype the narrow config cannot take), never a crash
                    assert rc in (1, 2, 4), (rc, err())
    # beyond the admitted limits: refused with a message
    plan = _native.Plan()
    assert lib.iris_hifigan_describe_plan(ctypes.byref(ccfg), 65536, 10, 0, 256, ctypes.byref(plan)) != 0
    assert lib.iris_hifigan_describe_plan(ctypes.byref(ccfg), 1, (1 << 30) // hop + 1, 0, 256, ctypes.byref(plan)) != 0
    assert lib.iris_hifigan_describe_plan(ctypes.byref(ccfg), -1, 10, 0, 256, ctypes.byref(plan)) != 0
    # fewer CUs than an MI355X: the plans must still be well-formed
    for cu in (1, 8, 104, 304):
        assert lib.iris_hifigan_describe_plan(ctypes.byref(ccfg), 3, 333, 0, cu, ctypes.byref(plan)) in (0, 4)

# create up to the first HIP call: packers run under the sanitizers, then the upload fails cleanly (no device here)
rng = np.random.default_rng(0)
for cfg in configs[:2]:
    n = expected_weight_count(cfg)
    blob = rng.standard_normal(n).astype(np.float32)
    h = ctypes.c_void_p()
    rc = lib.iris_hifigan_create(ctypes.byref(_native.make_config(cfg)), blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                 ctypes.c_uint64(n), ctypes.byref(h))
    if rc == 0:
        lib.iris_hifigan_destroy(h)             # (a box with a GPU: fine too)
    else:
        assert rc in (2, 3) and "weight upload failed" in err(), (rc, err())
    rc = lib.iris_hifigan_create(ctypes.byref(_native.make_config(cfg)), blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                 ctypes.c_uint64(n - 1), ctypes.byref(h))
    assert rc == 1 and "weight blob has" in err()
print("SANITIZER_SWEEP_OK", n_ok, n_refused)
'''


def _asan_runtime() -> str:
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"],
                         capture_output=True, text=True, check=True).stdout.strip()
    return out


def test_host_code_under_asan_and_ubsan(tmp_path):
    if not Path("/opt/rocm/bin/hipcc").exists():
        pytest.skip("hipcc not available")
    subprocess.run(["make", "-C", str(CSRC), "asan"], check=True, capture_output=True)
    assert ASAN_LIB.exists()
    runtime = _asan_runtime()
    assert Path(runtime).exists(), runtime
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, LD_PRELOAD=runtime,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66:allocator_may_return_null=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    proc = subprocess.run([sys.executable, str(script), str(ASAN_LIB), str(REPO / "iris-tts_amd")], env=env,
                          capture_output=True, text=True, timeout=900)
    tail = (proc.stdout + proc.stderr)[-4000:]
    assert proc.returncode == 0, tail
    assert "SANITIZER_SWEEP_OK" in proc.stdout, tail
    assert "AddressSanitizer" not in proc.stderr and "runtime error" not in proc.stderr, tail
