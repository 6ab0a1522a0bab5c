"""Pins the oracle (oracle/hifigan_oracle.py) against outputs of the reference itself.

The fixtures in tests/golden/ were produced by importing the reference's own
``iris.hifigan_pretrained`` (tests/golden/make_golden.py); the reference has no test vectors of
its own for this path.  CPU only.
"""
import numpy as np
import pytest

from conftest import oracle_config
from oracle import hifigan_oracle as orc

# The oracle and the reference run the same fp32 math in a different summation order (the numpy
# variant even accumulates in fp64): agreement is at rounding level, far inside the 1e-4 budget.
TOL_ORACLE = 2e-6


@pytest.mark.parametrize("case", ["v1_default_T4_taps", "v1_default_B2_T16", "v1_amplified_T24", "small_cfg_B3_T19"])
def test_torch_oracle_matches_reference_output(case, golden, case_setup):
    cfg, sd = case_setup(case)
    g = golden(case)
    folded = orc.to_torch_folded(sd)
    y = orc.generator_forward_torch(folded, g["mel"], oracle_config(cfg)).numpy()
    assert y.shape == g["wav"].shape
    assert np.abs(y - g["wav"]).max() <= TOL_ORACLE


@pytest.mark.parametrize("case", ["v1_default_T4_taps", "small_cfg_B3_T19"])
def test_numpy_oracle_matches_reference_output(case, golden, case_setup):
    cfg, sd = case_setup(case)
    g = golden(case)
    y = orc.generator_forward_np(orc.fold_state_dict(sd), g["mel"], oracle_config(cfg))
    assert y.shape == g["wav"].shape
    assert np.abs(y - g["wav"]).max() <= TOL_ORACLE


def test_oracle_taps_match_reference_taps(golden, case_setup):
    cfg, sd = case_setup("v1_default_T4_taps")
    g = golden("v1_default_T4_taps")
    taps_t, taps_n = {}, {}
    orc.generator_forward_torch(orc.to_torch_folded(sd), g["mel"], oracle_config(cfg), taps=taps_t)
    orc.generator_forward_np(orc.fold_state_dict(sd), g["mel"], oracle_config(cfg), taps=taps_n)
    names = ["conv_pre"] + [f"ups.{i}" for i in range(4)] + [f"mrf.{i}" for i in range(4)]
    for name in names:
        ref = g[name.replace(".", "_")]
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(taps_t[name].numpy() - ref).max() <= TOL_ORACLE * scale, name
        assert np.abs(taps_n[name] - ref).max() <= TOL_ORACLE * scale, name


def test_amplified_case_exercises_tanh(manifest, golden):
    """The amplified fixture must drive tanh and LeakyReLU off their linear range, otherwise the 1e-4
    tolerance would be vacuous (default-scale outputs are only ~0.06 in magnitude)."""
    meta = manifest["cases"]["v1_amplified_T24"]
    assert meta["pre_tanh_abs_max"] > 2.0
    assert 0.2 < meta["pre_tanh_frac_gt1"] < 0.9
    wav = golden("v1_amplified_T24")["wav"]
    assert np.abs(wav).max() > 0.95 and np.abs(wav).max() <= 1.0


def test_conv_formulas_agree_with_torch_functional():
    """conv1d_np / conv_transpose1d_np (explicit index formulas) vs ATen."""
    import torch
    import torch.nn.functional as F

    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 6, 37)).astype(np.float32)
    for k, d in ((3, 1), (7, 3), (11, 5), (5, 2)):
        w = rng.standard_normal((4, 6, k)).astype(np.float32)
        b = rng.standard_normal(4).astype(np.float32)
        ref = F.conv1d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), dilation=d,
                       padding=int((k * d - d) / 2)).numpy()
        assert np.abs(orc.conv1d_np(x, w, b, d) - ref).max() < 1e-5
    for k, u in ((16, 8), (4, 2), (9, 3), (8, 4), (2, 2)):
        w = rng.standard_normal((6, 5, k)).astype(np.float32)
        b = rng.standard_normal(5).astype(np.float32)
        ref = F.conv_transpose1d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=u,
                                 padding=(k - u) // 2).numpy()
        got = orc.conv_transpose1d_np(x, w, b, u, (k - u) // 2)
        assert got.shape == ref.shape == (2, 5, 37 * u)
        assert np.abs(got - ref).max() < 1e-5


def test_wrapper_shape_rules_match_reference(manifest):
    sem = manifest["wrapper_semantics"]
    for shape_s, per_entry in sem.items():
        if not shape_s.startswith("["):
            continue
        shape = tuple(int(v) for v in shape_s.strip("[]").split(","))
        for entry in ("generator_call", "infer_hifigan"):
            assert list(orc.wrapper_shapes(entry, shape)) == per_entry[entry]["shape"]
            assert per_entry[entry]["dtype"] == "float32"


def test_bf16_restatement_stays_close_to_the_pinned_fp32_oracle():
    """``generator_forward_bf16`` defines the bf16-storage variant (parity unpinned by the reference): its rounding
    points must not move it further from the pinned fp32 restatement than bf16 storage noise."""
    import torch
    from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
    cfg = GeneratorConfig()
    sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)
    folded = orc.to_torch_folded(sd)
    mel = seeded_mel(5, 2, 24, log_mel=True)
    taps16, taps32 = {}, {}
    w16 = orc.generator_forward_bf16(folded, mel, taps=taps16).numpy()
    w32 = orc.generator_forward_torch(folded, mel, taps=taps32).numpy()
    assert w16.shape == w32.shape == (2, 1, 24 * 256)
    d = np.abs(w16 - w32)
    assert d.max() <= 6e-2 and d.mean() <= 5e-3 and d.max() > 1e-5          # close, but really rounded
    # stored tensors hold bf16 values only
    for name in ("conv_pre", "ups.0", "ups.3"):
        t = taps16[name]
        assert torch.equal(t, t.to(torch.bfloat16).to(torch.float32))
        rel = (t - taps32[name]).abs().max() / taps32[name].abs().max()
        assert rel < 5e-2
