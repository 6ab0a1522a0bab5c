"""The split-product mode (dtype "f32s"): fp32 storage and accumulation, ResBlock conv products formed from two bf16
terms per operand (hi*hi + hi*mid + mid*hi on the bf16 MFMA).  It is held to the SAME bar as the fp32 path --
north_star's 1e-4 max-abs against the reference fp32 generator (goldens and oracle) -- and observed at <= 1e-5."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import hifigan_oracle as orc

pytestmark = pytest.mark.gpu

TOL_WAV = 1e-4          # north_star
TOL_LAYER = 6e-5        # relative to max|reference output| of one conv (two-term split: ~2^-16 per product)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda", 0)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


@pytest.mark.parametrize("B,L,C,k,d,use_res", [
    (1, 64, 32, 3, 1, False), (2, 517, 32, 11, 5, True), (1, 200, 64, 7, 3, True), (2, 131, 64, 3, 1, False),
    (1, 130, 128, 7, 5, True), (1, 70, 256, 11, 3, True), (1, 264, 256, 3, 1, False), (1, 1, 32, 11, 5, True)])
def test_f32s_conv1d_matches_oracle(B, L, C, k, d, use_res):
    from iris import _native
    lib = _native.load()
    rng = np.random.default_rng(B * 1000 + L + C + k + d)
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
    _native.check("op_conv1d_f32s", lib.iris_hifigan_op_conv1d_f32s(
        xd.data_ptr(), _fp(w), _fp(b), rd.data_ptr() if use_res else None, yd.data_ptr(), B, L, C, k, d, 0.1, None))
    got = yd.cpu().numpy().transpose(0, 2, 1)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= TOL_LAYER * np.abs(want).max()


@pytest.mark.parametrize("B,L,Ci,Co,k,u", [(1, 40, 512, 256, 16, 8), (2, 130, 256, 128, 16, 8), (1, 300, 128, 64, 4, 2),
                                            (2, 517, 64, 32, 4, 2), (1, 1, 64, 32, 4, 2), (1, 700, 64, 32, 4, 2)])
def test_f32s_conv_transpose1d_matches_oracle(B, L, Ci, Co, k, u):
    from iris import _native
    lib = _native.load()
    rng = np.random.default_rng(L + Ci + k)
    x = rng.standard_normal((B, Ci, L)).astype(np.float32)
    w = (rng.standard_normal((Ci, Co, k)) / np.sqrt(Ci * k / u)).astype(np.float32)
    b = rng.standard_normal(Co).astype(np.float32)
    want = orc.conv_transpose1d_np(orc.lrelu_np(x, 0.1), w, b, u, (k - u) // 2)
    xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda()
    yd = torch.full((B, L * u, Co), float("nan"), device="cuda")
    _native.check("op_conv_transpose1d_f32s", lib.iris_hifigan_op_conv_transpose1d_f32s(
        xd.data_ptr(), _fp(w), _fp(b), yd.data_ptr(), B, L, Ci, Co, k, u, 0.1, None))
    got = yd.cpu().numpy().transpose(0, 2, 1)
    assert got.shape == want.shape and np.isfinite(got).all()
    assert np.abs(got - want).max() <= TOL_LAYER * np.abs(want).max()


@pytest.mark.parametrize("case", ["v1_default_T4_taps", "v1_default_B2_T16", "v1_amplified_T24"])
def test_f32s_generator_matches_reference_goldens(case, golden, case_setup, dev):
    """Against the waveform the REFERENCE produced for the same weights and mel."""
    from iris._engine import GeneratorEngine
    cfg, sd = case_setup(case)
    g = golden(case)
    eng = GeneratorEngine(cfg, sd, dev)
    got = eng.forward(torch.from_numpy(g["mel"]).to(dev), dtype="f32s").cpu().numpy()
    assert np.abs(got - g["wav"][:, 0, :]).max() <= TOL_WAV
    eng.close()


@pytest.mark.parametrize("B,T,seed,log_mel", [(1, 100, 1001, False), (3, 57, 5, True), (1, 1, 9, False), (2, 300, 4, True),
                                               (1, 1000, 1002, False)])
def test_f32s_generator_matches_oracle(B, T, seed, log_mel, dev):
    from iris._engine import GeneratorEngine
    from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
    cfg = GeneratorConfig()
    sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)
    mel = seeded_mel(seed, B, T, log_mel=log_mel)
    eng = GeneratorEngine(cfg, sd, dev)
    md = torch.from_numpy(mel).to(dev)
    got = eng.forward(md, dtype="f32s")
    assert torch.equal(eng.forward(md, dtype="f32s"), got)                      # deterministic
    want = orc.generator_forward_torch(orc.to_torch_folded(sd), mel).numpy()[:, 0, :]
    err = np.abs(got.cpu().numpy() - want).max()
    assert err <= TOL_WAV, err
    assert err <= 3e-5          # observed <= 1e-5: an order of magnitude inside the budget
    eng.close()


def test_f32s_batch_independence_and_unsupported_config(dev, case_setup):
    from iris import _native
    from iris._engine import GeneratorEngine
    from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
    cfg = GeneratorConfig()
    eng = GeneratorEngine(cfg, seeded_state_dict(cfg, seed=3, gain=1.1, post_gain=10.0), dev)
    mel = torch.from_numpy(seeded_mel(77, 4, 40)).to(dev)
    full = eng.forward(mel, dtype="f32s").clone()
    for b in range(4):
        assert torch.equal(eng.forward(mel[b:b + 1].contiguous(), dtype="f32s")[0], full[b])
    assert eng.workspace_bytes(2, 50, "f32s") == eng.workspace_bytes(2, 50, "f32")
    eng.close()
    cfg2, sd2 = case_setup("small_cfg_B3_T19")      # ResBlock channels 24 / 12 / 6: no split-product mode
    eng2 = GeneratorEngine(cfg2, sd2, dev)
    with pytest.raises(_native.NativeCallError) as exc:
        eng2.forward(torch.zeros((1, cfg2.in_channels, 5), device=dev), dtype="f32s")
    assert "multiples of 32" in str(exc.value)
    eng2.close()
    # a single layer with a channel count the tiles do not divide is refused as well (96 = 1.5 x 64)
    lib = _native.load()
    x = torch.zeros((1, 8, 96), device=dev)
    w = np.zeros((96, 96, 3), np.float32)
    rc = lib.iris_hifigan_op_conv1d_f32s(x.data_ptr(), _fp(w), _fp(np.zeros(96, np.float32)), None, x.data_ptr(), 1, 8, 96, 3, 1, 0.1, None)
    assert rc == 4      # IRIS_HIFIGAN_UNSUPPORTED
