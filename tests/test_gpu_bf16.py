"""The bf16-storage path (dtype "bf16", BASELINE.json configs[2]) through the C-ABI on a real MI355X.

The reference has no bf16 path, so no tolerance is pinned by it ("parity unpinned" for this variant).
What is checked instead:
  * single layers against oracle/hifigan_oracle.py on bf16-valued inputs: the products are exact and the
    accumulation is fp32, so the only difference to the restatement is the summation order in front of ONE
    rounding to bf16 -- at most one bf16 ulp, and only on a small fraction of the elements;
  * the whole generator against the pinned fp32 oracle and against the bf16 restatement
    (``generator_forward_bf16``): bf16 storage noise, bounded by TOL_BF16_MAX / TOL_BF16_MEAN below
    (observed: max 2.2e-2, mean 1.8e-3 on waveforms of rms 0.74);
  * size-independent properties at the full configs[2] size: determinism, batch independence (bit-exact).
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import hifigan_oracle as orc

pytestmark = pytest.mark.gpu

TOL_BF16_MAX = 6e-2     # max-abs distance of a bf16-path waveform to the fp32 waveform (tanh range +-1)
TOL_BF16_MEAN = 5e-3    # mean-abs distance


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def engine(dev):
    from iris._engine import GeneratorEngine
    from iris._weights import GeneratorConfig, seeded_state_dict
    cfg = GeneratorConfig()
    sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)
    eng = GeneratorEngine(cfg, sd, dev)
    yield eng, sd
    eng.close()


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _r16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float32).numpy()


def _bf16_cl(x_cf):  # [B,C,L] fp32 numpy (bf16-valued) -> channels-last bf16 device tensor [B,L,C]
    return torch.from_numpy(np.ascontiguousarray(x_cf.transpose(0, 2, 1))).to(torch.bfloat16).cuda()


def _ulp_bf16(ref):
    """One bf16 ulp at the magnitude of ref (8 significand bits)."""
    return np.maximum(np.abs(ref), 2.0 ** -126) * 2.0 ** -7


# ------------------------------------------------------------------------------------------------
# single layers
# ------------------------------------------------------------------------------------------------
CONV_CASES = [
    # (B, L, C_in, C_out, k, d, act, residual): the V1 shapes of every tile configuration, plus ragged lengths
    (1, 300, 32, 32, 3, 1, 1, True), (2, 517, 32, 32, 11, 5, 1, False), (1, 1100, 32, 32, 7, 3, 1, True),
    (1, 200, 64, 64, 11, 5, 1, True), (2, 531, 64, 64, 3, 3, 1, False), (1, 260, 64, 64, 7, 1, 1, True),
    (1, 130, 128, 128, 7, 5, 1, True), (2, 259, 128, 128, 3, 1, 1, False),
    (1, 70, 256, 256, 11, 3, 1, True), (1, 264, 256, 256, 3, 1, 1, False),
    (1, 5, 32, 32, 11, 5, 1, True), (1, 1, 64, 64, 3, 1, 1, False), (3, 41, 24, 40, 5, 2, 1, False),
    (1, 9000, 256, 256, 3, 1, 1, True),    # 71 window items x 2 C_out blocks: the XCD-grouped block mapping, ragged
    (37, 300, 256, 256, 3, 1, 1, True),    # round 4: 3 window items per batch item x 37 items = 111: grouped only with the batch index
                                           # folded into the item index (Launch::fold_b); 111 is not a multiple of 8
]


@pytest.mark.parametrize("B,L,Ci,Co,k,d,act,use_res", CONV_CASES)
def test_bf16_conv1d_matches_oracle(B, L, Ci, Co, k, d, act, use_res):
    from iris import _native
    lib = _native.load()
    rng = np.random.default_rng(B * 1000 + L + Ci + k + d)
    x = _r16(rng.standard_normal((B, Ci, L)).astype(np.float32))
    w = (rng.standard_normal((Co, Ci, k)) / np.sqrt(Ci * k)).astype(np.float32)
    b = rng.standard_normal(Co).astype(np.float32)
    res = _r16(rng.standard_normal((B, Co, L)).astype(np.float32)) if use_res else None
    xin = _r16(orc.lrelu_np(x, 0.1)) if act else x
    want = orc.conv1d_np(xin, _r16(w), b, d)          # fp64 accumulation of exact bf16 x bf16 products
    if use_res:
        want = want + res
    xd = _bf16_cl(x)
    rd = _bf16_cl(res) if use_res else None
    yd = torch.full((B, L, Co), float("nan"), dtype=torch.bfloat16, device="cuda")
    _native.check("op_conv1d_bf16", lib.iris_hifigan_op_conv1d_bf16(
        xd.data_ptr(), _fp(w), _fp(b), rd.data_ptr() if use_res else None, yd.data_ptr(),
        B, L, Ci, Co, k, d, act, 0.1, None))
    got = yd.float().cpu().numpy().transpose(0, 2, 1)
    assert np.isfinite(got).all()
    err = np.abs(got - want)
    assert (err <= _ulp_bf16(want) * 1.001 + 1e-6).all()            # never more than one bf16 ulp
    exact = np.abs(got - _r16(want.astype(np.float32))) == 0
    assert exact.mean() > 0.99                                       # and almost always the same rounding


@pytest.mark.parametrize("B,L,Ci,Co,k,u", [(1, 40, 512, 256, 16, 8), (2, 130, 256, 128, 16, 8), (1, 300, 128, 64, 4, 2),
                                            (2, 517, 64, 32, 4, 2), (1, 1, 64, 32, 4, 2), (1, 9, 32, 16, 7, 3),
                                            (1, 25000, 64, 32, 4, 2),    # 66 window items x 2 phases: XCD-grouped
                                            # round 4, the GEMM form (convt_mfma_bf16.h): 64 x 256 blocks (>= 2 per CU), a second
                                            # row tile of two rows, nine batch items over the eight XCDs
                                            (4, 2100, 256, 128, 16, 8), (3, 65, 512, 256, 16, 8), (9, 63, 128, 64, 4, 2)])
def test_bf16_conv_transpose1d_matches_oracle(B, L, Ci, Co, k, u):
    from iris import _native
    lib = _native.load()
    rng = np.random.default_rng(L + Ci + k)
    x = _r16(rng.standard_normal((B, Ci, L)).astype(np.float32))
    w = (rng.standard_normal((Ci, Co, k)) / np.sqrt(Ci * k / u)).astype(np.float32)
    b = rng.standard_normal(Co).astype(np.float32)
    want = orc.conv_transpose1d_np(_r16(orc.lrelu_np(x, 0.1)), _r16(w), b, u, (k - u) // 2)
    xd = _bf16_cl(x)
    yd = torch.full((B, L * u, Co), float("nan"), dtype=torch.bfloat16, device="cuda")
    _native.check("op_conv_transpose1d_bf16", lib.iris_hifigan_op_conv_transpose1d_bf16(
        xd.data_ptr(), _fp(w), _fp(b), yd.data_ptr(), B, L, Ci, Co, k, u, 1, 0.1, None))
    got = yd.float().cpu().numpy().transpose(0, 2, 1)
    assert got.shape == want.shape and np.isfinite(got).all()
    assert (np.abs(got - want) <= _ulp_bf16(want) * 1.001 + 1e-6).all()
    assert (np.abs(got - _r16(want.astype(np.float32))) == 0).mean() > 0.99


# ------------------------------------------------------------------------------------------------
# the fused ResBlock conv pair (mrf_pair_bf16_kernel): conv1 -> LDS -> conv2 + residual in one launch
# ------------------------------------------------------------------------------------------------
PAIR_CASES = [
    # (B, L, C, dils): tile edges of the k = 3 / 7 / 11 branches (T_OUT = M - (k - 1), M = 384 at C = 32, 192 at C = 64),
    # one-row and shorter-than-halo inputs, several tiles with a ragged tail, a multi-XCD job range
    (1, 1, 32, (1, 1, 1)), (2, 9, 32, (5, 5, 5)), (1, 373, 32, (3, 3, 3)), (1, 374, 32, (5, 5, 5)), (1, 375, 32, (1, 1, 1)),
    (2, 1531, 32, (5, 5, 5)), (1, 20000, 32, (3, 3, 3)),
    (1, 7, 64, (5, 5, 5)), (1, 181, 64, (1, 1, 1)), (1, 182, 64, (3, 3, 3)), (2, 777, 64, (5, 5, 5)), (1, 9000, 64, (1, 1, 1)),
    (1, 3, 128, (1, 1, 1)), (1, 118, 128, (5, 5, 5)), (2, 119, 128, (3, 3, 3)), (1, 1000, 128, (5, 5, 5)), (1, 4100, 128, (1, 1, 1)),
]


@pytest.mark.parametrize("B,L,C,dils", PAIR_CASES)
def test_bf16_fused_pair_equals_separate_layers_and_oracle(B, L, C, dils):
    """y_j = conv2_j(lrelu(conv1_j(lrelu(x_j)))) + x_j for the three branches in ONE launch: bit for bit what the two
    separate bf16 layers produce (same MFMA order and rounding points; only HBM traffic differs), and within one
    bf16 ulp of the numpy restatement with those rounding points (hifigan_pretrained.py:64-71)."""
    from iris import _native
    lib = _native.load()
    rng = np.random.default_rng(C + L)
    ks = (3, 7, 11)
    xs = [_r16(rng.standard_normal((B, C, L)).astype(np.float32)) for _ in ks]
    w1 = [(rng.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32) for k in ks]
    w2 = [(rng.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32) for k in ks]
    b1 = [rng.standard_normal(C).astype(np.float32) for _ in ks]
    b2 = [rng.standard_normal(C).astype(np.float32) for _ in ks]
    xd = [_bf16_cl(x) for x in xs]
    yd = [torch.full((B, L, C), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in ks]
    vp3, fp3 = ctypes.c_void_p * 3, ctypes.POINTER(ctypes.c_float) * 3
    _native.check("op_mrf_pair_bf16", lib.iris_hifigan_op_mrf_pair_bf16(
        vp3(*[t.data_ptr() for t in xd]), fp3(*[_fp(w) for w in w1]), fp3(*[_fp(b) for b in b1]),
        fp3(*[_fp(w) for w in w2]), fp3(*[_fp(b) for b in b2]), vp3(*[t.data_ptr() for t in yd]),
        3, B, L, C, (ctypes.c_int32 * 3)(*ks), (ctypes.c_int32 * 3)(*dils), 0.1, None))
    for j, k in enumerate(ks):
        xt = torch.full((B, L, C), float("nan"), dtype=torch.bfloat16, device="cuda")
        y2 = torch.full((B, L, C), float("nan"), dtype=torch.bfloat16, device="cuda")
        _native.check("op_conv1d_bf16", lib.iris_hifigan_op_conv1d_bf16(
            xd[j].data_ptr(), _fp(w1[j]), _fp(b1[j]), None, xt.data_ptr(), B, L, C, C, k, dils[j], 1, 0.1, None))
        _native.check("op_conv1d_bf16", lib.iris_hifigan_op_conv1d_bf16(
            xt.data_ptr(), _fp(w2[j]), _fp(b2[j]), xd[j].data_ptr(), y2.data_ptr(), B, L, C, C, k, 1, 1, 0.1, None))
        assert torch.isfinite(yd[j].float()).all(), j
        assert torch.equal(yd[j], y2), (j, int((yd[j] != y2).sum()))
        if L <= 2000:     # the fp64 loops of the numpy restatement are slow
            xt_ref = _r16(orc.conv1d_np(_r16(orc.lrelu_np(xs[j], 0.1)), _r16(w1[j]), b1[j], dils[j]).astype(np.float32))
            want = orc.conv1d_np(_r16(orc.lrelu_np(xt_ref, 0.1)), _r16(w2[j]), b2[j], 1) + xs[j]
            got = yd[j].float().cpu().numpy().transpose(0, 2, 1)
            # (a one-ulp difference in xt propagates: the bound is loose by design, the bit-equality above is the test)
            assert np.abs(got - want).max() <= 0.05 * max(1.0, np.abs(want).max())
            assert (np.abs(got - _r16(want.astype(np.float32))) == 0).mean() > 0.9


MEAN_PAIR_CASES = [
    # (B, L, C, dils): tile edges (T_OUT = M - 10 for all three branches: M = 384 at C = 32, 192 at C = 64), short inputs, ragged tails
    (1, 1, 32, (1, 1, 1)), (2, 9, 32, (5, 5, 5)), (1, 373, 32, (3, 3, 3)), (1, 374, 32, (5, 5, 5)), (1, 375, 32, (1, 1, 1)),
    (2, 1531, 32, (5, 5, 5)), (1, 20000, 32, (3, 3, 3)),
    (1, 7, 64, (5, 5, 5)), (1, 181, 64, (1, 1, 1)), (1, 182, 64, (3, 3, 3)), (1, 183, 64, (5, 5, 5)), (2, 777, 64, (5, 5, 5)),
    (1, 9000, 64, (1, 1, 1)),
]


@pytest.mark.parametrize("mean_f32", [0, 1])
@pytest.mark.parametrize("B,L,C,dils", MEAN_PAIR_CASES)
def test_bf16_summing_pair_equals_three_tensor_path(B, L, C, dils, mean_f32):
    """The stage's last pair on the summing kernel (hifigan_pretrained.py:64-71, 131-137): ONE output that must be, bit for
    bit, what the consumer would build from the three y_j of the per-branch kernel -- bf16(LeakyReLU(((y0 + y1) + y2) * fp32(1/3)))
    for the next ConvTranspose1d, or that mean in fp32 for conv_post."""
    from iris import _native
    lib = _native.load()
    rng = np.random.default_rng(C + L + 17)
    ks = (3, 7, 11)
    xs = [_r16(rng.standard_normal((B, C, L)).astype(np.float32)) for _ in ks]
    w1 = [(rng.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32) for k in ks]
    w2 = [(rng.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32) for k in ks]
    b1 = [rng.standard_normal(C).astype(np.float32) for _ in ks]
    b2 = [rng.standard_normal(C).astype(np.float32) for _ in ks]
    xd = [_bf16_cl(x) for x in xs]
    yd = [torch.full((B, L, C), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in ks]
    vp3, fp3 = ctypes.c_void_p * 3, ctypes.POINTER(ctypes.c_float) * 3
    args = (fp3(*[_fp(w) for w in w1]), fp3(*[_fp(b) for b in b1]), fp3(*[_fp(w) for w in w2]), fp3(*[_fp(b) for b in b2]))
    _native.check("op_mrf_pair_bf16", lib.iris_hifigan_op_mrf_pair_bf16(
        vp3(*[t.data_ptr() for t in xd]), *args, vp3(*[t.data_ptr() for t in yd]),
        3, B, L, C, (ctypes.c_int32 * 3)(*ks), (ctypes.c_int32 * 3)(*dils), 0.1, None))
    md = torch.full((B, L, C), float("nan"), dtype=torch.float32 if mean_f32 else torch.bfloat16, device="cuda")
    _native.check("op_mrf_pair_mean_bf16", lib.iris_hifigan_op_mrf_pair_mean_bf16(
        vp3(*[t.data_ptr() for t in xd]), *args, md.data_ptr(), mean_f32,
        B, L, C, (ctypes.c_int32 * 3)(*ks), (ctypes.c_int32 * 3)(*dils), 0.1, None))
    y = [t.float().cpu() for t in yd]
    mean = (((y[0] + y[1]) + y[2]) + 0.0) * torch.tensor(1.0 / 3.0, dtype=torch.float32)
    if mean_f32:
        want = mean
    else:
        want = torch.where(mean > 0, mean, mean * torch.tensor(0.1, dtype=torch.float32)).to(torch.bfloat16)
    got = md.cpu()
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, want), int((got != want).sum())


def test_bf16_summing_pair_rejects_other_channel_counts():
    from iris import _native
    lib = _native.load()
    z = torch.zeros((1, 8, 128), dtype=torch.bfloat16, device="cuda")
    m = torch.zeros((1, 8, 128), dtype=torch.bfloat16, device="cuda")
    w = np.zeros((128, 128, 3), np.float32)
    b = np.zeros(128, np.float32)
    vp3, fp3 = ctypes.c_void_p * 3, ctypes.POINTER(ctypes.c_float) * 3
    rc = lib.iris_hifigan_op_mrf_pair_mean_bf16(vp3(*[z.data_ptr()] * 3), fp3(*[_fp(w)] * 3), fp3(*[_fp(b)] * 3), fp3(*[_fp(w)] * 3),
                                                fp3(*[_fp(b)] * 3), m.data_ptr(), 0, 1, 8, 128, (ctypes.c_int32 * 3)(3, 3, 3),
                                                (ctypes.c_int32 * 3)(1, 1, 1), 0.1, None)
    assert rc == 4      # IRIS_HIFIGAN_UNSUPPORTED


def test_bf16_fused_pair_rejects_other_channel_counts():
    from iris import _native
    lib = _native.load()
    z = torch.zeros((1, 8, 256), dtype=torch.bfloat16, device="cuda")
    w = np.zeros((256, 256, 3), np.float32)
    b = np.zeros(256, np.float32)
    vp1, fp1 = ctypes.c_void_p * 1, ctypes.POINTER(ctypes.c_float) * 1
    rc = lib.iris_hifigan_op_mrf_pair_bf16(vp1(z.data_ptr()), fp1(_fp(w)), fp1(_fp(b)), fp1(_fp(w)), fp1(_fp(b)), vp1(z.data_ptr()),
                                           1, 1, 8, 256, (ctypes.c_int32 * 1)(3), (ctypes.c_int32 * 1)(1), 0.1, None)
    assert rc == 4      # IRIS_HIFIGAN_UNSUPPORTED


# ------------------------------------------------------------------------------------------------
# whole generator
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T,seed,log_mel", [(1, 100, 1001, False), (3, 57, 5, True), (1, 1, 9, False), (5, 2, 10, False),
                                               (2, 300, 4, True)])
def test_bf16_generator_close_to_fp32_and_to_restatement(B, T, seed, log_mel, engine, dev):
    from iris._weights import seeded_mel
    eng, sd = engine
    mel = seeded_mel(seed, B, T, log_mel=log_mel)
    got = eng.forward(torch.from_numpy(mel).to(dev), dtype="bf16").cpu().numpy()
    folded = orc.to_torch_folded(sd)
    ref32 = orc.generator_forward_torch(folded, mel).numpy()[:, 0, :]      # the pinned fp32 oracle
    emu16 = orc.generator_forward_bf16(folded, mel).numpy()[:, 0, :]       # same rounding points on the CPU
    assert got.shape == (B, 256 * T) and np.isfinite(got).all() and np.abs(got).max() <= 1.0
    for want in (ref32, emu16):
        d = np.abs(got - want)
        assert d.max() <= TOL_BF16_MAX and d.mean() <= TOL_BF16_MEAN
    # the CPU restatement sits at the same distance from fp32 as the HIP path does (same noise source)
    assert np.abs(emu16 - ref32).mean() <= TOL_BF16_MEAN


def test_bf16_intermediates_match_restatement(engine, dev):
    """Stage by stage (iris_hifigan_forward_until): the upsample output and the MRF mean of every stage against the
    CPU restatement's taps -- the C = 64 / 32 stages run the fused conv-pair kernel, whose output alternates between
    the branch's two workspace buffers."""
    from iris._weights import seeded_mel
    eng, sd = engine
    cfg = eng.cfg
    mel_np = seeded_mel(12, 2, 40, log_mel=True)
    taps = {}
    orc.generator_forward_bf16(orc.to_torch_folded(sd), mel_np, taps=taps)
    mel = torch.from_numpy(mel_np).to(dev)
    last = 2 * len(cfg.resblock_dilation_sizes[0]) - 1
    inv_n = np.float32(1.0 / cfg.num_kernels)
    for i in range(cfg.num_upsamples):
        got = eng.forward_until(mel, i, last, dtype="bf16")
        assert not got["mean_in_y0"]
        up = taps[f"ups.{i}"].numpy()
        # (rounding differences of earlier stages propagate: only the first stage's input is bit-identical on both sides)
        assert np.abs(got["up"] - up).max() <= 0.03 * np.abs(up).max() and (got["up"] == up).mean() > (0.9 if i == 0 else 0.3), i
        y = got["y"]
        mrf = ((y[0] + y[1]) + y[2]) * inv_n
        want = taps[f"mrf.{i}"].numpy()
        d = np.abs(mrf - want)
        assert d.max() <= 0.03 * np.abs(want).max() and d.mean() <= 2e-3 * np.abs(want).max(), (i, d.max(), d.mean())
    # mid-pair stop: xt of the first pair of the last stage (that pair then runs as two launches)
    i = cfg.num_upsamples - 1
    got = eng.forward_until(mel, i, 0, dtype="bf16")
    x = torch.from_numpy(got["up"])              # the GPU's own stage input: isolates the one conv
    import torch.nn.functional as F
    r16 = lambda t: t.to(torch.bfloat16).to(torch.float32)
    folded = orc.to_torch_folded(sd)
    for j, k in enumerate(cfg.resblock_kernel_sizes):
        pfx = f"resblocks.{i * cfg.num_kernels + j}.convs1.0"
        d0 = cfg.resblock_dilation_sizes[j][0]
        want = r16(F.conv1d(r16(F.leaky_relu(x, 0.1)), r16(folded[pfx + ".weight"]), folded[pfx + ".bias"], dilation=d0,
                            padding=(k * d0 - d0) // 2)).numpy()
        assert np.abs(got["xt"][j] - want).max() <= 0.02 * np.abs(want).max() and (got["xt"][j] == want).mean() > 0.9, j


def test_bf16_is_deterministic_and_batch_independent(engine, dev):
    from iris._weights import seeded_mel
    eng, _ = engine
    mel = torch.from_numpy(seeded_mel(77, 4, 40)).to(dev)
    full = eng.forward(mel, dtype="bf16").clone()
    assert torch.equal(eng.forward(mel, dtype="bf16"), full)
    for b in range(4):
        assert torch.equal(eng.forward(mel[b:b + 1].contiguous(), dtype="bf16")[0], full[b])
    # several tiles per branch in the fused C = 32 / 64 stages (tile = 374 / 182 rows): neighbouring blocks read each
    # other's halo rows while others write -- repeated runs must agree bit for bit
    mel = torch.from_numpy(seeded_mel(78, 3, 700, log_mel=True)).to(dev)
    full = eng.forward(mel, dtype="bf16").clone()
    for _ in range(3):
        assert torch.equal(eng.forward(mel, dtype="bf16"), full)
    assert torch.equal(eng.forward(mel[1:2].contiguous(), dtype="bf16")[0], full[1])


def test_bf16_empty_inputs_and_bad_dtype(engine, dev):
    eng, _ = engine
    assert eng.forward(torch.empty((0, 80, 10), device=dev), dtype="bf16").shape == (0, 2560)
    assert eng.forward(torch.empty((2, 80, 0), device=dev), dtype="bf16").shape == (2, 0)
    with pytest.raises(ValueError):
        eng.forward(torch.zeros((1, 80, 4), device=dev), dtype="fp8")
    assert eng.workspace_bytes(2, 50, "bf16") * 2 <= eng.workspace_bytes(2, 50, "f32") + 4096


def test_bf16_profile_records_cover_algorithmic_work(engine, dev):
    from iris._engine import algorithmic_work
    from iris._weights import seeded_mel
    eng, _ = engine
    eng.set_profiling(True)
    B, T = 2, 50
    eng.forward(torch.from_numpy(seeded_mel(1, B, T)).to(dev), dtype="bf16")
    torch.cuda.synchronize()
    recs = eng.read_profile()
    eng.set_profiling(False)
    # 1 + 4 * (1 + 6) + 1 layers; the conv pairs of the C = 128 / 64 / 32 stages run fused (one launch per pair)
    assert len(recs) == 30 - 3 * 3
    work = algorithmic_work(eng.cfg)
    assert sum(r["flops"] for r in recs) == pytest.approx(work["flop_per_frame"] * B * T, rel=1e-12)
    # accounting L at 2 bytes per element; the mel (read) and the waveform (written) stay fp32; biases stay fp32
    n_bias = sum(s.c_out for s in __import__("iris._weights", fromlist=["layer_specs"]).layer_specs(eng.cfg))
    want = (2.0 * work["elements_per_frame"] * B * T + 2.0 * B * T * (eng.cfg.in_channels + eng.hop_length)
            + 2.0 * (work["weight_values"] - n_bias) + 4.0 * n_bias
            + 2.0 * (eng.cfg.stage_channels(eng.cfg.num_upsamples - 1) * eng.cfg.post_kernel_size))   # conv_post weights fp32
    assert sum(r["bytes"] for r in recs) == pytest.approx(want, rel=1e-9)
    assert all(r["ms"] > 0 for r in recs)
    eng.set_profiling(2)                     # grouped: one record per stage's MRF launches (6 separate or 3 fused)
    eng.forward(torch.from_numpy(seeded_mel(1, B, T)).to(dev), dtype="bf16")
    torch.cuda.synchronize()
    grouped = eng.read_profile()
    eng.set_profiling(False)
    assert len(grouped) == 10 and [g["launches"] for g in grouped if g["kind"] == "mrf_resblock_conv"] == [6, 3, 3, 3]
    assert sum(g["bytes"] for g in grouped) == pytest.approx(want, rel=1e-9)


def test_bf16_config3_full_size_properties(engine, dev):
    """configs[2]: batch 32 x 80 x 500 frames, bf16.  Too big for the CPU oracle in a test: checked through
    determinism, batch independence (bit-exact) and two items against the fp32 oracle."""
    from iris._weights import seeded_mel
    eng, sd = engine
    mel_np = seeded_mel(1003, 32, 500, log_mel=True)
    mel = torch.from_numpy(mel_np).to(dev)
    full = eng.forward(mel, dtype="bf16").clone()
    assert full.shape == (32, 128000) and torch.isfinite(full).all() and full.abs().max() <= 1.0
    for _ in range(3):
        assert torch.equal(eng.forward(mel, dtype="bf16"), full)
    for b in (0, 13, 31):
        assert torch.equal(eng.forward(mel[b:b + 1].contiguous(), dtype="bf16")[0], full[b])
    folded = orc.to_torch_folded(sd)
    for b in (5, 30):
        want = orc.generator_forward_torch(folded, mel_np[b:b + 1]).numpy()[0, 0]
        d = np.abs(full[b].cpu().numpy() - want)
        assert d.max() <= TOL_BF16_MAX and d.mean() <= TOL_BF16_MEAN


def test_bf16_hipgraph_replay_matches_eager(engine, dev):
    from iris._weights import seeded_mel
    eng, _ = engine
    mel = torch.from_numpy(seeded_mel(5, 2, 64)).to(dev)
    eager = eng.forward(mel, dtype="bf16").clone()
    assert torch.equal(eng.forward_graph(mel, dtype="bf16"), eager)
    assert torch.equal(eng.forward_graph(mel, dtype="bf16"), eager)


def test_default_dtype_switch(engine, dev, monkeypatch):
    """IRIS_VOCODER_DTYPE / GeneratorEngine(dtype=...) select what forward() computes in when no dtype is passed:
    that is how the drop-in wrappers (which know nothing about dtypes) are switched to bf16."""
    from iris._engine import GeneratorEngine
    from iris._weights import seeded_mel
    eng, sd = engine
    mel = torch.from_numpy(seeded_mel(3, 1, 30)).to(dev)
    want16 = eng.forward(mel, dtype="bf16").clone()
    want32 = eng.forward(mel).clone()
    assert not torch.equal(want16, want32)
    monkeypatch.setenv("IRIS_VOCODER_DTYPE", "bf16")
    e2 = GeneratorEngine(eng.cfg, sd, dev)
    assert e2.default_dtype == "bf16" and torch.equal(e2.forward(mel), want16)
    assert torch.equal(e2.forward(mel, dtype="f32"), want32)
    e2.close()
    monkeypatch.setenv("IRIS_VOCODER_DTYPE", "int4")
    with pytest.raises(ValueError):
        GeneratorEngine(eng.cfg, sd, dev)


def test_bf16_generic_config_and_unsupported_config(dev, case_setup):
    """A non-V1 generator (2 MRF kernels of sizes 3/5, rates 4/2/3, channels 64 -> 32/16/8) runs through the same
    bf16 kernels; one whose channel counts are not multiples of 8 is refused with UNSUPPORTED, not mis-computed."""
    from conftest import oracle_config
    from iris import _native
    from iris._engine import GeneratorEngine
    from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
    cfg = GeneratorConfig(in_channels=16, upsample_rates=(4, 2, 3), upsample_kernel_sizes=(8, 4, 9),
                          upsample_initial_channel=64, resblock_kernel_sizes=(3, 5),
                          resblock_dilation_sizes=((1, 2), (2, 6)))
    sd = seeded_state_dict(cfg, seed=7, gain=1.3, post_gain=6.0)
    eng = GeneratorEngine(cfg, sd, dev)
    rng = np.random.default_rng(3)
    mel = rng.standard_normal((3, 16, 37)).astype(np.float32)
    got = eng.forward(torch.from_numpy(mel).to(dev), dtype="bf16").cpu().numpy()
    want = orc.generator_forward_torch(orc.to_torch_folded(sd), mel, oracle_config(cfg)).numpy()[:, 0, :]
    assert got.shape == want.shape == (3, 37 * 24)
    d = np.abs(got - want)
    assert np.isfinite(got).all() and d.max() <= TOL_BF16_MAX and d.mean() <= TOL_BF16_MEAN
    assert np.abs(eng.forward(torch.from_numpy(mel).to(dev), dtype="f32").cpu().numpy() - want).max() <= 1e-4
    eng.close()

    cfg2, sd2 = case_setup("small_cfg_B3_T19")      # channels 48 -> 24 / 12 / 6
    eng2 = GeneratorEngine(cfg2, sd2, dev)
    with pytest.raises(_native.NativeCallError) as exc:
        eng2.forward(torch.zeros((1, cfg2.in_channels, 5), device=dev), dtype="bf16")
    assert "multiples of 8" in str(exc.value)
    assert eng2.forward(torch.zeros((1, cfg2.in_channels, 5), device=dev)).shape == (1, 5 * cfg2.hop_length)
    eng2.close()
