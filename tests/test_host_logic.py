"""CPU tests of the host side: weight-norm fold, blob order, checkpoint containers, wrapper
semantics, C-ABI export table.  No GPU, no compute calls into the HIP library."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import REPO

from iris import _native
from iris._weights import (GeneratorConfig, expected_weight_count, extract_state_dict, fold_weight_norm,
                           folded_layers, keras_to_reference_layout, layer_specs, reference_to_keras_layout,
                           seeded_mel, seeded_state_dict, state_dict_keys, weight_blob)


def test_v1_layer_inventory_matches_reference(manifest):
    cfg = GeneratorConfig()
    specs = layer_specs(cfg)
    assert len(specs) == 78                                   # 78 conv layers
    assert sorted(state_dict_keys(cfg)) == manifest["state_dict_keys"]   # the reference's 234 keys
    assert expected_weight_count(cfg) == 13_926_017            # effective weights + biases
    assert cfg.hop_length == 256


def test_fold_matches_reference_weight_norm(manifest, golden, case_setup):
    """Effective weights as the reference's parametrised modules expose them (``module.weight``)."""
    cfg, sd = case_setup("v1_default_T4_taps")
    stats = manifest["folded_weight_stats"]
    folded = {s.name: w for s, w, _ in folded_layers(cfg, sd)}
    assert set(folded) == set(stats)
    for name, (s1, s2) in stats.items():
        w = folded[name].astype(np.float64)
        assert w.sum() == pytest.approx(s1, rel=1e-5, abs=1e-5)
        assert (w ** 2).sum() == pytest.approx(s2, rel=1e-5)
    g = golden("v1_default_folded_weights")
    for key, ref in g.items():
        name = {"conv_post": "conv_post", "ups_3": "ups.3", "resblocks_9_convs1_0": "resblocks.9.convs1.0",
                "resblocks_11_convs2_2": "resblocks.11.convs2.2"}[key]
        assert np.abs(folded[name] - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max())


def test_fold_norm_axis_for_transposed_conv():
    """dim 0 is C_in for ConvTranspose1d: the norm runs over (C_out, k) per INPUT channel."""
    rng = np.random.default_rng(0)
    v = rng.standard_normal((6, 4, 5)).astype(np.float32)
    g = rng.uniform(0.5, 2, (6, 1, 1)).astype(np.float32)
    w = fold_weight_norm(g, v)
    np.testing.assert_allclose(np.sqrt((w.astype(np.float64) ** 2).sum(axis=(1, 2))), g[:, 0, 0], rtol=1e-6)
    t = torch._weight_norm(torch.from_numpy(v), torch.from_numpy(g), 0).numpy()
    assert np.abs(w - t).max() <= 2e-7


def test_blob_order_and_plain_weights_roundtrip():
    cfg = GeneratorConfig(in_channels=8, upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4),
                          upsample_initial_channel=16, resblock_kernel_sizes=(3, 5),
                          resblock_dilation_sizes=((1, 2), (1, 3)))
    sd = seeded_state_dict(cfg, seed=1)
    blob = weight_blob(cfg, sd)
    assert blob.dtype == np.float32 and blob.size == expected_weight_count(cfg)
    # walk the blob in the documented order (include/iris_hifigan.h) and compare with the folds
    off = 0
    for spec, w, b in folded_layers(cfg, sd):
        n = w.size
        np.testing.assert_array_equal(blob[off:off + n].reshape(w.shape), w)
        off += n
        np.testing.assert_array_equal(blob[off:off + b.size], b)
        off += b.size
    assert off == blob.size
    names = [s.name for s in layer_specs(cfg)]
    assert names[:6] == ["conv_pre", "ups.0", "resblocks.0.convs1.0", "resblocks.0.convs1.1",
                         "resblocks.0.convs2.0", "resblocks.0.convs2.1"]
    # plain (already folded) weights are accepted too -- the Keras twin has no weight-norm
    plain = {}
    for spec, w, b in folded_layers(cfg, sd):
        plain[f"{spec.name}.weight"], plain[f"{spec.name}.bias"] = w, b
    np.testing.assert_array_equal(weight_blob(cfg, plain), blob)
    with pytest.raises(KeyError):
        weight_blob(cfg, {k: v for k, v in plain.items() if not k.startswith("ups.1")})
    bad = dict(plain)
    bad["conv_pre.weight"] = bad["conv_pre.weight"][:, :, :3]
    with pytest.raises(ValueError):
        weight_blob(cfg, bad)


def test_keras_layout_map():
    spec_c = [s for s in layer_specs(GeneratorConfig()) if s.name == "resblocks.0.convs1.1"][0]
    spec_t = [s for s in layer_specs(GeneratorConfig()) if s.name == "ups.2"][0]
    rng = np.random.default_rng(1)
    kc = rng.standard_normal((3, 256, 256)).astype(np.float32)       # Conv1D [k, C_in, C_out]
    kt = rng.standard_normal((4, 64, 128)).astype(np.float32)        # Conv1DTranspose [k, C_out, C_in]
    wc, wt = keras_to_reference_layout(spec_c, kc), keras_to_reference_layout(spec_t, kt)
    assert wc.shape == spec_c.weight_shape == (256, 256, 3) and wt.shape == spec_t.weight_shape == (128, 64, 4)
    assert wc[5, 7, 2] == kc[2, 7, 5] and wt[100, 33, 1] == kt[1, 33, 100]
    np.testing.assert_array_equal(reference_to_keras_layout(spec_c, wc), kc)


def test_extract_state_dict_variants():
    sd = {"a": 1}
    for key in ("generator", "model", "state_dict"):
        assert extract_state_dict({key: sd, "other": 0}) is sd
    assert extract_state_dict(sd) is sd
    # precedence as in the reference (hifigan_pretrained.py:174-182): generator > model > state_dict
    assert extract_state_dict({"state_dict": 3, "model": 2, "generator": sd}) is sd
    with pytest.raises(ValueError, match="Unexpected checkpoint format"):
        extract_state_dict([1, 2])


def test_seeded_generators_are_deterministic():
    a, b = seeded_state_dict(seed=5), seeded_state_dict(seed=5)
    assert all(np.array_equal(a[k], b[k]) for k in a) and len(a) == 234
    assert not np.array_equal(a["conv_pre.weight_v"], seeded_state_dict(seed=6)["conv_pre.weight_v"])
    m = seeded_mel(1002, 1, 1000)
    assert m.shape == (1, 80, 1000) and m.dtype == np.float32
    lm = seeded_mel(3, 2, 10, log_mel=True)
    assert lm.min() >= np.float32(np.log(1e-5)) and lm.max() <= 2.0


# ---- drop-in modules (no GPU: construction, state dict, errors) ------------------------------
def test_hifigan_model_state_dict_is_reference_compatible(manifest):
    from iris.hifigan_pretrained import HiFiGANModel
    m = HiFiGANModel()
    assert sorted(m.state_dict().keys()) == manifest["state_dict_keys"]
    sd = seeded_state_dict(seed=2024)
    res = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    np.testing.assert_array_equal(m.state_dict()["ups.1.weight_g"].numpy(), sd["ups.1.weight_g"])
    assert m.ups[0].weight_v.shape == (512, 256, 16) and m.resblocks[11].convs2[2].weight_v.shape == (32, 32, 11)
    small = HiFiGANModel(upsample_initial_channel=32)
    assert small.conv_post.weight_v.shape == (1, 2, 7)
    assert not m.training


def test_generator_wrapper_errors_without_gpu(tmp_path, manifest):
    from iris import hifigan_pretrained as hp
    with pytest.raises(FileNotFoundError, match="Checkpoint not found"):
        hp.HiFiGANGenerator(tmp_path / "missing.ckpt")
    assert manifest["wrapper_semantics"]["missing_error"].startswith("FileNotFoundError: Checkpoint not found")
    bad = tmp_path / "list.ckpt"
    torch.save([1, 2, 3], bad)
    with pytest.raises(ValueError, match="Unexpected checkpoint format"):
        hp.HiFiGANGenerator(bad)
    with pytest.raises(FileNotFoundError):
        hp.get_pretrained_hifigan()          # the default path does not exist (no checkpoint ships)
    assert "models--speechbrain--tts-hifigan-ljspeech" in str(hp.default_checkpoint_path())
    if not torch.cuda.is_available():
        ck = tmp_path / "generator.ckpt"
        torch.save({"generator": {k: torch.from_numpy(v) for k, v in seeded_state_dict().items()}}, ck)
        with pytest.raises(RuntimeError, match="no HIP device"):
            hp.HiFiGANGenerator(ck)          # loud failure, never a CPU fallback


def test_keras_twin_construction_and_weight_files(tmp_path):
    from iris import vocoder
    g = vocoder.HiFiGANGenerator(seed=0)
    assert g.count_params() == 13_926_017
    assert g.weights["conv_pre.kernel"].shape == (7, 80, 512)
    assert g.weights["ups.0.kernel"].shape == (16, 256, 512)         # Conv1DTranspose [k, C_out, C_in]
    assert not g.weights["conv_post.bias"].any()
    lines = []
    g.summary(print_fn=lines.append)
    assert "13,926,017" in lines[-1]
    path = tmp_path / "w.npz"
    g.save_weights(str(path))
    g2 = vocoder.HiFiGANGenerator(seed=1)
    g2.load_weights(str(path))
    np.testing.assert_array_equal(g2.weights["ups.2.kernel"], g.weights["ups.2.kernel"])
    sd = g.reference_state_dict()
    assert sd["ups.0.weight"].shape == (512, 256, 16) and sd["conv_pre.weight"].shape == (512, 80, 7)
    with pytest.raises(NotImplementedError):
        g.load_weights(str(tmp_path / "w.weights.h5"))
    if not torch.cuda.is_available():
        voc = vocoder.create_vocoder(str(tmp_path / "does_not_exist.keras"))   # no error, random weights
        with pytest.raises(RuntimeError, match="no HIP device"):
            voc.infer(np.zeros((80, 4)))


# ---- C-ABI ------------------------------------------------------------------------------------
def test_cabi_exports_every_declared_symbol():
    header = (REPO / "include" / "iris_hifigan.h").read_text()
    declared = set(re.findall(r"\b(iris_(?:hifigan|postnet)_[a-z0-9_]+)\s*\(", header))
    assert {"iris_postnet_create", "iris_postnet_forward", "iris_hifigan_forward"} <= declared
    assert declared == set(_native.SYMBOLS), declared ^ set(_native.SYMBOLS)
    lib = _native.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.iris_hifigan_abi_version() == _native.ABI_VERSION == 4
    assert ctypes.sizeof(_native.Config) == 4 * (3 + 8 + 8 + 1 + 8 + 8 + 64 + 2) + 4
    assert ctypes.sizeof(_native.LaunchRecord) == 40
    assert ctypes.sizeof(_native.WorkspaceMap) == 8 * (2 + 8 + 8 + 1) + 8
    assert ctypes.sizeof(_native.PlanLaunch) == 80 + 4 * 4 + 8 and ctypes.sizeof(_native.Plan) == 24 + 96 * 104


def test_cabi_argument_validation_without_gpu():
    """Pure host-side checks of the library (no kernel launch, no device needed)."""
    lib = _native.load()
    cfg = GeneratorConfig()
    n = ctypes.c_uint64()
    assert lib.iris_hifigan_weight_count(ctypes.byref(_native.make_config(cfg)), ctypes.byref(n)) == 0
    assert n.value == expected_weight_count(cfg)
    small = GeneratorConfig(in_channels=20, upsample_rates=(4, 2, 3), upsample_kernel_sizes=(8, 4, 9),
                            upsample_initial_channel=48, resblock_kernel_sizes=(3, 5),
                            resblock_dilation_sizes=((1, 2), (2, 6)))
    assert lib.iris_hifigan_weight_count(ctypes.byref(_native.make_config(small)), ctypes.byref(n)) == 0
    assert n.value == expected_weight_count(small)
    bad = _native.make_config(cfg)
    bad.upsample_kernel_sizes[0] = 15            # (k - u) odd: the reference's padding (k-u)//2 would change L_out
    assert lib.iris_hifigan_weight_count(ctypes.byref(bad), ctypes.byref(n)) == 1
    assert b"kernel" in lib.iris_hifigan_last_error()
    bad = _native.make_config(cfg)
    bad.num_upsamples = 9
    assert lib.iris_hifigan_weight_count(ctypes.byref(bad), ctypes.byref(n)) == 1
    with pytest.raises(_native.NativeCallError):
        _native.check("iris_hifigan_forward", lib.iris_hifigan_forward(None, None, 1, 1, None, None, 0, 0, None))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setenv("IRIS_HIFIGAN_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeLibraryError, match="no CPU fallback"):
        _native.load()


def test_product_never_imports_the_oracle():
    for path in (REPO / "iris-tts_amd").rglob("*"):
        if path.suffix in (".py", ".hip", ".h", ".cpp") and path.is_file():
            text = path.read_text()
            assert "import oracle" not in text and "from oracle" not in text and "hifigan_oracle" not in text, path


def test_release_library_reads_no_diagnostic_switch():
    """A shipped library must not change what it computes because of a stray environment variable: every
    diagnostic switch (IRIS_HIFIGAN_*, IRIS_B16_*, IRIS_S3_*, IRIS_MRF_*) is compiled in only with -DIRIS_MRF_DIAG
    (csrc/diag_env.h), so none of their names may occur in the release .so, and it must not import getenv at all."""
    import subprocess
    data = _native.library_path().read_bytes()
    names = set(re.findall(rb"IRIS_(?:HIFIGAN|B16|S3|MRF)_[A-Z0-9_]+", data))
    assert names == set(), f"environment switch names in the release library: {sorted(names)}"
    syms = subprocess.run(["nm", "-D", "--undefined-only", str(_native.library_path())], capture_output=True, text=True)
    if syms.returncode == 0:
        assert not re.search(r"\bgetenv\b", syms.stdout), "the release library imports getenv"
    # the sources name their switches only through the gated macro
    for path in (REPO / "iris-tts_amd" / "csrc").glob("*"):
        if path.suffix in (".h", ".hip") and path.name != "diag_env.h":
            assert "getenv" not in path.read_text(), path


def test_receptive_field_from_config():
    """streaming.receptive_field_frames: exact interval propagation through the layers; 13 frames for V1
    (+-12.64 frames probed on the reference), larger for wider kernels / dilations."""
    from iris.streaming import RECEPTIVE_FIELD_FRAMES, StreamingVocoder, receptive_field_frames
    assert receptive_field_frames(GeneratorConfig()) == RECEPTIVE_FIELD_FRAMES == 13
    wide = GeneratorConfig(resblock_kernel_sizes=(3, 7, 13), resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 9)))
    assert receptive_field_frames(wide) > 13
    small = GeneratorConfig(in_channels=20, upsample_rates=(4, 2, 3), upsample_kernel_sizes=(8, 4, 9),
                            upsample_initial_channel=48, resblock_kernel_sizes=(3, 5), resblock_dilation_sizes=((1, 2), (2, 6)))
    need = receptive_field_frames(small)
    assert need >= 1
    sv = StreamingVocoder(lambda m: m, config=small)                      # hop and halo come from the configuration
    assert sv.hop_length == 24 and sv.halo_frames == need
    with pytest.raises(ValueError):
        StreamingVocoder(lambda m: m, config=wide, halo_frames=13)        # V1's halo is too small for the wide config
    with pytest.raises(ValueError):
        StreamingVocoder(lambda m: m, config=small, hop_length=256)


def test_receptive_field_is_tight_on_the_oracle():
    """For a non-V1 configuration: chunking with the computed halo reproduces the one-shot oracle output, and one
    frame less visibly does not (so the computed value is the minimum, not merely sufficient)."""
    from conftest import oracle_config
    from iris.streaming import StreamingVocoder, plan_chunks, receptive_field_frames
    from oracle import hifigan_oracle as orc
    cfg = GeneratorConfig(in_channels=6, upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=8,
                          resblock_kernel_sizes=(3, 5), resblock_dilation_sizes=((1, 2), (1, 3)))
    sd = seeded_state_dict(cfg, seed=4, gain=1.3, post_gain=3.0)
    folded, ocfg = orc.to_torch_folded(sd), oracle_config(cfg)
    fwd = lambda m: orc.generator_forward_torch(folded, np.ascontiguousarray(m), ocfg).numpy()[:, 0, :]
    need = receptive_field_frames(cfg)
    mel = seeded_mel(3, 1, 90, n_mels=6)
    full = fwd(mel)
    ok = StreamingVocoder(fwd, chunk_frames=16, config=cfg).infer(mel)
    assert np.abs(ok - full).max() <= 2e-7
    hop = cfg.hop_length
    short = np.concatenate([fwd(mel[:, :, c.win_start:c.win_stop])[:, c.emit_slice(hop)]
                            for c in plan_chunks(90, 16, need - 1)], axis=1)
    assert np.abs(short - full).max() > 1e-6     # the outermost taps carry little weight, but they are not nothing


def test_model_repacks_after_in_place_parameter_edits(monkeypatch):
    """HiFiGANModel notices in-place edits of its parameters (tensor version counters) and drops the packed engine;
    invalidate() does it on request.  (A stand-in engine: no GPU in this test.)"""
    from iris import hifigan_pretrained as hp
    built = []

    class FakeEngine:
        def __init__(self, cfg, sd, device):
            built.append(float(sd["conv_pre.bias"][0]))
            self.device = device

        def close(self):
            pass

    monkeypatch.setattr(hp, "GeneratorEngine", FakeEngine)
    monkeypatch.setattr(hp, "require_gpu", lambda: torch.device("cpu"))
    m = hp.HiFiGANModel(upsample_initial_channel=32)
    e1 = m.engine()
    assert m.engine() is e1 and len(built) == 1                      # nothing changed: same engine
    with torch.no_grad():
        m.conv_pre.bias.fill_(0.25)                                   # tracked in-place edit: noticed by itself
    e2 = m.engine()
    assert e2 is not e1 and built[-1] == 0.25
    m.conv_pre.bias.data.fill_(0.5)                                   # `.data` bypasses the version counter by design ...
    assert m.engine() is e2
    m.invalidate()                                                    # ... so such edits are announced explicitly
    assert m.engine() is not e2 and built[-1] == 0.5 and len(built) == 3


def test_model_built_under_inference_mode(monkeypatch):
    """Tensors created under torch.inference_mode() have no version counter (reading ``_version`` raises): such a
    model still builds its engine and keeps it until invalidate()."""
    from iris import hifigan_pretrained as hp

    class FakeEngine:
        def __init__(self, cfg, sd, device):
            self.device = device

        def close(self):
            pass

    monkeypatch.setattr(hp, "GeneratorEngine", FakeEngine)
    monkeypatch.setattr(hp, "require_gpu", lambda: torch.device("cpu"))
    with torch.inference_mode():
        m = hp.HiFiGANModel(upsample_initial_channel=32)
        e1 = m.engine()
        assert m.engine() is e1                                       # second call: the version check must not raise
    assert m.engine() is e1
    m.invalidate()
    assert m.engine() is not e1


def test_launch_plan_of_the_wide_stages_follows_the_timing_model():
    """mrf_plan (csrc/mrf_conv_mfma_f32.h) on the host: between whole rounds of tiles the wide fp32 stages run (tile, branch)
    jobs in snake order (template argument `, 2>`), at exact rounds the tile-serial grid, on short inputs fixed per-branch
    ranges or the small-problem kernel -- the choices the forced-plan sweeps of profiles/r03_plan_sweep/ found best -- and
    every grid stays within two blocks per CU."""
    cfg = GeneratorConfig()

    def wide(frames, batch=1, cu=256):
        plan = _native.describe_plan(cfg, batch, frames, _native.DTYPE_F32, cu_count=cu)
        out = []
        for l in plan["launches"]:
            k = l["kernel"]
            if k.startswith("mrf_conv_mfma_f32_kernel"):
                args = k[k.index("<") + 1:k.rindex(">")].replace(" ", "").split(",")     # (10 arguments; 12 for the 128-row LEAN form)
                out.append({"mode": int(args[9]), "sum": args[8] == "true", "mt": int(args[2]), "grid": l["grid"][0]})
            elif k.startswith("mrf_small"):
                out.append({"mode": "small", "grid": l["grid"][0]})
        return out

    at700, at1000x4, at282, at100 = wide(700), wide(1000, 4), wide(282), wide(100)
    assert all(l["mode"] == 2 and l["grid"] == 512 for l in at700[:12])                 # 350 / 1400 tiles on 512 slots: jobs
    assert at700[0]["mt"] == 1 and at700[6]["mt"] == 2
    assert all(l["mode"] == 0 and l["grid"] in (500, 504, 512) for l in at1000x4[:12])                   # whole rounds: tile-serial,
    assert all(l["mt"] == (2 if l["sum"] else 4) for l in at1000x4[:12])                                 # on 128-row tiles (round 4) ...
    assert at1000x4[5]["sum"] and at1000x4[11]["sum"]                                  # ... whose last step (64 rows) forms the mean itself
    assert all(l["mode"] == 1 for l in at282[:12])                                      # fixed ranges
    assert at100[0]["mode"] == "small" and at100[6]["mode"] == 2
    for frames in (64, 100, 150, 282, 350, 500, 650, 700, 850, 1000, 1400, 2000):
        for batch in (1, 3):
            for l in wide(frames, batch):
                assert 1 <= l["grid"] and (l["mode"] == "small" or l["grid"] <= 512), (frames, batch, l)
    assert all(l["mode"] == "small" or l["grid"] <= 128 for l in wide(700, 1, cu=64))   # a smaller chip: its own slot count


def test_describe_plan_on_the_host():
    """iris_hifigan_describe_plan: the forward's launch plan without a device -- the same code path as a forward
    (argument checks, workspace layout, mrf_plan / pair plans), every launch recorded instead of issued."""
    cfg = GeneratorConfig()
    # configs[1]: 26 launches at 1000 frames (two fused conv pairs per narrow stage + the summing pair), 24 at 100 frames
    p = _native.describe_plan(cfg, 1, 1000, _native.DTYPE_F32)
    kernels = [l["kernel"] for l in p["launches"]]
    assert p["n_launches"] == len(kernels) and kernels[0].startswith("conv_mfma_f32_kernel") and "conv_post" in kernels[-1]
    assert sum(k.startswith("mrf_") for k in kernels) == p["n_launches"] - 6
    assert sum(k.startswith("convt_mfma_f32_kernel") for k in kernels) == 4           # every upsampler as one GEMM launch (round 4)
    assert all(1 <= l["grid"][0] and l["block"] == 256 and l["lds_bytes"] <= 160 * 1024 for l in p["launches"])
    short = _native.describe_plan(cfg, 1, 100, _native.DTYPE_F32)
    assert any(k == "mrf_small_f32_kernel" for k in (l["kernel"] for l in short["launches"]))   # stage 0 on short inputs
    assert short["n_launches"] <= p["n_launches"]
    # workspace figure == what the engine would be asked to provide (bytes per mel frame are constant)
    assert p["workspace_bytes"] == 10 * short["workspace_bytes"] or abs(p["workspace_bytes"] / short["workspace_bytes"] - 10) < 1e-3
    # bf16 configs[2]: fused pairs (persistent at C <= 64), no launch for xt
    b16 = _native.describe_plan(cfg, 32, 500, _native.DTYPE_BF16)
    assert sum("pair" in l["kernel"] for l in b16["launches"]) == 9 and b16["workspace_bytes"] < p["workspace_bytes"] * 16 * 0.51
    # the plans are sized for the chip they are told about: a smaller one changes the persistent kernels' grids (and may
    # change whether a stage's last pair forms the mean itself: whole-tile jobs must fill whole rounds of the chip)
    small_chip = _native.describe_plan(cfg, 1, 1000, _native.DTYPE_F32, cu_count=64)
    assert abs(small_chip["n_launches"] - p["n_launches"]) <= 2 and small_chip["cu_count"] == 64
    assert [l["grid"] for l in small_chip["launches"]] != [l["grid"] for l in p["launches"]]
    assert max(l["grid"][0] for l in small_chip["launches"] if "pf_kernel" in l["kernel"]) <= 64 * 4
    # refusals are statuses, not crashes
    with pytest.raises(_native.NativeCallError):
        _native.describe_plan(cfg, 65536, 10)
    with pytest.raises(_native.NativeCallError):
        _native.describe_plan(cfg, 1, (1 << 30) // 256 + 1)
    with pytest.raises(_native.NativeCallError, match="2\\^31"):
        _native.describe_plan(cfg, 1, 1 << 18, _native.DTYPE_BF16)
    assert _native.describe_plan(cfg, 0, 100)["n_launches"] == 0
    # a large batch runs as passes over sub-batches that share one workspace: BASELINE.json configs[3] on one GPU
    # (256 x 1000 frames) takes 4 passes of <= 65 items in 14.9 GB instead of one pass in 58 GB
    big = _native.describe_plan(cfg, 256, 1000, _native.DTYPE_F32)
    assert big["passes"] == 4 and big["workspace_bytes"] == 65 * p["workspace_bytes"] < 45e9
    assert big["n_launches"] in (4 * p["n_launches"], 4 * p["n_launches"] - 4, 4 * p["n_launches"] - 8)
    assert _native.describe_plan(cfg, 3, 70000)["passes"] == 3          # items longer than a pass: one item per pass


def test_deferred_init_keeps_the_reference_semantics_of_a_partial_checkpoint():
    """The checkpoint loader constructs the module WITHOUT drawing 13.9 M random parameters it is about to overwrite
    (1.2 s -> 2 ms of cold start); what load_state_dict(strict=False) leaves untouched must still end up with a default
    draw, like in the reference (hifigan_pretrained.py:186-190), and what the checkpoint provides must be the checkpoint's."""
    import torch
    from iris import hifigan_pretrained as hp
    torch.manual_seed(0)
    donor = hp.HiFiGANModel(upsample_initial_channel=32)
    sd = {k: v.clone() for k, v in donor.state_dict().items()}
    dropped = [k for k in sd if k.startswith("ups.1.") or k == "conv_post.weight_g"]
    for k in dropped:
        del sd[k]
    m = hp.HiFiGANModel(upsample_initial_channel=32, _init_weights=False)
    res = m.load_state_dict(sd, strict=False)
    assert sorted(res.missing_keys) == sorted(dropped)
    m.finish_init(res.missing_keys)
    torch.nn.Module.load_state_dict(m, {k: v for k, v in sd.items() if k.startswith("conv_post.")}, strict=False)
    got = m.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k], v), k                                   # provided by the checkpoint
    for name in ("ups.1", "conv_post"):
        mod = dict(m.named_modules())[name]
        assert mod.initialised and torch.isfinite(mod.weight_v).all() and mod.weight_v.abs().max() > 0
    g, v = got["ups.1.weight_g"], got["ups.1.weight_v"]
    assert torch.allclose(g.flatten(), v.flatten(1).norm(dim=1), rtol=1e-6)   # weight_norm's init: g = ||v||
    assert all(mod.initialised for mod in m.modules() if isinstance(mod, hp._WeightNormedConv))
    # a deferred module nobody finished is completed by the first engine() call instead of running on garbage
    lazy = hp.HiFiGANModel(upsample_initial_channel=32, _init_weights=False)
    assert not lazy.conv_pre.initialised
    try:
        lazy.engine()
    except RuntimeError:
        pass                                                               # (no HIP device here: the engine itself cannot be built)
    assert lazy.conv_pre.initialised and torch.isfinite(lazy.conv_pre.weight_v).all()
