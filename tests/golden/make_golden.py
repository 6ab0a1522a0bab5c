#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (the reference does not exist on the GPU box):

    cd /root/repo && python tests/golden/make_golden.py

It imports the reference's own ``iris.hifigan_pretrained`` from /root/reference/src (the PyTorch
twin: HiFiGANModel, HiFiGANGenerator, infer_hifigan), loads seeded synthetic weights into it with
``load_state_dict`` (no checkpoint ships with the reference) and stores inputs, outputs and a few
intermediate activations as data.  No reference source is copied: fixtures hold numbers only.
The seeded weight generator is loaded BY PATH from iris-tts_amd/iris/_weights.py (numpy only) so the
product and the reference are fed bit-identical state dicts.
"""
import importlib.util
import json
import sys
import tempfile
import warnings
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
REF_SRC = Path("/root/reference/src")
OUT = Path(__file__).resolve().parent


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    warnings.filterwarnings("ignore")
    amd_w = load_by_path("amd_weights", REPO / "iris-tts_amd" / "iris" / "_weights.py")
    sys.path.insert(0, str(REF_SRC))
    import torch
    import iris.hifigan_pretrained as ref  # the reference module

    assert Path(ref.__file__).resolve().is_relative_to(REF_SRC), ref.__file__
    torch.manual_seed(0)
    torch.set_num_threads(8)

    def ref_model(cfg, sd_np):
        m = ref.HiFiGANModel(
            in_channels=cfg.in_channels, upsample_rates=list(cfg.upsample_rates),
            upsample_kernel_sizes=list(cfg.upsample_kernel_sizes),
            upsample_initial_channel=cfg.upsample_initial_channel,
            resblock_kernel_sizes=list(cfg.resblock_kernel_sizes),
            resblock_dilation_sizes=[list(d) for d in cfg.resblock_dilation_sizes])
        res = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        return m.eval()

    def run_with_taps(m, mel):
        taps = {}
        hooks = [m.conv_pre.register_forward_hook(lambda _m, _i, o: taps.__setitem__("conv_pre", o.detach().numpy().copy()))]
        for i, up in enumerate(m.ups):
            hooks.append(up.register_forward_hook(
                lambda _m, _i, o, i=i: taps.__setitem__(f"ups.{i}", o.detach().numpy().copy())))
        # the MRF output of stage i is the input of ups[i+1] before its leaky_relu; take it from the
        # resblock outputs: mrf.i = (rb0 + rb1 + rb2) / 3 in the reference's own order
        rb_out = {}
        for idx, rb in enumerate(m.resblocks):
            hooks.append(rb.register_forward_hook(
                lambda _m, _i, o, idx=idx: rb_out.__setitem__(idx, o.detach().clone())))
        with torch.no_grad():
            y = m(torch.from_numpy(mel))
        for h in hooks:
            h.remove()
        nk = m.num_kernels
        for i in range(m.num_upsamples):
            xs = None
            for j in range(nk):
                # NOTE: the reference accumulates with in-place `+=` into rb0's output tensor, so the
                # hooked tensor of branch 0 already holds the running sum; clone() above was taken at
                # hook time (before the +=), so rebuild the sum in the same left-to-right order.
                xs = rb_out[i * nk + j] if xs is None else xs + rb_out[i * nk + j]
            taps[f"mrf.{i}"] = (xs / nk).numpy().copy()
        return y.numpy().copy(), taps

    V1 = amd_w.GeneratorConfig()
    manifest = {"generator": "tests/golden/make_golden.py", "reference": "ZECTBynmo/iris-tts @ /root/reference",
                "torch": torch.__version__, "cases": {}}

    # ---- case A: V1 config, default-scale weights, T=4, with per-layer taps -------------------
    sd = amd_w.seeded_state_dict(V1, seed=2024)
    m = ref_model(V1, sd)
    mel = amd_w.seeded_mel(1001, 1, 4)
    y, taps = run_with_taps(m, mel)
    np.savez(OUT / "v1_default_T4_taps.npz", mel=mel, wav=y, **{k.replace(".", "_"): v for k, v in taps.items()})
    manifest["cases"]["v1_default_T4_taps"] = {"cfg": "V1", "weights": {"seed": 2024, "gain": 1.0, "post_gain": 1.0},
                                               "mel": {"seed": 1001, "shape": list(mel.shape)},
                                               "max_abs_wav": float(np.abs(y).max())}

    # ---- case B: V1, default-scale weights, batch 2, T=16 (batch + longer receptive field) ----
    mel = amd_w.seeded_mel(1002, 2, 16)
    with torch.no_grad():
        y = m(torch.from_numpy(mel)).numpy()
    np.savez(OUT / "v1_default_B2_T16.npz", mel=mel, wav=y)
    manifest["cases"]["v1_default_B2_T16"] = {"cfg": "V1", "weights": {"seed": 2024, "gain": 1.0, "post_gain": 1.0},
                                              "mel": {"seed": 1002, "shape": list(mel.shape)},
                                              "max_abs_wav": float(np.abs(y).max())}

    # ---- case C: V1, amplified weights (pre-tanh spans +-3), log-mel-like input, T=24 ----------
    gain, post_gain = 1.18, 20.0
    sd_amp = amd_w.seeded_state_dict(V1, seed=2025, gain=gain, post_gain=post_gain)
    m_amp = ref_model(V1, sd_amp)
    mel = amd_w.seeded_mel(1003, 1, 24, log_mel=True)
    pre_tanh = {}
    h = m_amp.conv_post.register_forward_hook(lambda _m, _i, o: pre_tanh.__setitem__("v", o.detach().numpy().copy()))
    with torch.no_grad():
        y = m_amp(torch.from_numpy(mel)).numpy()
    h.remove()
    np.savez(OUT / "v1_amplified_T24.npz", mel=mel, wav=y)
    manifest["cases"]["v1_amplified_T24"] = {"cfg": "V1", "weights": {"seed": 2025, "gain": gain, "post_gain": post_gain},
                                             "mel": {"seed": 1003, "shape": list(mel.shape), "log_mel": True},
                                             "max_abs_wav": float(np.abs(y).max()),
                                             "pre_tanh_abs_max": float(np.abs(pre_tanh["v"]).max()),
                                             "pre_tanh_frac_gt1": float((np.abs(pre_tanh["v"]) > 1).mean())}

    # ---- case D: small config (channels 16,8,4,2; ragged rates/kernels) exercises generic paths ----
    small = amd_w.GeneratorConfig(in_channels=20, upsample_rates=(4, 2, 3), upsample_kernel_sizes=(8, 4, 9),
                                  upsample_initial_channel=48, resblock_kernel_sizes=(3, 5),
                                  resblock_dilation_sizes=((1, 2), (2, 6)))
    sd_s = amd_w.seeded_state_dict(small, seed=7, gain=1.3, post_gain=6.0)
    m_s = ref_model(small, sd_s)
    mel = amd_w.seeded_mel(1004, 3, 19, n_mels=20)
    with torch.no_grad():
        y = m_s(torch.from_numpy(mel)).numpy()
    np.savez(OUT / "small_cfg_B3_T19.npz", mel=mel, wav=y)
    manifest["cases"]["small_cfg_B3_T19"] = {
        "cfg": {"in_channels": 20, "upsample_rates": [4, 2, 3], "upsample_kernel_sizes": [8, 4, 9],
                "upsample_initial_channel": 48, "resblock_kernel_sizes": [3, 5],
                "resblock_dilation_sizes": [[1, 2], [2, 6]]},
        "weights": {"seed": 7, "gain": 1.3, "post_gain": 6.0},
        "mel": {"seed": 1004, "shape": list(mel.shape)}, "max_abs_wav": float(np.abs(y).max())}

    # ---- weight-norm folding as the reference's parametrised modules compute it -----------------
    fold = {}
    eff = {name: getattr(mod, "weight").detach().numpy() for name, mod in m.named_modules() if hasattr(mod, "weight_v")}
    stats = {}
    for name, w in eff.items():
        stats[name] = [float(w.astype(np.float64).sum()), float((w.astype(np.float64) ** 2).sum())]
    for name in ("conv_post", "ups.3", "resblocks.9.convs1.0", "resblocks.11.convs2.2"):
        fold[name.replace(".", "_")] = eff[name]
    np.savez(OUT / "v1_default_folded_weights.npz", **fold)
    manifest["folded_weight_stats"] = stats
    manifest["state_dict_keys"] = sorted(m.state_dict().keys())

    # ---- wrapper semantics: HiFiGANGenerator.__call__ / infer_hifigan shapes and dtypes ----------
    wrap = {}
    with tempfile.TemporaryDirectory() as td:
        ck = Path(td) / "generator.ckpt"
        torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, ck)   # our own file, plain tensors
        gen = ref.HiFiGANGenerator(ck)
        for shape in ((80, 5), (1, 80, 5), (2, 80, 5)):
            x = np.random.default_rng(0).standard_normal(shape)  # float64 on purpose
            a = gen(x)
            b = ref.infer_hifigan(x, checkpoint_path=ck)
            wrap[str(list(shape))] = {"generator_call": {"shape": list(a.shape), "dtype": str(a.dtype)},
                                      "infer_hifigan": {"shape": list(b.shape), "dtype": str(b.dtype)}}
        # container variants accepted by the loader
        for key in ("generator", "model", "state_dict"):
            ck2 = Path(td) / f"nested_{key}.ckpt"
            torch.save({key: {k: torch.from_numpy(v) for k, v in sd.items()}}, ck2)
            g2 = ref.HiFiGANGenerator(ck2)
            wrap[f"nested_{key}_matches_flat"] = bool(np.array_equal(g2(np.zeros((80, 3), np.float32)),
                                                                     gen(np.zeros((80, 3), np.float32))))
        try:
            ref.HiFiGANGenerator(Path(td) / "missing.ckpt")
        except FileNotFoundError as exc:
            wrap["missing_error"] = type(exc).__name__ + ": " + str(exc).replace(td, "<tmp>")
        ck3 = Path(td) / "list.ckpt"
        torch.save([1, 2, 3], ck3)
        try:
            ref.HiFiGANGenerator(ck3)
        except ValueError as exc:
            wrap["bad_format_error"] = type(exc).__name__ + ": " + str(exc)
    manifest["wrapper_semantics"] = wrap

    (OUT / "manifest.json").write_text(json.dumps(manifest, indent=1, sort_keys=True) + "\n")
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)
    print(json.dumps({k: v for k, v in manifest["cases"].items()}, indent=1))


if __name__ == "__main__":
    main()
