"""Whole waveforms against the ORACLE across the launch planner's regime boundaries.

``mrf_plan`` (csrc/mrf_conv_mfma_f32.h) picks, per MRF stage and per (batch, frames), among tile-serial launches
(full / half tile height), fixed per-branch block ranges (all block slots / one block per CU), snake-ordered
(tile, branch) jobs (full / half height), 128-row tiles where they fill the chip, the 16 x 16-job kernel for short inputs, fused conv pairs and the summing
forms.  Every one of them must compute ``HiFiGANModel.forward`` (reference src/iris/hifigan_pretrained.py:123-143): the
shapes below are drawn across the switches (batch 1: 130 ... 999 frames; batches 2, 3, 5 at 200 ... 800 frames), a CPU
test asserts -- from ``iris_hifigan_describe_plan``, which needs no device -- that the sweep really visits every plan
kind, and the GPU tests compare item 0 and the last item of every shape with ``generator_forward_torch`` (<= 1e-4,
north_star's bar) and, for the bf16-storage variant (one launch plan for all shapes), with its restatement.
"""
import collections
import re

import numpy as np
import pytest
import torch

from iris import _native
from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
from oracle import hifigan_oracle as orc

TOL_WAV = 1e-4                                   # north_star: <= 1e-4 max-abs vs the reference fp32 generator
TOL_BF16_MAX, TOL_BF16_MEAN = 6e-2, 5e-3         # tests/test_gpu_bf16.py: unpinned by the reference (it has no bf16 path)

SWEEP_SHAPES = ([(1, t) for t in (130, 150, 190, 200, 240, 282, 350, 420, 480, 501, 560, 600, 650, 700, 850, 950, 999)]
                + [(2, 200), (2, 450), (2, 800), (3, 260), (3, 333), (3, 700), (5, 200), (5, 500), (5, 800)])
BF16_SHAPES = [(1, 130), (1, 282), (1, 501), (1, 850), (2, 450), (3, 333), (5, 200)]

PLAN_KINDS = {"small", "pair", "pair_sum", "serial_tall", "serial_full", "serial_half", "sum_full", "sum_half", "ranges_all", "ranges_percu",
              "snake_full", "snake_half", "dyn_tiles"}

_MRF = re.compile(r"mrf_conv_mfma_f32_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\w+), (\d+)(?:, \w+)*>")


def plan_kinds(cfg, B, T, cu_count=256):
    """Counter of the MRF plan kinds of one fp32 forward, from the host-only launch plan."""
    plan = _native.describe_plan(cfg, B, T, _native.DTYPE_F32, cu_count)
    kinds = collections.Counter()
    stage, L = -1, T
    for rec in plan["launches"]:
        name, grid = rec["kernel"], rec["grid"][0]
        if name.startswith(("conv_mfma_f32_kernel", "convt_mfma_f32_kernel")):      # conv_pre, then one upsampler per stage
            if kinds.get("_pre"):
                stage += 1
                L *= cfg.upsample_rates[stage]
            kinds["_pre"] = 1
        if name.startswith("mrf_small_f32_kernel"):
            kinds["small"] += 1
        elif name.startswith("mrf_pair_f32_pf_kernel"):
            kinds["pair_sum"] += 1
        elif name.startswith("mrf_pair_f32_kernel"):
            kinds["pair"] += 1
        else:
            m = _MRF.match(name)
            if not m:
                continue
            WT, WC, MT = int(m.group(1)), int(m.group(2)), int(m.group(3))
            summing, mode = m.group(9) == "true", int(m.group(10))
            height = {1: "half", 2: "full", 4: "tall"}[MT]       # 32- / 64- / 128-row tiles
            if summing:
                kinds["sum_" + height] += 1
            elif mode == 0:
                kinds["serial_" + height] += 1
            elif mode == 1:
                kinds["ranges_percu" if grid <= cu_count else "ranges_all"] += 1
            else:
                kinds["snake_" + height] += 1
            if mode == 0 and B * T >= 2000:
                # tiles drawn from the per-launch counter: used when a block walks four or more tiles (launch_mrf_conv)
                C = cfg.stage_channels(stage)
                tiles = -(-L // (WT * MT * 32)) * -(-C // (WC * 32)) * B
                if tiles >= 4 * grid:
                    kinds["dyn_tiles"] += 1
    kinds.pop("_pre", None)
    return kinds


def test_sweep_visits_every_plan_kind():
    """Host-only: the shapes of the GPU sweep below exercise every launch-plan kind of the fp32 MRF stages at least once
    (if the planner's model changes, this fails here and the shape list is re-drawn -- not silently on the GPU box)."""
    cfg = GeneratorConfig()
    seen = collections.Counter()
    for B, T in SWEEP_SHAPES:
        kinds = plan_kinds(cfg, B, T)
        assert sum(kinds[k] for k in kinds if k != "dyn_tiles") >= 4 * 3       # every MRF stage launches something
        seen.update(kinds.keys())
    assert set(seen) == PLAN_KINDS, sorted(PLAN_KINDS - set(seen))
    for kind in PLAN_KINDS - {"serial_half", "sum_half", "dyn_tiles"}:
        assert seen[kind] >= 2, (kind, seen[kind])                             # the common ones on more than one shape


@pytest.fixture(scope="module")
def sweep_engine():
    from iris._engine import GeneratorEngine
    dev = torch.device("cuda", 0)
    cfg = GeneratorConfig()
    sd = seeded_state_dict(cfg, seed=2025, gain=1.18, post_gain=20.0)         # the amplified set: tanh reaches +-0.99
    eng = GeneratorEngine(cfg, sd, dev, graph_max_frames=0)                   # eager launches: the planner's own choices
    yield eng, orc.to_torch_folded(sd), dev
    eng.close()


@pytest.mark.gpu
def test_planner_boundary_sweep_matches_oracle(sweep_engine):
    eng, folded, dev = sweep_engine
    worst = 0.0
    for n, (B, T) in enumerate(SWEEP_SHAPES):
        mel = seeded_mel(7000 + n, B, T, log_mel=bool(n & 1))
        got = eng.forward(torch.from_numpy(mel).to(dev), dtype="f32").cpu().numpy()
        idx = sorted({0, B - 1})
        want = orc.generator_forward_torch(folded, mel[idx]).numpy()[:, 0, :]
        assert got.shape == (B, 256 * T) and np.isfinite(got).all()
        err = float(np.abs(got[idx] - want).max())
        assert err <= TOL_WAV, (B, T, err, dict(plan_kinds(eng.cfg, B, T)))
        worst = max(worst, err)
    assert worst > 0.0                                                        # (not an all-zero waveform compared with itself)


@pytest.mark.gpu
def test_planner_boundary_sweep_bf16_matches_restatement(sweep_engine):
    """The bf16-storage variant has ONE launch plan whatever the shape (fused pairs at C <= 128, summing pairs at C <= 64, the
    generic kernel at C = 256); the same shape classes, against its CPU restatement and the fp32 oracle."""
    eng, folded, dev = sweep_engine
    for n, (B, T) in enumerate(BF16_SHAPES):
        mel = seeded_mel(7100 + n, B, T, log_mel=bool(n & 1))
        got = eng.forward(torch.from_numpy(mel).to(dev), dtype="bf16").cpu().numpy()
        idx = sorted({0, B - 1})
        ref32 = orc.generator_forward_torch(folded, mel[idx]).numpy()[:, 0, :]
        emu16 = orc.generator_forward_bf16(folded, mel[idx]).numpy()[:, 0, :]
        assert got.shape == (B, 256 * T) and np.isfinite(got).all() and np.abs(got).max() <= 1.0
        for want in (ref32, emu16):
            d = np.abs(got[idx] - want)
            assert d.max() <= TOL_BF16_MAX and d.mean() <= TOL_BF16_MEAN, (B, T, float(d.max()), float(d.mean()))
