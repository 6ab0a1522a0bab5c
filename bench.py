#!/usr/bin/env python3
"""Headline benchmark of the HiFiGAN vocoder path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path -- mel [B, 80, T] -> waveform [B, 256*T] through
``iris_hifigan_forward`` -- over one batch of synthetic mels that is already resident in HBM when
the timed region starts.

N = 1   BASELINE.json configs[1]: batch 1, 80 x 1000 frames, fp32 (the headline `value`).  Outside the
        headline's timed region the same run also measures, as sub-records of the ONE JSON line:
          grid            B = 1 x {100, 500, 1000} frames fp32 (north_star's reporting grid): ms, samples/s, RTF,
                          MRF-kernel roofline fraction
          host_inclusive  numpy in -> H2D -> forward -> D2H -> numpy out, what the reference's API does
                          (src/iris/hifigan_pretrained.py:228,235); never `value`
          configs2        BASELINE.json configs[2]: batch 32 x 500 frames, bf16 storage, HBM roofline of the MRF kernels
          configs4        BASELINE.json configs[4]: PostNet -> vocoder on device, 1024 frames streamed in 256-frame chunks
                          (first_chunk_ms, total_ms, one_shot_ms, halo_overhead_frac), fp32 and bf16
          configs3_n1     the one-GPU leg of configs[3]: batch 256 x 1000 frames fp32 (so that 8-vs-1 is computable)
          cpu_baseline    the oracle timed on the host cores; other_modes: the other arithmetic modes
N > 1   BASELINE.json configs[3]: a GLOBAL batch of 256 mels x 1000 frames sharded over the N ranks (strong
        scaling: total work fixed), one process per GPU, no data-path collective but the final RCCL all-gather
        of the waveform shards, which IS inside the step; `grid` adds the same global batch at 100 and 500 frames.
        When WORLD_SIZE is not set, ``python bench.py --gpus N`` starts the N ranks itself (torch.distributed.run
        as a child process, before this process touches the GPU) and passes rank 0's line through.

The JSON line also carries
  roofline      the dominant kernel (the MFMA Conv1d kernel behind the MRF ResBlocks): algorithmic
                FLOP of its launches / their HIP-event durations, measured live in the timed region
                on the stream the kernels run on; peak = 157.3 TFLOP/s (fp32 MFMA, MI355X_MICROARCH.md).
                The path is a dense contraction at 118 FLOP/B: fp32 is MFMA-bound, not HBM-bound
                (SURVEY.md 8d), so ``bound`` is "mfma"; the HBM fraction of the same launches is
                reported beside it as ``hbm_frac``.  ``traffic`` is NOT live: it is the committed PMC
                measurement of the same workload (``traffic_source`` names the file).
  cpu_baseline  the oracle (torch fp32 restatement of the reference, oracle/hifigan_oracle.py) timed on
                this box's host cores on a bounded sample of the same workload; rank 0, N = 1 only.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
for p in (str(REPO / "iris-tts_amd"), str(REPO)):
    if p not in sys.path:
        sys.path.insert(0, p)

SAMPLE_RATE = 22050
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0 # dense, same table
PEAK_HBM_GBS = 8000.0          # spec; 6.3 TB/s achievable
GLOBAL_BATCH = 256             # BASELINE.json configs[3]
MODE_TEXT = {"f32": "fp32", "bf16": "bf16 storage / fp32 accumulate",
             "f32s": "fp32 storage, split-bf16 products in the ResBlock convs"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="N = 1: mels per step (default 1: configs[1])")
    ap.add_argument("--global-batch", type=int, default=GLOBAL_BATCH,
                    help="N > 1: mels per step over ALL ranks (default 256: configs[3]); rank r takes shard_bounds()[r]")
    ap.add_argument("--frames", type=int, default=1000, help="mel frames per item")
    ap.add_argument("--dtype", choices=["f32", "bf16", "f32s"], default="f32",
                    help="f32 = the parity path and the headline (default); bf16 = bf16 storage + bf16 MFMA "
                         "(BASELINE.json configs[2] with --batch 32 --frames 500)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline measurement (no grid / host_inclusive / "
                    "configs2 / configs3_n1 / other_modes sub-records)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket launches with HIP events")
    ap.add_argument("--per-launch-events", action="store_true",
                    help="one HIP-event pair per launch (31 events per forward, ~0.1 ms of stream time) instead of one per "
                         "layer group (the MRF launches of a stage share a pair: 11 events)")
    ap.add_argument("--graph", action="store_true", help="replay the forward as a hipGraph (no per-launch records: "
                    "roofline is then null; the default eager mode is the measured configuration)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and run the collective "
                    "path (barrier, all-gather, max-reduce) even with one rank: exercises the RCCL code path of N > 1 "
                    "on a one-GPU box")
    ap.add_argument("--include-h2d", action="store_true", help="upload this rank's mel shard (pinned host memory -> HBM) inside "
                    "every timed step: SURVEY.md 8e names the 10.2 MB-per-rank upload as the scaling risk; the line is "
                    "then marked h2d_in_step and is not the headline (inputs resident in HBM)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; "
                    "gloo only for rehearsing the control flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--cold-start-child", metavar="CKPT", default=None,
                    help="(internal) fresh-process leg of the `load` record: load CKPT through iris.hifigan_pretrained, vocode one "
                         "100-frame mel, print the timings as one JSON line")
    ap.add_argument("--stub-engine", action="store_true", help="CONTROL-FLOW REHEARSAL ONLY (CPU tests of the launcher, the "
                    "sharding and the gather): replaces the HIP engine by a stand-in that does not compute the vocoder; "
                    "the line is marked \"stub\" and is not a measurement")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------
# parent of an N > 1 run that was started as plain `python bench.py --gpus N`
# ------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    """Starts one rank per GPU with torch.distributed.run and waits.  This process has made no HIP call (torch is
    not even imported here): the ranks are children, nothing that touched the GPU is replaced by another program."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


# ------------------------------------------------------------------------------------------------------
# measurement helpers (rank processes)
# ------------------------------------------------------------------------------------------------------
def csrc_digest():
    """sha256 (16 hex digits) over the kernel sources the library is built from (iris-tts_amd/csrc/*.h, *.hip, sorted by name).
    tools/hbm_traffic.py stamps it into every committed PMC file; `committed_traffic` refuses a file whose stamp differs
    from the tree's -- a kernel change without a fresh PMC pass then reports `traffic: null`, not a stale figure."""
    import hashlib
    h = hashlib.sha256()
    csrc = REPO / "iris-tts_amd" / "csrc"
    for f in sorted(list(csrc.glob("*.h")) + list(csrc.glob("*.hip"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def committed_traffic(batch, frames, dtype="f32", live_launches=None):
    """(HBM bytes per launch of the dominant kernel, source, reason) from the PMC passes committed under profiles/
    (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, gfx950 correction applied:
    profiles/r*_hbm_traffic.json, produced by tools/hbm_traffic.sh).  PMC counters cannot be collected from
    inside this process, so the figure is the committed measurement of the same workload -- and only when that pass
    (a) counted the same number of MRF launches per forward as this run issues and (b) was taken on the kernel sources of
    this tree (`csrc_sha16`); otherwise (None, None, why)."""
    if (dtype, batch, frames) == ("f32", 1, 1000):
        files = [f for f in sorted((REPO / "profiles").glob("r*_hbm_traffic.json")) if "bf16" not in f.name]
    elif (dtype, batch, frames) == ("bf16", 32, 500):
        files = sorted((REPO / "profiles").glob("r*_bf16_c3_hbm_traffic.json"))
    else:
        return None, None, "no PMC pass is committed for this workload"
    if not files:
        return None, None, "no PMC pass is committed for this workload"
    f = files[-1]
    try:
        rec = json.loads(f.read_text())
        traffic = rec["mrf_traffic_bytes_per_launch"]
    except (KeyError, ValueError):
        return None, None, f"profiles/{f.name} is not readable"
    if live_launches is not None and rec.get("mrf_launches") != live_launches:
        return None, None, (f"profiles/{f.name} counted {rec.get('mrf_launches')} MRF launches per forward, this run issues "
                            f"{live_launches}: stale PMC pass (re-run tools/hbm_traffic.sh)")
    if rec.get("csrc_sha16") != csrc_digest():
        return None, None, (f"profiles/{f.name} was taken on kernel sources {rec.get('csrc_sha16')}, this tree is "
                            f"{csrc_digest()}: stale PMC pass (re-run tools/hbm_traffic.sh)")
    return traffic, f"profiles/{f.name} (committed PMC pass of these kernel sources, not live)", None


def cpu_baseline(cfg, sd, mel, budget_s=24.0):
    """Times the oracle's torch-fp32 forward (the arithmetic the reference's PyTorch twin runs,
    src/iris/hifigan_pretrained.py:123-143) on the host cores, at the thread count that is fastest ON THE TIMED WORKLOAD:
    ATen's convolutions at batch 1 get slower with too many threads, and the best count differs between a 100-frame clip
    and the 1000-frame utterance (round 3 chose on the clip and reported 1.8x less than the box could do).
    Step 1 ranks {8, 16, 32, 64, all} threads on a 100-frame clip (cheap: also the `c1` record, BASELINE.json configs[0] shape);
    step 2 times the HEADLINE mel itself (after one untimed warm-up run) at each of the two best counts; step 3 repeats the
    better one (median of all its timed runs).  Every timing is in `threads_tried`."""
    import torch
    from oracle import hifigan_oracle as orc

    folded = orc.to_torch_folded(sd)
    all_cores = torch.get_num_threads()
    x = torch.from_numpy(mel)
    clip = x[:1, :, :100] if x.shape[2] >= 100 else x[:1]
    t_begin = time.perf_counter()
    tried = {}
    for n in sorted({c for c in (8, 16, 32, 64, all_cores) if 1 <= c <= all_cores} | {all_cores}):
        torch.set_num_threads(n)
        orc.generator_forward_torch(folded, clip)                 # warm-up (allocator, thread pool)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            orc.generator_forward_torch(folded, clip)
            ts.append(time.perf_counter() - t0)
        tried[n] = min(ts)
        if time.perf_counter() - t_begin > 0.25 * budget_s:       # (a slow box: keep the rest of the budget for the utterance)
            break
    clip_best = min(tried, key=tried.get)
    candidates = sorted(tried, key=tried.get)[:2]
    full, out_ref = {}, None
    for n in candidates:                                           # the timed workload itself: one warm-up, one timed run per candidate
        torch.set_num_threads(n)
        orc.generator_forward_torch(folded, x)                     # (the first run at a new thread count pays the pool's start-up)
        t0 = time.perf_counter()
        out_ref = orc.generator_forward_torch(folded, x)
        full[n] = [time.perf_counter() - t0]
        if time.perf_counter() - t_begin > 0.6 * budget_s:
            break
    best_n = min(full, key=lambda n: full[n][0])
    torch.set_num_threads(best_n)
    while len(full[best_n]) < 3 and time.perf_counter() - t_begin + full[best_n][-1] < budget_s:
        t0 = time.perf_counter()
        out_ref = orc.generator_forward_torch(folded, x)
        full[best_n].append(time.perf_counter() - t0)
    torch.set_num_threads(all_cores)
    best = statistics.median(full[best_n])
    cpu_baseline.last_output = out_ref.numpy()[:, 0, :]       # the checker's waveform of this mel (parity of the GPU modes)
    clip_samples = clip.shape[2] * 256
    samples = mel.shape[0] * mel.shape[2] * 256
    return {"value": samples / best, "unit": "samples/s", "cores": best_n, "kind": "port",
            "host_cores": all_cores, "threads_used": best_n,
            "threads_tried": {str(n): {"ms_100_frames": 1e3 * t, "samples_per_s_100_frames": clip_samples / t,
                                       **({"s_timed_workload": full[n], "samples_per_s_timed_workload": samples / statistics.median(full[n])}
                                          if n in full else {})} for n, t in tried.items()},
            "c1": {"workload": "batch 1 x 80-mel x 100 frames (BASELINE.json configs[0] shape)", "threads": clip_best,
                   "ms": 1e3 * tried[clip_best], "samples_per_s": clip_samples / tried[clip_best],
                   "rtf": tried[clip_best] / (clip_samples / SAMPLE_RATE)},
            "sample": f"{len(full[best_n])} full forward(s) of the same B={mel.shape[0]} x 80 x {mel.shape[2]} mel at {best_n} threads; median "
                      f"{best:.3f} s; torch {torch.__version__} CPU fp32; thread count chosen on this workload between {sorted(full)} "
                      f"(the two fastest of {sorted(tried)} on a 100-frame clip; the box has {all_cores}); "
                      f"oracle/hifigan_oracle.py:generator_forward_torch",
            "rtf": best / (mel.shape[2] * 256 / SAMPLE_RATE) if mel.shape[0] == 1 else None}


def summarize_records(recs, steps, cfg):
    """Per-kind and per-MRF-stage sums of the live HIP-event records of `steps` forwards."""
    by_kind, detail = {}, {}
    for r in recs:
        k = by_kind.setdefault(r["kind"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0})
        k["ms"] += r["ms"]; k["flops"] += r["flops"]; k["bytes"] += r["bytes"]; k["n"] += r.get("launches", 1)
    for kind, k in by_kind.items():
        detail[kind] = {"launches_per_step": k["n"] // steps, "ms_per_step": k["ms"] / steps,
                        "tflops": k["flops"] / (k["ms"] * 1e-3) / 1e12, "gbs": k["bytes"] / (k["ms"] * 1e-3) / 1e9}
    for stage in range(cfg.num_upsamples):
        rs = [r for r in recs if r["kind"] == "mrf_resblock_conv" and r["stage"] == stage]
        ms = sum(r["ms"] for r in rs)
        if ms > 0:
            detail[f"mrf_stage{stage}_C{cfg.stage_channels(stage)}"] = {
                "ms_per_step": ms / steps, "tflops": sum(r["flops"] for r in rs) / (ms * 1e-3) / 1e12,
                "gbs": sum(r["bytes"] for r in rs) / (ms * 1e-3) / 1e9}
    return by_kind, detail


def roofline_of(by_kind, dtype, steps, ms_per_step, batch, frames):
    """The `roofline` object of the dominant kernel (the MRF ResBlock conv launches) from live records."""
    dom = by_kind.get("mrf_resblock_conv")
    if not dom or dom["ms"] <= 0:
        return None
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
    traffic, source, why = (committed_traffic(batch, frames, "bf16" if dtype == "bf16" else "f32", dom["n"] // steps)
                            if dtype != "f32s" else (None, None, "no PMC pass is committed for this mode"))
    common = {"traffic": traffic, "traffic_source": source, "traffic_null_reason": why, "launches_per_step": dom["n"] // steps,
              "avg_launch_ms": dom["ms"] / dom["n"], "flop_per_launch": dom["flops"] / dom["n"],
              "bytes_per_launch": dom["bytes"] / dom["n"], "share_of_step": dom["ms"] / steps / ms_per_step}
    if dtype in ("f32", "f32s"):
        # f32s: three bf16 MFMAs per fp32 product -> the matrix roof for fp32-equivalent FLOP is a third of the bf16 peak
        peak = PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_BF16_MFMA_TFLOPS / 3.0
        return {"kernel": ("fp32 MRF ResBlock kernels: mrf_conv_mfma_f32_kernel, mrf_small_f32_kernel (short inputs), mrf_pair_f32_kernel "
                           "and mrf_pair_f32_pf_kernel<sum> (fused conv pairs, C <= 64)" if dtype == "f32"
                           else "conv_mfma_f32s_kernel (split-bf16 products)") + " (MRF ResBlock Conv1d steps)",
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "hbm_achieved_gbs": gbs, "hbm_peak_gbs": PEAK_HBM_GBS, "hbm_frac": gbs / PEAK_HBM_GBS, **common}
    # bf16.  Two byte counts, kept apart:
    #   accounting L (`hbm_accountingL_*`): every conv reads its input and writes its output once -- five
    #       tensor passes per ResBlock conv pair -- whether or not a fused launch moves them.  north_star's ">= 50 % of the
    #       HBM roofline" is quoted in this accounting, so it stays reported, under its own name;
    #   traffic (`frac` when available): the bytes the launches really move -- the committed PMC pass of the same workload
    #       (2 * FETCH_SIZE + WRITE_SIZE per launch) -- over the live launch time.  A fused pair moves ~2 passes, not 5.
    # The bound named is the roof the launches are closer to: real HBM bytes / 8 TB/s against FLOP / 2.5 PFLOP/s.
    hbm_frac_L = gbs / PEAK_HBM_GBS
    mfma_frac = achieved / PEAK_BF16_MFMA_TFLOPS
    real_gbs = traffic / (dom["ms"] / dom["n"] * 1e-3) / 1e9 if traffic else None
    hbm_frac_real = real_gbs / PEAK_HBM_GBS if real_gbs else None
    base = {"kernel": "bf16 MRF ResBlock kernels (mrf_pair_bf16_kernel / mrf_pair_bf16_sum_kernel: fused conv pairs at C <= 128; "
                      "conv_mfma_bf16_kernel: the C = 256 steps)",
            "hbm_accountingL_gbs": gbs, "hbm_accountingL_frac": hbm_frac_L, "hbm_peak_gbs": PEAK_HBM_GBS,
            "hbm_traffic_gbs": real_gbs, "hbm_traffic_frac": hbm_frac_real,
            "mfma_achieved_tflops": achieved, "mfma_peak_tflops": PEAK_BF16_MFMA_TFLOPS, "mfma_frac": mfma_frac, **common}
    if hbm_frac_real is not None and hbm_frac_real >= mfma_frac:
        return {"bound": "hbm", "achieved": real_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_frac_real, **base}
    return {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": mfma_frac, **base}


def cold_start_child(ckpt):
    """Runs in a FRESH process (bench.py --cold-start-child): what the reference's only real caller pays before its first
    audio -- load the checkpoint and vocode one utterance (scripts/synthesize.py:197-198 -> hifigan_pretrained.py:250-283,
    208-242) -- through the drop-in module, numpy in / numpy out.  Everything is wall-clock in this process; the first
    forward includes the HIP runtime start, the load of the library's code objects and both PCIe copies."""
    t_proc = time.perf_counter()
    import numpy as np
    import torch
    t_torch = time.perf_counter()
    from iris import hifigan_pretrained as hp
    from iris._engine import LAST_LOAD_TIMINGS
    from iris._weights import seeded_mel
    torch.cuda.init()
    torch.zeros(1, device="cuda").item()                       # HIP runtime + torch's own kernels: not the library's cost
    t_hip = time.perf_counter()
    mel = seeded_mel(1001, 1, 100)
    gen = hp.get_pretrained_hifigan(ckpt)
    t_loaded = time.perf_counter()
    wav = gen(mel)                                             # first call: fold + create + first forward
    t_first = time.perf_counter()
    timings = dict(LAST_LOAD_TIMINGS)
    wav2 = gen(mel)
    t_second = time.perf_counter()
    eng = gen.model.engine()
    t0 = time.perf_counter()
    eng.prepare("bf16")
    torch.cuda.synchronize()
    prepare_bf16 = time.perf_counter() - t0
    first_call = 1e3 * (t_first - t_loaded)
    out = {"import_torch_ms": 1e3 * (t_torch - t_proc), "hip_runtime_init_ms": 1e3 * (t_hip - t_torch),
           "torch_load_ms": timings.get("torch_load_ms"), "model_construct_ms": timings.get("model_construct_ms"),
           "load_state_dict_ms": timings.get("load_state_dict_ms"),
           "get_pretrained_hifigan_ms": 1e3 * (t_loaded - t_hip),
           "fold_ms": timings.get("fold_ms"), "create_ms": timings.get("create_ms"),
           "first_forward_ms": first_call - timings.get("fold_ms", 0.0) - timings.get("create_ms", 0.0),
           "first_call_ms": first_call, "second_call_ms": 1e3 * (t_second - t_first),
           "prepare_bf16_ms": 1e3 * prepare_bf16,
           "create_plus_first_forward_ms": first_call - timings.get("fold_ms", 0.0),
           "load_to_first_audio_ms": 1e3 * (t_first - t_hip),
           "frames": 100, "same_waveform_twice": bool(np.array_equal(wav, wav2)),
           "note": "fresh process; V1 checkpoint (weight-normed state dict under 'generator') from a temp file through "
                   "iris.hifigan_pretrained.get_pretrained_hifigan(); first_forward_ms = first __call__ minus fold and create: "
                   "H2D + code-object load + 24 launches + D2H"}
    print(json.dumps(out), flush=True)


def cold_start_record(sd):
    """The `load` sub-record: writes the V1 checkpoint to a temp file and runs `cold_start_child` in a child process."""
    import tempfile
    import torch
    with tempfile.TemporaryDirectory() as tmp:
        ckpt = os.path.join(tmp, "generator.ckpt")
        torch.save({"generator": {k: torch.from_numpy(v) for k, v in sd.items()}}, ckpt)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        try:
            proc = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--cold-start-child", ckpt],
                                  env=env, capture_output=True, text=True, timeout=600)
            if proc.returncode != 0:
                return {"error": proc.stderr[-500:]}
            return json.loads(proc.stdout.strip().splitlines()[-1])
        except Exception as exc:
            return {"error": str(exc)}


class StubEngine:
    """--stub-engine: stands in for GeneratorEngine so that the launcher, the sharding, the gather and the JSON
    contract can be rehearsed on CPU (gloo).  It does NOT compute the vocoder (every output sample is the mean of
    its frame's mel bins); nothing measured with it is a result."""
    hop_length = 256

    def forward(self, mel, out=None, dtype=None):
        wav = mel.mean(dim=1).repeat_interleave(self.hop_length, dim=1)
        if out is not None:
            out.copy_(wav)
            return out
        return wav

    def set_profiling(self, enabled):
        pass

    def read_profile(self):
        return []


def timed_forward(eng, mel, dtype, steps, warmup, dev, profile, cfg):
    """One single-GPU measurement outside the headline: ms per step (+ live per-launch records when profiling)."""
    import torch
    wav = torch.empty((mel.shape[0], mel.shape[2] * eng.hop_length), dtype=torch.float32, device=dev)
    for _ in range(warmup):
        eng.forward(mel, out=wav, dtype=dtype)
    torch.cuda.synchronize(dev)
    eng.set_profiling(profile)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward(mel, out=wav, dtype=dtype)
    torch.cuda.synchronize(dev)
    ms = 1e3 * (time.perf_counter() - t0) / steps
    recs = eng.read_profile() if profile else []
    eng.set_profiling(False)
    return ms, recs, wav


def rank_main(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
    from iris.distributed import gather_waveforms, shard_range

    # ONE JSON line on stdout, nothing else: libraries that chat on fd 1 (RCCL prints a five-line version banner at init, gloo its
    # connection lines) are sent to stderr for the life of the rank; the line itself is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(json_fd, (line + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    stub = args.stub_engine
    if stub:
        dev = torch.device("cpu")
        if args.backend == "nccl" and (world > 1 or args.force_dist):
            raise SystemExit("--stub-engine rehearses on CPU: use --backend gloo")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (the vocoder path has no CPU fallback)")
        n_dev = torch.cuda.device_count()
        if local_rank >= n_dev and args.backend == "nccl":
            raise SystemExit(f"LOCAL_RANK {local_rank} but only {n_dev} HIP device(s): RCCL needs one GPU per rank")
        dev = torch.device("cuda", local_rank % n_dev)     # (ranks share a GPU only in a gloo rehearsal)
        torch.cuda.set_device(dev)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    cfg = GeneratorConfig()
    sd = seeded_state_dict(cfg, seed=2024)                       # random-init weights of the V1 architecture
    T = args.frames
    strong = world > 1
    if strong:
        G = args.global_batch                                    # configs[3]: the global batch is fixed, ranks split it
        lo, hi = shard_range(G, rank, world)
        B = hi - lo
        seed = 1004 + rank                                       # SURVEY.md 8d seeds
    else:
        B = args.batch if args.batch is not None else 1
        G = B
        seed = 1002
    if stub:
        eng = StubEngine()
    else:
        from iris._engine import GeneratorEngine, algorithmic_work
        eng = GeneratorEngine(cfg, sd, dev)
    hop = eng.hop_length

    def sync():
        if not stub:
            torch.cuda.synchronize(dev)

    def fence():
        sync()
        if use_dist:
            dist.barrier()
        sync()

    def run_sharded(frames, steps, warmup, profile):
        """`steps` timed steps of this rank's shard at `frames` frames (+ the gather); max over ranks."""
        mel_np = seeded_mel(seed, max(B, 1), frames)[:B]
        mel = torch.from_numpy(mel_np).to(dev)
        mel_host = torch.from_numpy(mel_np).pin_memory() if (args.include_h2d and not stub) else None
        wav = torch.empty((B, frames * hop), dtype=torch.float32, device=dev)
        gathered = torch.empty((G, frames * hop), dtype=torch.float32, device=dev) if use_dist else None

        def step():
            if mel_host is not None:
                mel.copy_(mel_host, non_blocking=True)           # same stream as the forward: ordered in front of it
            if args.graph and not stub:
                out = eng.forward_graph(mel, dtype=args.dtype)
            else:
                out = eng.forward(mel, out=wav, dtype=args.dtype)
            if use_dist:
                gather_waveforms(out, G, out=gathered)

        for _ in range(warmup):
            step()
        fence()
        eng.set_profiling(profile)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        recs = eng.read_profile() if profile else []
        eng.set_profiling(False)
        per_rank = [elapsed]
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            every = torch.empty(world, dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(every, t)                # each rank's own clock, for the line's per_rank_ms
            per_rank = [float(v) for v in every.cpu()]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        run_sharded.last = {"per_rank_s": per_rank, "wav": wav, "gathered": gathered}
        return elapsed, recs, mel_np, mel

    # ---- the headline: W untimed warm-up steps, exactly K timed steps between barrier + synchronize -------------
    profile = 0 if (args.no_profile or args.graph or stub) else (1 if args.per_launch_events else 2)
    elapsed, recs, mel_np, mel = run_sharded(T, args.steps, args.warmup, profile)
    samples_per_step = G * T * hop
    value = samples_per_step * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    by_kind, detail = summarize_records(recs, args.steps, cfg) if recs else ({}, {})
    roofline = roofline_of(by_kind, args.dtype, args.steps, ms_per_step, B, T) if recs else None

    per_rank_ms = [1e3 * v / args.steps for v in run_sharded.last["per_rank_s"]]
    collective = None
    if use_dist:
        # The one collective of the path on its own: 10 timed all-gathers of the real waveform shard (outside the
        # headline's timed region), and what the process group itself says about backend and world size -- so that
        # "RCCL saw N ranks" can be read off the line.
        wav_l, gath = run_sharded.last["wav"], run_sharded.last["gathered"]
        for _ in range(2):
            gather_waveforms(wav_l, G, out=gath)
        fence()
        t0 = time.perf_counter()
        for _ in range(10):
            gather_waveforms(wav_l, G, out=gath)
        fence()
        tg = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        collective = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "op": "all_gather of the waveform shards",
                      "gather_only_ms": 1e3 * float(tg.item()) / 10, "bytes_per_rank": int(wav_l.numel() * 4),
                      "bytes_gathered_per_rank": int(gath.numel() * 4), "timed_gathers": 10,
                      "share_of_step": (1e3 * float(tg.item()) / 10) / (1e3 * elapsed / args.steps)}

    # ---- N > 1: the rest of the reporting grid (same global batch, 100 and 500 frames); every rank takes part --------
    grid = []
    if strong and not args.no_extras:
        for frames in (100, 500, 1000):
            if frames == T:
                grid.append({"frames": T, "global_batch": G, "ms_per_step": ms_per_step, "samples_per_s": value})
                continue
            e, _, _, _ = run_sharded(frames, max(3, args.steps // 2), 1, False)
            n = max(3, args.steps // 2)
            grid.append({"frames": frames, "global_batch": G, "ms_per_step": 1e3 * e / n, "samples_per_s": G * frames * hop * n / e})

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    flop_per_frame = 614_105_088 if stub else algorithmic_work(cfg)["flop_per_frame"]
    if strong:
        workload = (f"HiFiGAN-V1 generator, global batch {G} x 80-mel x {T} frames sharded over {world} GPUs "
                    f"({B} items on rank 0) -> {T * hop} samples each, {MODE_TEXT[args.dtype]}"
                    + (" (BASELINE.json configs[3])" if (G, args.dtype) == (GLOBAL_BATCH, "f32") else ""))
    else:
        workload = (f"HiFiGAN-V1 generator, batch {B} x 80-mel x {T} frames -> {T * hop} samples each, {MODE_TEXT[args.dtype]}"
                    + (" (BASELINE.json configs[1])" if (B, T, args.dtype) == (1, 1000, "f32") else "")
                    + (" (BASELINE.json configs[2])" if (B, T, args.dtype) == (32, 500, "bf16") else ""))
    out = {
        "metric": "audio samples/sec (22.05 kHz) on 80-mel x %d-frame batch" % T,
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": args.dtype,
        "data": "stub engine: control-flow rehearsal, NOT a measurement" if stub else "synthetic",
        "config": {"workload": workload, "batch_per_gpu": B, "global_batch": G, "frames": T, "hop_length": hop,
                   "weights": "random-init (seeded) V1 architecture, 13,926,017 values",
                   "sharding": (f"contiguous balanced shards of the global batch, one process per GPU, all-gather of the "
                                f"waveform shards over {args.backend}"
                                + (" (RCCL)" if args.backend == "nccl" else " (control-flow rehearsal, not a measurement)"))
                               if world > 1 else "single GPU",
                   "launch_events_in_timed_region": {0: False, 1: "per launch", 2: "per layer group"}[profile], "hipgraph_replay": bool(args.graph)},
        "rtf": (ms_per_step * 1e-3) / (T * hop / SAMPLE_RATE) if G == 1 else None,
        "flop_per_step": flop_per_frame * G * T,
        "tflops_whole_path": flop_per_frame * G * T / (ms_per_step * 1e-3) / 1e12,
        "roofline": roofline, "kernels": detail,
    }
    if use_dist:
        out["collective"] = collective
        out["per_rank_ms"] = per_rank_ms
    if args.include_h2d:
        out["h2d_in_step"] = {"bytes_per_rank": int(B * 80 * T * 4), "note": "pinned host memory -> HBM inside every timed step; not the headline"}
    out["scaling_note"] = ("strong scaling on the fixed global batch of BASELINE.json configs[3]" if strong else
                           "N = 1 runs BASELINE.json configs[1] (batch 1); the one-GPU leg of the strong-scaling series is configs3_n1")
    if stub:
        out["stub"] = True
    if grid:
        out["grid"] = grid

    extras = world == 1 and not stub and not args.no_extras and not args.force_dist
    if extras:
        # ---- north_star's reporting grid at batch 1 (outside the headline's timed region) ------------------------------
        g = []
        for frames in (100, 500, 1000):
            m = torch.from_numpy(seeded_mel(1001 if frames == 100 else 1002, 1, frames)).to(dev)
            limit, eng.graph_max_frames = eng.graph_max_frames, 0              # eager launches: the configuration with live events
            ms, r, _ = timed_forward(eng, m, "f32", 20, 5, dev, 2, cfg)
            ms_plain, _, _ = timed_forward(eng, m, "f32", 20, 5, dev, False, cfg)
            eng.graph_max_frames = limit
            for _ in range(3):
                eng.forward_graph(m, dtype="f32")
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(20):
                eng.forward_graph(m, dtype="f32")
            torch.cuda.synchronize(dev)
            ms_graph = 1e3 * (time.perf_counter() - t0) / 20
            ms_default, _, _ = timed_forward(eng, m, "f32", 20, 5, dev, False, cfg)   # what GeneratorEngine.forward does by itself
            bk, _ = summarize_records(r, 20, cfg)
            rf = roofline_of(bk, "f32", 20, ms, 1, frames)
            g.append({"frames": frames, "batch": 1, "dtype": "f32", "ms": ms, "samples_per_s": frames * hop / (ms * 1e-3),
                      "rtf": (ms * 1e-3) / (frames * hop / SAMPLE_RATE),
                      "roofline_frac": rf["frac"] if rf else None, "mrf_tflops": rf["achieved"] if rf else None,
                      "ms_eager_no_events": ms_plain, "ms_hipgraph_replay": ms_graph, "ms_engine_default": ms_default,
                      "engine_default_is_graph": bool(frames <= eng.graph_max_frames),
                      "note": "ms / roofline_frac: eager launches with the live per-layer-group events; ms_hipgraph_replay: the same "
                              "launches replayed as one hipGraph (static buffers, no events); ms_engine_default: "
                              "GeneratorEngine.forward as the drop-in wrappers call it (eager launches, no events; graph replay + "
                              f"copy-out only up to graph_max_frames = {eng.graph_max_frames} frames)"})
        out["grid"] = g
        # ---- host-inclusive: numpy in -> numpy out, PCIe both ways, synchronous (never `value`) --------------------------
        def host_call(x):
            t = torch.from_numpy(x).float().to(dev)                      # hifigan_pretrained.py:228
            return eng.forward(t, dtype="f32").cpu().numpy()            # :231-235
        for _ in range(3):
            host_call(mel_np)
        t0 = time.perf_counter()
        for _ in range(10):
            host_call(mel_np)
        hi_ms = 1e3 * (time.perf_counter() - t0) / 10
        out["host_inclusive"] = {"ms": hi_ms, "samples_per_s": B * T * hop / (hi_ms * 1e-3), "batch": B, "frames": T,
                                 "note": "numpy mel -> H2D -> forward -> D2H -> numpy waveform per call (what the reference's "
                                         "API does, hifigan_pretrained.py:228,235); never `value`"}
        # ---- BASELINE.json configs[2]: batch 32 x 500 frames, bf16 storage ------------------------------------------------
        try:
            m2 = torch.from_numpy(seeded_mel(1003, 32, 500)).to(dev)
            ms2, r2, _ = timed_forward(eng, m2, "bf16", 10, 3, dev, 2, cfg)
            bk2, det2 = summarize_records(r2, 10, cfg)
            out["configs2"] = {"workload": "batch 32 x 80-mel x 500 frames, bf16 storage / fp32 accumulate (BASELINE.json configs[2])",
                               "dtype": "bf16", "ms_per_step": ms2, "samples_per_s": 32 * 500 * hop / (ms2 * 1e-3), "steps": 10,
                               "roofline": roofline_of(bk2, "bf16", 10, ms2, 32, 500), "kernels": det2}
            del m2
        except Exception as exc:
            out["configs2"] = {"error": str(exc)}
        # ---- BASELINE.json configs[4]: PostNet -> vocoder chained on the device, one utterance of 1024 frames streamed in
        # 256-frame chunks (+ the 13-frame halo).  The reference has no streaming (src/iris/model.py:17-27 is a stub), so the
        # contract is BASELINE.json; PostNet as scripts/synthesize.py:152-158 instantiates it.  Mel resident in HBM.
        try:
            from iris.pipeline import MelToWavePipeline
            from iris.postnet import PostNet
            from iris.streaming import plan_chunks, receptive_field_frames
            T4, chunk = 1024, 256
            post = PostNet(n_mels=80, num_layers=3, channels=256, kernel_size=5, dropout=0.3, seed=5)
            mel4 = torch.from_numpy(seeded_mel(1005, 1, T4, log_mel=True)).to(dev)
            halo = receptive_field_frames(cfg)
            windows = plan_chunks(T4, chunk, halo)
            c4 = {"workload": f"PostNet (3 x 256 ch, k=5) -> HiFiGAN-V1 on device, 1 x 80-mel x {T4} frames streamed in {chunk}-frame "
                              f"chunks + {halo}-frame halo (BASELINE.json configs[4])",
                  "chunks": len(windows), "halo_frames": halo,
                  "halo_overhead_frac": sum(c.win_stop - c.win_start for c in windows) / T4 - 1.0, "reps": 5, "modes": {}}
            for mode in ("f32", "bf16"):
                shots = []
                for group in (1, 4):
                    pipe = MelToWavePipeline(post, lambda m, mode=mode: eng.forward(m, dtype=mode), device=dev,
                                             chunk_frames=chunk, group_chunks=group, config=cfg)
                    for _ in range(2):
                        for c in pipe.stream(mel4):
                            pass
                        eng.forward(pipe.refine(mel4), dtype=mode)
                    firsts, totals = [], []
                    for _ in range(5):
                        torch.cuda.synchronize(dev)
                        t0 = time.perf_counter()
                        for i, c in enumerate(pipe.stream(mel4)):
                            if i == 0:
                                torch.cuda.synchronize(dev)                   # the first audio is usable here
                                firsts.append(time.perf_counter() - t0)
                        torch.cuda.synchronize(dev)
                        totals.append(time.perf_counter() - t0)
                        t0 = time.perf_counter()
                        eng.forward(pipe.refine(mel4), dtype=mode)
                        torch.cuda.synchronize(dev)
                        shots.append(time.perf_counter() - t0)
                    first, total, shot = (statistics.median(v) for v in (firsts, totals, shots))
                    rec = {"first_chunk_ms": 1e3 * first, "total_ms": 1e3 * total, "one_shot_ms": 1e3 * shot,
                           "samples_per_s": T4 * hop / total, "rtf": total / (T4 * hop / SAMPLE_RATE),
                           "streaming_over_one_shot": total / shot, "group_chunks": group}
                    if group == 1:
                        c4["modes"][mode] = rec
                    else:
                        c4["modes"][mode]["group_chunks_4"] = rec
            c4["note"] = ("group_chunks = 1: every 256-frame chunk is its own window (+ halo); group_chunks_4: the first chunk alone "
                          "(same time to first audio), then up to four consecutive chunks as ONE merged window (one halo per group): "
                          "iris/streaming.py")
            out["configs4"] = c4
            del mel4, post
        except Exception as exc:
            out["configs4"] = {"error": str(exc)}
        # ---- the one-GPU leg of configs[3]: batch 256 x 1000 frames fp32 (4 passes over sub-batches, 15 GB of workspace) ----
        try:
            m3 = torch.from_numpy(seeded_mel(1004, GLOBAL_BATCH, 1000)).to(dev)
            ms3, _, _ = timed_forward(eng, m3, "f32", 3, 1, dev, False, cfg)
            out["configs3_n1"] = {"workload": f"global batch {GLOBAL_BATCH} x 80-mel x 1000 frames on ONE GPU, fp32 (the N = 1 leg of "
                                              "BASELINE.json configs[3])", "dtype": "f32", "ms_per_step": ms3, "steps": 3,
                                  "samples_per_s": GLOBAL_BATCH * 1000 * hop / (ms3 * 1e-3),
                                  "tflops_whole_path": flop_per_frame * GLOBAL_BATCH * 1000 / (ms3 * 1e-3) / 1e12}
            del m3
        except Exception as exc:
            out["configs3_n1"] = {"error": str(exc)}
        eng.release_workspace()                                           # the 15 GB, before the CPU leg
        torch.cuda.empty_cache()

    if extras:
        out["load"] = cold_start_record(sd)                               # fresh child process; outside every timed region
    if world == 1 and not stub and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, sd, mel_np)
        if extras:
            # Beside the headline (outside its timed region): the same workload in the other arithmetic modes of the
            # library, each with its max-abs distance to the CPU checker's waveform.  `value` above is always --dtype.
            ref = cpu_baseline.last_output
            modes = []
            time.sleep(1.0)                     # let the CPU leg's worker threads go idle (they slow the launching thread)
            for mode in ("f32", "f32s", "bf16"):
                try:
                    ms, _, w = timed_forward(eng, mel, mode, 10, 5, dev, False, cfg)
                    modes.append({"dtype": mode, "ms_per_step": ms, "value": B * T * hop / (ms * 1e-3), "unit": "samples/s",
                                  "max_abs_err_vs_cpu_checker": float(np.abs(w.cpu().numpy() - ref).max()),
                                  "note": {"f32": "exact fp32 MFMA (the parity path; no launch events here)",
                                           "f32s": "fp32 storage/accumulate, split-bf16 products in the ResBlock convs (opt-in)",
                                           "bf16": "bf16 storage, fp32 accumulate (opt-in; tolerance unpinned by the reference)"}[mode]})
                except Exception as exc:      # a mode the configuration does not support
                    modes.append({"dtype": mode, "error": str(exc)})
            out["other_modes"] = modes
    else:
        out["cpu_baseline"] = None
    emit(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.cold_start_child:
        cold_start_child(args.cold_start_child)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))           # no GPU call has happened in this process
    rank_main(args)


if __name__ == "__main__":
    main()
