#!/usr/bin/env python3
"""Headline benchmark of the HiFiGAN vocoder path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--batch B --frames T]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path -- mel [B, 80, T] -> waveform [B, 256*T] through
``iris_hifigan_forward`` -- over one batch of synthetic mels that is already resident in HBM when
the timed region starts.  N = 1 runs BASELINE.json configs[1] (batch 1, 80 x 1000 frames, fp32).
For N > 1 every rank vocodes its own shard of the same size (weak scaling, no data-path collective
but the final RCCL all-gather of the waveforms, which IS inside the step).

The JSON line also carries
  roofline      the dominant kernel (the MFMA Conv1d kernel behind the MRF ResBlocks): algorithmic
                FLOP of its launches / their HIP-event durations, measured live in the timed region
                on the stream the kernels run on; peak = 157.3 TFLOP/s (fp32 MFMA, MI355X_MICROARCH.md).
                The path is a dense contraction at 118 FLOP/B: fp32 is MFMA-bound, not HBM-bound
                (SURVEY.md 8d), so ``bound`` is "mfma"; the HBM fraction of the same launches is
                reported beside it as ``hbm_frac``.
  cpu_baseline  the oracle (torch fp32 restatement of the reference, oracle/hifigan_oracle.py) timed on
                this box's host cores on a bounded sample of the same workload; rank 0, N = 1 only.
"""
import argparse
import json
import os
import statistics
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


def committed_traffic(batch, frames, dtype="f32"):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/
    (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, gfx950 correction applied:
    profiles/r*_hbm_traffic.json, produced by tools/hbm_traffic.sh).  PMC counters cannot be collected from
    inside this process, so the figure is the committed measurement of the same workload, or None when the
    workload differs."""
    if (dtype, batch, frames) == ("f32", 1, 1000):
        files = [f for f in sorted((REPO / "profiles").glob("r*_hbm_traffic.json")) if "bf16" not in f.name]
    elif (dtype, batch, frames) == ("bf16", 32, 500):
        files = sorted((REPO / "profiles").glob("r*_bf16_c3_hbm_traffic.json"))
    else:
        files = []
    if not files:
        return None
    try:
        return json.loads(files[-1].read_text())["mrf_traffic_bytes_per_launch"]
    except (KeyError, ValueError):
        return None


def cpu_baseline(cfg, sd, mel, budget_s=20.0):
    """Times the oracle's torch-fp32 forward (the arithmetic the reference's PyTorch twin runs) on the
    host cores.  Bounded: 1 warm-up on a short clip, then whole utterances until ~budget_s is spent
    (at least 1, at most 3)."""
    import torch
    from oracle import hifigan_oracle as orc

    folded = orc.to_torch_folded(sd)
    cores = torch.get_num_threads()
    x = torch.from_numpy(mel)
    orc.generator_forward_torch(folded, x[:, :, :50])            # warm-up (allocator, thread pool)
    times, t_start = [], time.perf_counter()
    while len(times) < 3 and (not times or time.perf_counter() - t_start + times[-1] < budget_s):
        t0 = time.perf_counter()
        out_ref = orc.generator_forward_torch(folded, x)
        times.append(time.perf_counter() - t0)
    best = statistics.median(times)
    cpu_baseline.last_output = out_ref.numpy()[:, 0, :]       # the checker's waveform of this mel (parity of the GPU modes)
    samples = mel.shape[0] * mel.shape[2] * 256
    return {"value": samples / best, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} full forward(s) of the same B={mel.shape[0]} x 80 x {mel.shape[2]} mel after one "
                      f"50-frame warm-up; median {best:.3f} s; torch {torch.__version__} CPU fp32, "
                      f"{cores} threads; oracle/hifigan_oracle.py:generator_forward_torch",
            "rtf": best / (mel.shape[2] * 256 / SAMPLE_RATE) if mel.shape[0] == 1 else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="mels per GPU (default: configs[1], batch 1)")
    ap.add_argument("--frames", type=int, default=1000, help="mel frames per item")
    ap.add_argument("--dtype", choices=["f32", "bf16", "f32s"], default="f32",
                    help="f32 = the parity path and the headline (default); bf16 = bf16 storage + bf16 MFMA "
                         "(BASELINE.json configs[2] with --batch 32 --frames 500)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket launches with HIP events")
    ap.add_argument("--graph", action="store_true", help="replay the forward as a hipGraph (no per-launch records: "
                    "roofline is then null; the default eager mode is the measured configuration)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and run the collective "
                    "path (barrier, all-gather, max-reduce) even with one rank: exercises the RCCL code path of N > 1 "
                    "on a one-GPU box")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; "
                    "gloo only for rehearsing the control flow on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from iris._engine import GeneratorEngine, algorithmic_work
    from iris._weights import GeneratorConfig, seeded_mel, seeded_state_dict
    from iris.distributed import gather_waveforms

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with "
                             f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
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
    B, T = args.batch, args.frames
    mel_np = seeded_mel(1002 + rank, B, T)                       # SURVEY.md 8d seeds
    eng = GeneratorEngine(cfg, sd, dev)
    mel = torch.from_numpy(mel_np).to(dev)
    wav = torch.empty((B, T * eng.hop_length), dtype=torch.float32, device=dev)
    gathered = torch.empty((B * world, T * eng.hop_length), dtype=torch.float32, device=dev) if use_dist else None

    def step():
        if args.graph:
            out = eng.forward_graph(mel, dtype=args.dtype)
        else:
            out = eng.forward(mel, out=wav, dtype=args.dtype)
        if use_dist:
            gather_waveforms(out, B * world, out=gathered)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    profile = not args.no_profile and not args.graph
    fence()
    eng.set_profiling(profile)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    recs = eng.read_profile() if profile else []
    eng.set_profiling(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_step = world * B * T * eng.hop_length
    value = samples_per_step * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel, from the live HIP-event records -----------------------
    roofline = None
    detail = {}
    if recs:
        by_kind = {}
        for r in recs:
            k = by_kind.setdefault(r["kind"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0})
            k["ms"] += r["ms"]; k["flops"] += r["flops"]; k["bytes"] += r["bytes"]; k["n"] += 1
        for kind, k in by_kind.items():
            detail[kind] = {"launches_per_step": k["n"] // args.steps, "ms_per_step": k["ms"] / args.steps,
                            "tflops": k["flops"] / (k["ms"] * 1e-3) / 1e12, "gbs": k["bytes"] / (k["ms"] * 1e-3) / 1e9}
        for stage in range(cfg.num_upsamples):
            rs = [r for r in recs if r["kind"] == "mrf_resblock_conv" and r["stage"] == stage]
            ms = sum(r["ms"] for r in rs)
            detail[f"mrf_stage{stage}_C{cfg.stage_channels(stage)}"] = {
                "ms_per_step": ms / args.steps, "tflops": sum(r["flops"] for r in rs) / (ms * 1e-3) / 1e12,
                "gbs": sum(r["bytes"] for r in rs) / (ms * 1e-3) / 1e9}
        dom = by_kind["mrf_resblock_conv"]
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        if args.dtype in ("f32", "f32s"):
            # f32s: three bf16 MFMAs per fp32 product -> the matrix roof for fp32-equivalent FLOP is a third of the bf16 peak
            peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS / 3.0
            roofline = {"kernel": ("mrf_conv_mfma_f32_kernel" if args.dtype == "f32" else "conv_mfma_f32s_kernel (split-bf16 products)")
                                  + " (MRF ResBlock Conv1d steps, 24 launches/forward)",
                        "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                        "frac": achieved / peak, "traffic": committed_traffic(B, T) if args.dtype == "f32" else None,
                        "avg_launch_ms": dom["ms"] / dom["n"], "flop_per_launch": dom["flops"] / dom["n"],
                        "hbm_achieved_gbs": gbs, "hbm_peak_gbs": PEAK_HBM_GBS, "hbm_frac": gbs / PEAK_HBM_GBS,
                        "bytes_per_launch": dom["bytes"] / dom["n"],
                        "share_of_step": dom["ms"] / args.steps / ms_per_step}
        else:
            # bf16: the MRF launches taken together need more HBM time (bytes / 8 TB/s) than MFMA time
            # (FLOP / 2.5 PFLOP/s) -- 235 FLOP/B against a machine balance of 312 -- so HBM is the binding roof
            roofline = {"kernel": "conv_mfma_bf16_kernel (MRF ResBlock Conv1d steps, 24 launches/forward)",
                        "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": gbs / PEAK_HBM_GBS, "traffic": committed_traffic(B, T, "bf16"),
                        "avg_launch_ms": dom["ms"] / dom["n"], "bytes_per_launch": dom["bytes"] / dom["n"],
                        "flop_per_launch": dom["flops"] / dom["n"],
                        "mfma_achieved_tflops": achieved, "mfma_peak_tflops": PEAK_BF16_MFMA_TFLOPS,
                        "mfma_frac": achieved / PEAK_BF16_MFMA_TFLOPS,
                        "share_of_step": dom["ms"] / args.steps / ms_per_step}

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    work = algorithmic_work(cfg)
    out = {
        "metric": "audio samples/sec (22.05 kHz) on 80-mel x %d-frame batch" % T,
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"HiFiGAN-V1 generator, batch {B} per GPU x 80-mel x {T} frames -> {T * eng.hop_length} samples "
                               f"each, " + {"f32": "fp32", "bf16": "bf16 storage / fp32 accumulate",
                                            "f32s": "fp32 storage, split-bf16 products in the ResBlock convs"}[args.dtype]
                               + (" (BASELINE.json configs[1])" if (B, T, args.dtype) == (1, 1000, "f32") else "")
                               + (" (BASELINE.json configs[2])" if (B, T, args.dtype) == (32, 500, "bf16") else ""),
                   "batch_per_gpu": B, "global_batch": B * world, "frames": T, "hop_length": eng.hop_length,
                   "weights": "random-init (seeded) V1 architecture, 13,926,017 values",
                   "sharding": (f"batch items across ranks, all-gather of waveforms over {args.backend}"
                                + (" (RCCL)" if args.backend == "nccl" else " (control-flow rehearsal, not a measurement)"))
                               if world > 1 else "single GPU",
                   "launch_events_in_timed_region": profile, "hipgraph_replay": bool(args.graph)},
        "rtf": (ms_per_step * 1e-3) / (T * eng.hop_length / SAMPLE_RATE) if B == 1 else None,
        "flop_per_step": work["flop_per_frame"] * B * T * world,
        "tflops_whole_path": work["flop_per_frame"] * B * T * world / (ms_per_step * 1e-3) / 1e12,
        "roofline": roofline, "kernels": detail,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, sd, mel_np)
        # Beside the headline (outside its timed region): the same workload in the other arithmetic modes of the
        # library, each with its max-abs distance to the CPU checker's waveform.  `value` above is always --dtype.
        ref = cpu_baseline.last_output
        modes = []
        time.sleep(1.0)                     # let the CPU leg's worker threads go idle (they slow the launching thread)
        for mode in ("f32", "f32s", "bf16"):
            try:
                for _ in range(5):
                    w = eng.forward(mel, dtype=mode)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(10):
                    w = eng.forward(mel, dtype=mode)
                torch.cuda.synchronize(dev)
                dt = (time.perf_counter() - t0) / 10
                modes.append({"dtype": mode, "ms_per_step": 1e3 * dt, "value": B * T * eng.hop_length / dt, "unit": "samples/s",
                              "max_abs_err_vs_cpu_checker": float(np.abs(w.cpu().numpy() - ref).max()),
                              "note": {"f32": "exact fp32 MFMA (the parity path; no launch events here)",
                                       "f32s": "fp32 storage/accumulate, split-bf16 products in the ResBlock convs (opt-in)",
                                       "bf16": "bf16 storage, fp32 accumulate (opt-in; tolerance unpinned by the reference)"}[mode]})
            except Exception as exc:      # a mode the configuration does not support
                modes.append({"dtype": mode, "error": str(exc)})
        out["other_modes"] = modes
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
