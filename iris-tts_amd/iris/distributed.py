"""Batch-sharded vocoding across the GPUs of one node (one process per GPU).

Every batch item of the generator is independent (no BatchNorm, no cross-item op anywhere in
``HiFiGANModel.forward``, reference src/iris/hifigan_pretrained.py:123-143), so the path shards
embarrassingly: rank r of R vocodes mels ``[lo_r, hi_r)`` with its own replica of the weights, and
the only exchange is the final gather of the waveform shards -- ``torch.distributed`` with backend
``"nccl"`` (= RCCL over xGMI on ROCm) on GPUs, ``"gloo"`` in the CPU tests.  There is no mel scatter
collective: every rank is handed (or generates) only its own shard.

The reference itself has no distributed code; this module is new.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced partition of ``range(n_items)``: the first ``n_items % world_size``
    ranks get one extra item.  Ranks beyond ``n_items`` get empty shards."""
    if n_items < 0 or world_size < 1:
        raise ValueError("n_items must be >= 0 and world_size >= 1")
    base, extra = divmod(n_items, world_size)
    bounds, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    return shard_bounds(n_items, world_size)[rank]


def gather_waveforms(local_wav: torch.Tensor, n_items: int, group: Optional[dist.ProcessGroup] = None,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All-gathers the per-rank waveform shards ``[n_local, samples]`` into ``[n_items, samples]``
    on every rank, in batch order.  Shards may be uneven (or empty); equal shards use a single
    ``all_gather_into_tensor``, uneven ones are padded to the largest shard."""
    if not dist.is_initialized():
        return local_wav
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = shard_bounds(n_items, world)
    lo, hi = bounds[rank]
    if local_wav.dim() != 2 or local_wav.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: local shard has shape {tuple(local_wav.shape)}, expected [{hi - lo}, samples]")
    samples = local_wav.shape[1]
    if out is None:
        out = torch.empty((n_items, samples), dtype=local_wav.dtype, device=local_wav.device)
    elif out.shape != (n_items, samples):
        raise ValueError("out has the wrong shape")
    sizes = [b - a for a, b in bounds]
    if len(set(sizes)) == 1:
        dist.all_gather_into_tensor(out, local_wav.contiguous(), group=group)
        return out
    biggest = max(sizes)
    padded = torch.zeros((biggest, samples), dtype=local_wav.dtype, device=local_wav.device)
    padded[: hi - lo] = local_wav
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    for (a, b), part in zip(bounds, parts):
        out[a:b] = part[: b - a]
    return out


def vocode_sharded(forward: Callable[[torch.Tensor], torch.Tensor], local_mel: torch.Tensor, n_items: int,
                   group: Optional[dist.ProcessGroup] = None, gather: bool = True) -> torch.Tensor:
    """Runs ``forward`` (e.g. ``GeneratorEngine.forward``) on this rank's mel shard
    ``[n_local, n_mels, T]`` and gathers the waveforms of all ranks."""
    wav = forward(local_mel)
    if gather and dist.is_initialized() and dist.get_world_size(group) > 1:
        return gather_waveforms(wav, n_items, group)
    return wav
