"""N > 1 path on CPU: batch sharding + waveform gather over ``gloo`` with world_size 2 and 3.

The GPU engine is replaced by a deterministic stand-in (item b of the batch -> a ramp tagged with
b): what is under test is the partition, the gather order and the uneven/empty-shard handling of
``iris.distributed``, which is exactly what runs over RCCL on the GPUs.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from iris.distributed import gather_waveforms, shard_bounds, shard_range, vocode_sharded


def test_shard_bounds_partition():
    for n in (0, 1, 5, 8, 256, 257):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and len(b) == w
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes
    assert shard_bounds(256, 8) == [(32 * r, 32 * r + 32) for r in range(8)]      # BASELINE configs[3]
    assert shard_range(5, 2, 3) == (4, 5)
    with pytest.raises(ValueError):
        shard_bounds(4, 0)


def _fake_forward(mel):  # [n, 80, T] -> [n, 256*T]; item id is carried in mel[:, 0, 0]
    n, _, t = mel.shape
    ramp = torch.arange(256 * t, dtype=torch.float32)[None, :] / (256 * t)
    return mel[:, 0, 0][:, None] + ramp.expand(n, -1)


def _worker(rank, world, port, n_items, frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n_items, rank, world)
        mel = torch.zeros((hi - lo, 80, frames))
        mel[:, 0, 0] = torch.arange(lo, hi, dtype=torch.float32)
        out = vocode_sharded(_fake_forward, mel, n_items)
        expect = _fake_forward(torch.cat([torch.full((1, 80, frames), float(i)) for i in range(n_items)]))
        ok = out.shape == (n_items, 256 * frames) and torch.equal(out, expect)
        # preallocated output + direct gather call
        out2 = torch.empty_like(expect)
        gather_waveforms(_fake_forward(mel), n_items, out=out2)
        ok = ok and torch.equal(out2, expect)
        # a wrongly sized shard is rejected on every rank before any collective is issued
        try:
            gather_waveforms(torch.zeros((hi - lo + 1, 4)), n_items)
            ok = False
        except ValueError:
            pass
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_items", [(2, 4), (2, 5), (3, 2)])
def test_sharded_vocode_and_gather_gloo(world, n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def test_gather_is_identity_without_process_group():
    w = torch.arange(6.0).reshape(2, 3)
    assert gather_waveforms(w, 2) is w
