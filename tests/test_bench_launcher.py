"""bench.py as the driver runs it for N > 1: ``python bench.py --gpus N`` must start its own ranks (one process per
GPU) BEFORE it makes any GPU call, shard BASELINE.json configs[3]'s global batch over them, gather the waveforms
inside the step and print ONE JSON line.  Rehearsed here on CPU (gloo) with ``--stub-engine``, a stand-in that does
not compute the vocoder: only the launcher, the sharding, the collective and the JSON contract are under test
(the arithmetic is covered by the -m gpu parity tests)."""
import ast
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
BENCH = REPO / "bench.py"


def _run(*argv, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(BENCH), *argv], env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout          # exactly ONE JSON line (rank 0's)
    return json.loads(lines[0])


@pytest.mark.parametrize("world,global_batch,extras", [(2, 256, False), (3, 7, True)])
def test_self_launch_shards_the_global_batch(world, global_batch, extras):
    out = _run("--gpus", str(world), "--backend", "gloo", "--stub-engine", "--steps", "2", "--warmup", "1",
               "--global-batch", str(global_batch), "--frames", "12", *([] if extras else ["--no-extras"]))
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out.stdout)
    assert line["n_gpus"] == world and line["steps"] == 2 and line["warmup"] == 1
    assert line["scaling"] == "strong" and line["higher_is_better"] is True and line["unit"] == "samples/s"
    assert line["config"]["global_batch"] == global_batch
    assert line["config"]["batch_per_gpu"] == -(-global_batch // world)          # rank 0 holds the larger shard
    assert line["stub"] is True and "NOT a measurement" in line["data"]
    # value = whole-job samples per second: global batch x frames x hop x steps / max-over-ranks time
    assert line["value"] == pytest.approx(global_batch * 12 * 256 / (line["ms_per_step"] * 1e-3), rel=1e-9)
    # the line is self-verifying: what the process group itself reported, the collective on its own, every rank's clock
    coll = line["collective"]
    assert coll["backend"] == "gloo" and coll["world_size"] == world and coll["timed_gathers"] == 10
    assert coll["bytes_per_rank"] == -(-global_batch // world) * 12 * 256 * 4
    assert coll["bytes_gathered_per_rank"] == global_batch * 12 * 256 * 4 and coll["gather_only_ms"] > 0
    assert len(line["per_rank_ms"]) == world and max(line["per_rank_ms"]) == pytest.approx(line["ms_per_step"], rel=1e-6)
    if extras:
        assert [g["frames"] for g in line["grid"]] == [100, 500, 1000]
        assert all(g["global_batch"] == global_batch and g["samples_per_s"] > 0 for g in line["grid"])
    else:
        assert "grid" not in line


def test_launched_by_torchrun_like_the_driver():
    """The driver's own command line for N > 1 (torch.distributed.run sets WORLD_SIZE: no second launcher level)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29577", str(BENCH), "--gpus", "2", "--backend", "gloo",
                          "--stub-engine", "--steps", "1", "--warmup", "0", "--global-batch", "4", "--frames", "8", "--no-extras"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out.stdout)
    assert line["n_gpus"] == 2 and line["config"]["batch_per_gpu"] == 2 and "grid" not in line


def test_failing_rank_fails_the_parent():
    # nccl with the CPU stand-in is refused by every rank: the parent must report the failure, not print a line
    out = _run("--gpus", "2", "--backend", "nccl", "--stub-engine", "--steps", "1", "--warmup", "0")
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_refused():
    out = _run("--gpus", "2", "--stub-engine", "--backend", "gloo", env_extra={"WORLD_SIZE": "3", "RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_parent_makes_no_gpu_call_before_spawning():
    """Static check of bench.py: torch is imported only inside rank_main / the measurement helpers, and main() reaches
    launch_ranks() before rank_main(); launch_ranks itself imports nothing."""
    tree = ast.parse(BENCH.read_text())
    top_imports = {a.name.split(".")[0] for n in tree.body if isinstance(n, ast.Import) for a in n.names}
    top_imports |= {n.module.split(".")[0] for n in tree.body if isinstance(n, ast.ImportFrom) and n.module}
    assert "torch" not in top_imports and "iris" not in top_imports and "numpy" not in top_imports
    funcs = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    for name in ("launch_ranks", "main", "parse_args"):
        assert not [n for n in ast.walk(funcs[name]) if isinstance(n, (ast.Import, ast.ImportFrom))], name
    line_of = {n.func.id: n.lineno for n in ast.walk(funcs["main"]) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name)}
    assert line_of["launch_ranks"] < line_of["rank_main"]
