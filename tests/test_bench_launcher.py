"""CPU: `bench.py --gpus N` really runs N ranks.  With no torchrun environment it spawns them itself (one process
per rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), under torch.distributed.run it is one of N.  --dry-run
keeps every piece of launcher plumbing real (spawn, gloo rendezvous, barrier, max-reduce, all_gather, the
block-range sharding of the 10 B-integer column) and skips only the GPU work -- its line says so and carries no
number, so it can never be mistaken for a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), stdout     # stdout carries ONE line (rank 0's), nothing else --
    #                                                                 not even gloo's "[Gloo] Rank 0 is connected ..." chatter
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 8])
def test_self_spawn_runs_n_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 0, r.stderr
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == n and out["dry_run"] is True and out["value"] is None
    assert out["ranks"] == list(range(n))
    assert out["blocks_per_rank"] == [10_000_000] * n                     # weak-scaled headline: 10 M blocks per GPU
    c5 = out["config5_strong"]["per_rank"]                                # strong-scaled 10 B-integer column
    assert [p["rank"] for p in c5] == list(range(n))
    assert sum(p["blocks"] for p in c5) == 9_765_625
    assert all(c5[i]["first_block"] + c5[i]["blocks"] == c5[i + 1]["first_block"] for i in range(n - 1))
    if n == 8:
        assert [p["blocks"] for p in c5] == [1_220_704] + [1_220_703] * 7


def test_under_torch_distributed_run():
    """The driver's launch line: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH,
                        "--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 0, r.stderr
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["ranks"] == [0, 1]


def test_a_failing_rank_fails_the_launch():
    """No GPU here: without --dry-run every rank must exit loudly (there is no CPU path) and the launcher must
    return non-zero instead of hanging on the survivors."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr


def test_rccl_failure_falls_back_to_gloo():
    """The data path needs no collective, so RCCL must never be a single point of failure: with no GPU here the RCCL
    probe (a throw-away child per rank) fails, every rank agrees over gloo, and the run completes on gloo saying so."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--probe-nccl", "--nccl-probe-timeout", "30",
                        "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 0, r.stderr
    out = _one_json_line(r.stdout)
    assert out["control_backend"] == "gloo"
    assert "RCCL probe" in out["control_fallback_reason"]
    assert [p["correct"] for p in out["per_rank"]] == [True, True]


def test_one_mismatching_rank_fails_every_rank():
    """per_rank[i].correct is gathered from every rank and any False makes the launch exit non-zero."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--dry-run", "--inject-mismatch", "1",
                        "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode != 0
    out = _one_json_line(r.stdout)
    assert [p["correct"] for p in out["per_rank"]] == [True, False]
    assert "MISMATCH" in out["correctness"] and "1" in out["correctness"]
