"""`python bench.py --gpus N` must start N ranks by itself (--gpus was parsed and
ignored, so a scaling run would have measured one GPU N times).  CPU test: the launcher path
under SGCN_DIST_BACKEND=gloo with --dry-run (the rendezvous / collective skeleton of the run
without kernels), and the refusal paths."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_2_spawns_two_distinct_ranks_that_share_an_allreduce(monkeypatch, capfd):
    import bench
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SGCN_DIST_BACKEND", "gloo")
    line = bench.main(["--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run"])
    assert line["n_gpus"] == 2 and line["dry_run"] is True
    assert sorted(line["ranks"]) == [0, 1]                  # two distinct ranks met in the all-gather
    assert line["allreduce_of_rank_plus_1"] == 3.0          # 1 + 2: both contributed to the all-reduce
    # the strong-scaling record rides in the SAME line: keys present, and its partition + all-gather
    # skeleton (one graph, nnz-balanced row blocks) ran over the two ranks
    st = line["strong"]
    assert {"resident_ms", "allgather_ms", "edges_per_s"} <= set(st) and st["allgather_ok"] is True
    assert st["rows"][0] == 0 and 0 < st["rows"][1] < 1024 and abs(st["local_nnz"] - st["nnz"] / 2) < 0.1 * st["nnz"]
    out = capfd.readouterr().out
    printed = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(printed) == 1 and printed[0]["n_gpus"] == 2   # ONE line, from rank 0


def test_gpus_8_dry_run_meets_eight_ranks(monkeypatch, capfd):
    """The driver's widest launch (`--gpus 8`) over gloo: eight distinct ranks, one line, an eight-way strong partition."""
    import bench
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SGCN_DIST_BACKEND", "gloo")
    line = bench.main(["--gpus", "8", "--steps", "2", "--warmup", "0", "--dry-run"])
    assert line["n_gpus"] == 8 and sorted(line["ranks"]) == list(range(8))
    assert line["allreduce_of_rank_plus_1"] == 36.0
    st = line["strong"]
    assert st["allgather_ok"] is True and abs(st["local_nnz"] - st["nnz"] / 8) < 0.2 * st["nnz"]
    printed = [json.loads(l) for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert len(printed) == 1 and printed[0]["n_gpus"] == 8


def test_gpus_2_without_two_gpus_fails_loudly():
    """On a box with fewer than N GPUs (this container has none) the nccl launcher must refuse --
    never run one rank and print n_gpus: 1."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                         env=_env(SGCN_DIST_BACKEND="nccl"), capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert "--gpus 2 requested but only" in res.stderr
    assert '"n_gpus"' not in res.stdout


def test_world_size_mismatch_is_refused():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"],
                         env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True,
                         timeout=300)
    assert res.returncode != 0 and "refusing" in (res.stderr + res.stdout)
    assert '"n_gpus"' not in res.stdout
