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


def test_live_pmc_passes_are_three_separate_counter_runs_of_the_same_command(monkeypatch, tmp_path):
    """bench.live_traffic (round 6: roofline.traffic counted in the bench run): three child runs of THIS command under
    `rocprofv3 --pmc <one counter group> --kernel-trace` -- never combined with another trace domain (the GPU pool refuses
    that), each bounded by a timeout, the parent's own --pmc choice not passed on, no epoch / CPU / setup legs, the clock the
    parent tuned; a pass that fails or times out makes the caller fall back (None), nothing is left in TMPDIR."""
    import sqlite3
    import bench
    calls = []
    exe = tmp_path / "rocprofv3"
    exe.write_text("#!/bin/sh\nexit 0\n")
    exe.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    values = {"FETCH_SIZE": 500000.0, "WRITE_SIZE": 60000.0, "TCC_HIT_sum": 3.0e7, "TCC_MISS_sum": 1.0e7}

    def fake_run(cmd, cwd=None, env=None, stdout=None, stderr=None, timeout=None):
        calls.append((list(cmd), cwd, dict(env), timeout))
        out = cmd[cmd.index("-d") + 1]
        os.makedirs(os.path.join(out, "host"))
        db = sqlite3.connect(os.path.join(out, "host", "1_results.db"))
        db.execute("create table pmc_events (name text, counter_name text, counter_value real, duration real)")
        pmc = cmd[cmd.index("--pmc") + 1:cmd.index("--kernel-trace")]
        for c in pmc:
            for _ in range(4):
                db.execute("insert into pmc_events values (?, ?, ?, ?)",
                           ("void sgcn::cs_spmm16g2k_kernel<4, false, false>(sgcn::CsArgs)", c, values[c], 312000.0))
            db.execute("insert into pmc_events values (?, ?, ?, ?)", ("void sgcn::spmm_seg_kernel<64, 3, 4, 4>(sgcn::SpmmArgs)", c, 9e9, 7.6e6))
        db.commit()
        db.close()
        return subprocess.CompletedProcess(cmd, 0)
    monkeypatch.setattr(subprocess, "run", fake_run)
    got = bench.live_traffic(["--gpus", "1", "--steps", "20", "--pmc", "live", "--warmup", "3"], 195, 192, "void sgcn::cs_spmm16", 10)
    assert got is not None and len(calls) == 3
    groups = [c[0][c[0].index("--pmc") + 1:c[0].index("--kernel-trace")] for c in calls]
    assert groups == [["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"]]
    for cmd, cwd, env, timeout in calls:
        assert os.path.basename(cmd[0]) == "rocprofv3" and cwd == "/tmp" and env["TMPDIR"] == "/tmp" and env["SGCN_BENCH_CHILD"] == "1"
        assert 0 < timeout <= 120
        head, child = cmd[:cmd.index("--")], cmd[cmd.index("--") + 1:]
        assert not any(f in head for f in ("-s", "--sys-trace", "-r", "--runtime-trace", "--hip-trace", "--hsa-trace",
                                           "--memory-copy-trace", "--scratch-memory-trace", "--marker-trace", "-i"))
        assert child[1].endswith("bench.py") and child.count("--pmc") == 1 and child[child.index("--pmc") + 1] == "off"
        assert "live" not in child and {"--no-epoch", "--no-cpu-baseline", "--no-setup-report"} <= set(child)
        assert child[-2:] == ["--tune", "cs_pace=195"] and "--gpus" in child
    # the timed kernel = the cs_spmm16 instantiation, not the row-gather kernel the same run launches; gfx950's x 2 on FETCH
    assert got["kernel"].startswith("void sgcn::cs_spmm16g2k") and got["fetch_bytes_corrected"] == 500000.0 * 1024 * 2
    assert got["write_bytes"] == 60000.0 * 1024 and abs(got["l2_hit_rate"] - 0.75) < 1e-12
    assert got["hbm_bytes_per_spmm"] == (got["fetch_bytes_corrected"] + got["write_bytes"]) * 10
    assert not [p for p in os.listdir(tmp_path) if p.startswith("sgcn_pmc_")]
    # a failing pass -> None (the caller keeps the committed record)
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: subprocess.CompletedProcess(a[0], 1))
    assert bench.live_traffic(["--gpus", "1"], 195, 192, "void sgcn::cs_spmm16", 10) is None

    def hang(*a, **k):
        raise subprocess.TimeoutExpired(a[0], k.get("timeout"))
    monkeypatch.setattr(subprocess, "run", hang)
    assert bench.live_traffic(["--gpus", "1"], 195, 192, "void sgcn::cs_spmm16", 10) is None
