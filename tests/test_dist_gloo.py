"""The N>1 path of bench.py on CPU: world_size 2 over gloo — contiguous env shards, distinct move-stream
offsets, the single counter all-reduce and the max-over-ranks timing."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = bench.shard_plan(rank, world, 32768)
    counters = torch.tensor([plan["n_envs"] * 10, rank + 1, 0, 0], dtype=torch.int64)
    bench.reduce_counters(counters, dist)
    slowest = bench.reduce_max(1.0 + rank, torch.device("cpu"), dist)
    q.put((rank, plan, counters.tolist(), slowest))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_shard_and_counter_allreduce():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, p0, c0, t0), (r1, p1, c1, t1) = res
    assert p0["first_env"] == 0 and p1["first_env"] == 32768 and p0["global_envs"] == 65536
    assert p0["first_env"] + p0["n_envs"] == p1["first_env"]  # contiguous, disjoint
    assert c0 == c1 == [2 * 32768 * 10, 3, 0, 0]
    assert t0 == t1 == 2.0


def test_single_process_helpers_are_identity():
    sys.path.insert(0, ROOT)
    import bench
    c = torch.tensor([5, 1, 0, 0], dtype=torch.int64)
    assert bench.reduce_counters(c, None).tolist() == [5, 1, 0, 0]
    assert bench.reduce_max(3.5, torch.device("cpu"), None) == 3.5
    assert bench.shard_plan(0, 1, 65536) == {"first_env": 0, "n_envs": 65536, "global_envs": 65536}


# ---- the launcher behind `bench.py --gpus N` (--gpus was parsed and never read) ---------------------------
def _launch(argv, n):
    """bench.launch_ranks in a child interpreter (it relays rank 0's stdout to its own), with the CPU stand-in as the worker"""
    import subprocess
    code = ("import sys, bench; sys.exit(bench.launch_ranks(%d, %r, worker_cmd=[sys.executable, %r], timeout_s=100))"
            % (n, argv, os.path.join(ROOT, "tests", "stub_rank.py")))
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=150, env=env, cwd=ROOT)


@pytest.mark.timeout(200)
def test_launcher_starts_n_ranks_and_relays_rank0_line():
    import json
    out = _launch(["--gpus", "2", "--envs", "32768", "--steps", "10"], 2)
    assert out.returncode == 0, out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0's line only
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["global_envs"] == 65536
    assert r["steps_total"] == 2 * 32768 * 10 and r["slowest"] == 2.0 and r["first_env"] == 0


@pytest.mark.timeout(300)
def test_gpus_8_keeps_the_headline_batch_per_gpu_and_reduces_inside_the_region():
    """the driver's command shape `bench.py --gpus 8 ...`: 65,536 envs per GPU as on one GPU (weak scaling with the per-GPU work
    fixed), and the counter all-reduce sits between the region's two barriers (bench.timed_region, the function the real worker
    times with); BASELINE config 4 (262,144 envs on 8 GPUs) is `--envs 32768`"""
    import json
    out = _launch(["--gpus", "8", "--steps", "10"], 8)
    assert out.returncode == 0, out.stderr
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert r["n_gpus"] == 8 and r["rccl_ranks"] == 8 and r["envs_per_gpu"] == 65536 and r["global_envs"] == 524288
    assert r["steps_total"] == 524288 * 10 and r["slowest"] == 8.0
    assert r["region"] == ["barrier", "steps", "allreduce", "barrier"]
    out = _launch(["--gpus", "8", "--envs", "32768", "--steps", "10"], 8)
    assert out.returncode == 0, out.stderr
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert r["envs_per_gpu"] == 32768 and r["global_envs"] == 262144 and r["steps_total"] == 262144 * 10


def test_default_batch_per_gpu(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.parse_args([]).envs == 65536 and bench.parse_args(["--gpus", "1"]).envs == 65536
    assert bench.parse_args(["--gpus", "8"]).envs == 65536 and bench.parse_args(["--gpus", "2"]).envs == 65536
    assert bench.parse_args(["--gpus", "8", "--envs", "4096"]).envs == 4096
    monkeypatch.setenv("WORLD_SIZE", "8")  # under torch.distributed.run
    assert bench.parse_args(["--gpus", "8"]).envs == 65536


@pytest.mark.timeout(200)
def test_launcher_fails_when_a_rank_fails():
    out = _launch(["--gpus", "2", "--fail-rank", "1"], 2)
    assert out.returncode != 0
    assert "rank 1 exited with 3" in out.stderr


def test_bench_main_becomes_the_launcher_only_without_a_rendezvous(monkeypatch):
    """`python bench.py --gpus N` launches; under torch.distributed.run (WORLD_SIZE set) the same command line is a rank"""
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    monkeypatch.setattr(bench, "launch_ranks", lambda n, argv, **kw: calls.append((n, argv)) or 0)
    monkeypatch.setattr(bench, "worker", lambda args: calls.append(("worker", args.gpus)))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.main(["--gpus", "8", "--envs", "32768"]) == 0
    assert calls == [(8, ["--gpus", "8", "--envs", "32768"])]
    monkeypatch.setenv("WORLD_SIZE", "8")
    bench.main(["--gpus", "8"])
    bench.main([])
    assert calls[1:] == [("worker", 8), ("worker", 1)]
