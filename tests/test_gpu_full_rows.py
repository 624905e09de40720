"""BASELINE's full size (65,536 envs, config 3) for the rows around the tick: SimpleAgent games with fresh boards and the
observation export at full size, checked against the oracle on slices (envs are independent: a slice of the batch must equal
the oracle's run of the same global env indices) and through size-independent properties."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 65536


@pytest.mark.gpu
def test_config3_full_size_policy_games_on_generated_boards(hip_lib, oracle):
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    seed, bseed, cap, ticks = 9, 2024, 50, 120
    env = BatchEnvironment(N, mode=MODE_ENV, auto_reset=True, max_steps=cap, fresh_boards=True, board_seed=bseed)
    env.generate(bseed)
    env.step_simple(seed, ticks)
    got = env.get_state()
    eps = env.episodes()
    cnt = env.counters()
    assert cnt[0] == N * ticks and cnt[2] == eps.sum()          # every env stepped every tick; resets = games started
    assert eps.min() >= 2                                        # 120 ticks under a 50-tick cap
    assert np.all(got["timeStep"] <= cap) and np.all(got["timeStep"] >= 0)
    # slices from the start, the middle (not wavefront-aligned) and the end of the batch, against the oracle
    for first, m in ((0, 512), (30001, 700), (N - 300, 300)):
        ref = oracle.boardgen(bseed, first + np.arange(m), np.zeros(m))
        e = np.zeros(m, dtype=np.int32)
        mems = np.zeros((m, 4, 16), dtype=np.int32)
        oracle.run_simple_fresh(ref, e, mems, ticks, seed, bseed, first, 0, cap)
        assert got[first:first + m].tobytes() == ref.tobytes(), first
        assert np.array_equal(eps[first:first + m], e)
        assert np.array_equal(env.policy_memory(first, m), mems)
    # determinism: a second handle, different stream split, same result
    env2 = BatchEnvironment(N, mode=MODE_ENV, auto_reset=True, max_steps=cap, fresh_boards=True, board_seed=bseed, streams=1)
    env2.generate(bseed)
    env2.step_simple(seed, ticks)
    assert env2.get_state().tobytes() == got.tobytes()
    env.close()
    env2.close()


@pytest.mark.gpu
def test_observation_full_size_slices_and_plane_invariants(hip_lib):
    spec = importlib.util.spec_from_file_location("pom_observe_oracle", os.path.join(ROOT, "oracle", "pom_observe_oracle.py"))
    ob = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ob)
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    env = BatchEnvironment(N, mode=MODE_ENV, auto_reset=True, max_steps=800)
    env.generate(5)
    env.step_random(3, 1, ticks=40)
    planes, attrs, eattrs = env.observe()
    p = planes.cpu().numpy()
    st = env.get_state()
    for first, m in ((0, 300), (41111, 300), (N - 200, 200)):
        want, want_attrs, _ = ob.observe(st[first:first + m])
        assert np.array_equal(p[first:first + m], want) and np.array_equal(attrs.cpu().numpy()[first:first + m], want_attrs)
    # every cell shows exactly one item plane (no fog on these boards); agent planes hold each live agent once
    assert np.all(p[:, :12].sum(axis=1) == 1)
    alive = attrs.cpu().numpy()[:, :, 2]
    assert np.array_equal(p[:, 8:12].reshape(N, 4, -1).sum(axis=2), alive)
    assert np.array_equal(eattrs.cpu().numpy()[:, 1], alive.sum(axis=1))
    # the per-agent view is the global one with the agent planes rotated
    v = env.observe(per_agent=True, attrs=False)[0].cpu().numpy()
    for a in range(4):
        order = list(range(8)) + [8 + ((a + j) & 3) for j in range(4)] + list(range(12, 16))
        assert np.array_equal(v[:, a], p[:, order])
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fresh", [False, True])
def test_long_horizon_random_play(hip_lib, oracle, fresh):
    """3,000 ticks (the 800-tick cap is reached, every env plays dozens of games): rare paths get their turn"""
    import pomcpp_amd as pa
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    n, seed, bseed, cap, offset = 2048, 17, 4, 800, 10_000
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=cap, env_offset=offset, fresh_boards=fresh, board_seed=bseed)
    if fresh:
        env.generate(bseed)
        ref = oracle.boardgen(bseed, offset + np.arange(n), np.zeros(n))
    else:
        start = pa.make_boards(n, seed=3)
        env.make_game(start)
        ref = start.copy()
    eps = np.zeros(n, dtype=np.int32)
    tick = 0
    for chunk in (1000, 1000, 1000):
        env.step_random(seed, 1, ticks=chunk)
        if fresh:
            oracle.run_random_fresh(ref, eps, chunk, seed, bseed, offset, tick, 1, cap)
        else:
            oracle.run_random(ref, start, chunk, seed, offset, tick, 1, cap)
        tick += chunk
        assert env.get_state().tobytes() == ref.tobytes(), (fresh, tick)
    if fresh:
        assert np.array_equal(env.episodes(), eps) and eps.min() > 20
    env.close()


@pytest.mark.gpu
def test_long_horizon_simple_agents_with_fresh_boards(hip_lib, oracle):
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    n, seed, bseed, cap, offset = 1024, 23, 6, 800, 777
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=cap, env_offset=offset, fresh_boards=True, board_seed=bseed)
    env.generate(bseed)
    ref = oracle.boardgen(bseed, offset + np.arange(n), np.zeros(n))
    eps, mems = np.zeros(n, dtype=np.int32), np.zeros((n, 4, 16), dtype=np.int32)
    tick = 0
    for chunk in (400, 400, 400):
        env.step_simple(seed, chunk)
        oracle.run_simple_fresh(ref, eps, mems, chunk, seed, bseed, offset, tick, cap)
        tick += chunk
        assert env.get_state().tobytes() == ref.tobytes(), tick
        assert np.array_equal(env.policy_memory(), mems), tick
    assert np.array_equal(env.episodes(), eps) and eps.min() >= 1 and eps.mean() > 4  # 1,200 ticks: a game lasts ~190
    env.close()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_launcher_runs_two_real_ranks_on_one_gpu(hip_lib):
    """`python bench.py --gpus 2` as the driver issues it, on a one-GPU box: POM_BENCH_REHEARSAL lets the two ranks share the device
    (gloo instead of RCCL).  The real worker runs in both ranks: shard plan, env_offset, stepping, counter all-reduce, max-over-ranks
    timing, the JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, POM_BENCH_REHEARSAL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--envs", "8192", "--steps", "12", "--warmup", "3",
                          "--burn-in", "40", "--no-cpu-baseline", "--no-config3"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["ranks"] == 2 and r["config"]["rccl_ranks"] == 2 and r["config"]["collective_backend"] == "gloo"
    assert r["config"]["global_envs"] == 16384 and r["scaling"] == "weak" and "REHEARSAL" in r["data"]
    assert "all-reduce" in r["config"]["collective_in_timed_region"]  # issued between the region's barriers (bench.timed_region)
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 - 16384) < 1.0  # value = all ranks' env-steps over the slowest rank's time
    base = r["config"]["single_gpu_base"]  # rank 0's shard stepped alone, in the same run: the like-for-like base of a scaling curve
    assert base and base["value"] > 0 and base["steps"] == 12


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_multi_rank_path_under_rccl_with_one_rank(hip_lib):
    """The N > 1 code of bench.py under RCCL itself (the rehearsal above runs it over gloo): POM_BENCH_RCCL_SOLO makes a one-rank
    nccl communicator and takes every `several ranks` branch — communicator warm-up, the stream vote, barriers, the counters'
    all-reduce on a side stream inside the timed region.  And the region must not pay for first uses: with 65,536 envs its steps run
    ~12 us each; the all-reduce and the closing barrier may add tens of microseconds to the whole region, not hundreds per step."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    with socket.socket() as sk:  # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, POM_BENCH_RCCL_SOLO="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-config3"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["config"]["collective_backend"] == "nccl" and r["config"]["rccl_ranks"] == 1
    assert "all-reduce" in r["config"]["collective_in_timed_region"]
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 - 65536) < 1.0
    base = r["config"]["single_gpu_base"]
    assert base and base["steps"] == 20
    assert r["ms_per_step"] < 2.0 * base["ms_per_step"], (r["ms_per_step"], base["ms_per_step"])  # (it was 2.9 x before the side stream was warmed)
