"""GPU: BASELINE.json's full sizes through size-independent properties, and the edge cases of the boundary."""
import hashlib

import numpy as np
import pytest

import pomcpp_amd as pa
from pomcpp_amd.batch import (BatchEnvironment, PomError, MODE_ENV, MODE_RAW, DIST_RANDOM, DIST_STRESS, CNT_STEPS, CNT_EPISODES,
                              CNT_RESETS, CNT_UB_TICKS, UB_LOST_AGENT)
from pomcpp_amd.state import Item, Move

pytestmark = pytest.mark.gpu


def _digest(states):
    s = states.copy()
    s["agents"]["pad"] = 0
    return hashlib.blake2b(s.tobytes(), digest_size=16).hexdigest()


@pytest.mark.parametrize("n,kind,dist", [(65536, "ffa", DIST_RANDOM), (65536, "stress", DIST_STRESS)])
def test_full_size_is_deterministic_residency_invariant_and_matches_oracle_sample(hip_lib, oracle, n, kind, dist):
    """65,536 envs (configs 3/5 size): the result must not depend on how many ticks stay resident in LDS per
    launch, must be reproducible, must account for every env-step, and a 4,096-env slice must equal the oracle."""
    ticks, seed = 96, 1234
    start = pa.make_boards(n, seed=77, kind=kind)
    digests, finals = [], []
    for tpl in (1, 1, 16, 96):
        with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800) as env:
            env.make_game(start)
            env.step_random(seed, dist, ticks=ticks, ticks_per_launch=tpl)
            got = env.get_state()
            cnt = env.counters()
            st = env.status()
        assert cnt[CNT_STEPS] == n * ticks
        # an episode that ended is either still finished at the end or was restarted
        assert cnt[CNT_EPISODES] == cnt[CNT_RESETS] + st["done"].sum()
        digests.append(_digest(got))
        finals.append(got)
    assert len(set(digests)) == 1
    lo = 20000
    want = start[lo:lo + 4096].copy()
    oracle.run_random(want, np.ascontiguousarray(start[lo:lo + 4096]), ticks, seed, lo, 0, dist, 800)
    assert _digest(finals[0][lo:lo + 4096]) == _digest(want)


def test_262144_envs_step_counts_and_shard_equivalence(hip_lib, oracle):
    """Config 4's size on one GPU: stepping the whole batch equals stepping its shards with env_offset — 4 of 65,536 and the 8 of
    32,768 that BASELINE's config 4 names — and slices of it (in the middle, across a shard boundary, inside two of the 8 shards)
    equal the oracle."""
    n, ticks, seed = 262144, 40, 5
    start = pa.make_boards(n, seed=3)
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800) as env:
        env.make_game(start)
        env.step_random(seed, DIST_RANDOM, ticks=ticks)
        whole = env.get_state()
        assert env.counters()[CNT_STEPS] == n * ticks
    for lo, cnt in ((100000, 2048), (3 * (n // 4) - 1000, 2048)):
        want = np.ascontiguousarray(start[lo:lo + cnt]).copy()
        oracle.run_random(want, np.ascontiguousarray(start[lo:lo + cnt]), ticks, seed, lo, 0, DIST_RANDOM, 800)
        assert _digest(whole[lo:lo + cnt]) == _digest(want), lo
    q = n // 4
    for k in (0, 3):
        with BatchEnvironment(q, mode=MODE_ENV, auto_reset=True, max_steps=800, env_offset=k * q) as env:
            env.make_game(start[k * q:(k + 1) * q])
            env.step_random(seed, DIST_RANDOM, ticks=ticks)
            assert _digest(env.get_state()) == _digest(whole[k * q:(k + 1) * q])
    # config 4 as BASELINE states it: 8 shards of 32,768 (`bench.py --gpus 8 --envs 32768`: rank r owns envs [r * 32768, (r + 1) * 32768),
    # env_offset keys its move stream) — every shard, and an oracle slice inside each of two of them
    q8 = n // 8
    assert q8 == 32768
    for k in range(8):
        with BatchEnvironment(q8, mode=MODE_ENV, auto_reset=True, max_steps=800, env_offset=k * q8) as env:
            env.make_game(start[k * q8:(k + 1) * q8])
            env.step_random(seed, DIST_RANDOM, ticks=ticks)
            got = env.get_state()
            assert env.counters()[CNT_STEPS] == q8 * ticks
        assert _digest(got) == _digest(whole[k * q8:(k + 1) * q8]), k
        if k in (2, 7):
            lo = k * q8 + 30000
            want = np.ascontiguousarray(start[lo:lo + 1536]).copy()
            oracle.run_random(want, np.ascontiguousarray(start[lo:lo + 1536]), ticks, seed, lo, 0, DIST_RANDOM, 800)
            assert _digest(got[30000:31536]) == _digest(want), k


def test_1048576_envs_beyond_the_memory_side_cache(hip_lib, oracle):
    """The size of `bench.other_configs.headline_1048576_envs` (470 MB of records + as much of snapshots: every tick streams them from
    and to HBM proper; sub-batches on parallel streams, not chained launches): boards drawn on the device, 60 ticks of random moves with
    auto-reset; three slices — the first tile, one in the middle across a 65,536 boundary, the batch's last envs — equal the oracle, the
    step count is exact, and a 65,536-env shard stepped on a handle of its own with env_offset (chained launches) ends in the same states."""
    n, ticks, seed = 1048576, 60, 9
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800) as env:
        env.generate(21)
        slices = ((0, 1024), (7 * 65536 - 700, 1400), (n - 1024, 1024))
        starts = [env.get_state(lo, cnt) for lo, cnt in slices]
        shard_lo = 11 * 65536
        shard_start = env.get_state(shard_lo, 65536)
        env.step_random(seed, DIST_RANDOM, ticks=ticks)
        assert env.counters()[CNT_STEPS] == n * ticks
        for (lo, cnt), ini in zip(slices, starts):
            want = ini.copy()
            oracle.run_random(want, ini, ticks, seed, lo, 0, DIST_RANDOM, 800)
            assert _digest(env.get_state(lo, cnt)) == _digest(want), lo
        shard_end = env.get_state(shard_lo, 65536)
    with BatchEnvironment(65536, mode=MODE_ENV, auto_reset=True, max_steps=800, env_offset=shard_lo) as env:
        env.make_game(shard_start)
        env.step_random(seed, DIST_RANDOM, ticks=ticks)
        assert _digest(env.get_state()) == _digest(shard_end)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 129])
def test_ragged_batch_sizes(hip_lib, oracle, n):
    start = pa.make_boards(n, seed=n)
    rng = np.random.default_rng(n)
    ref = start.copy()
    with BatchEnvironment(n, mode=MODE_RAW) as env:
        env.make_game(start)
        for _ in range(30):
            mv = rng.integers(0, 6, size=(n, 4), dtype=np.int32)
            env.step(mv)
            oracle.step_batch(ref, mv)
        assert _digest(env.get_state()) == _digest(ref)
        assert env.counters()[CNT_STEPS] == 30 * n


def test_partial_upload_download_ranges(hip_lib):
    n = 200
    a, b = pa.make_boards(n, seed=1), pa.make_boards(n, seed=2)
    with BatchEnvironment(n, mode=MODE_RAW) as env:
        env.make_game(a)
        env.make_game(b[50:120], first=50)
        got = env.get_state()
        assert _digest(got[:50]) == _digest(a[:50]) and _digest(got[50:120]) == _digest(b[50:120])
        assert _digest(got[120:]) == _digest(a[120:])
        assert _digest(env.get_state(60, 7)) == _digest(b[60:67])
        with pytest.raises(PomError) as e:
            env.get_state(150, 51)
        assert e.value.code == 1
        with pytest.raises(PomError):
            env.make_game(a, first=1)


def test_empty_batch_and_bad_arguments(hip_lib):
    with pytest.raises(PomError) as e:
        BatchEnvironment(0)
    assert e.value.code == 1
    with pytest.raises(PomError):
        BatchEnvironment(64, device=99)
    with BatchEnvironment(8) as env:
        with pytest.raises(ValueError):
            env.step(np.zeros((8, 3), dtype=np.int32))
        with pytest.raises(PomError):
            env.step_random(1, 7, 1, 1)
        assert env.get_state(0, 0).size == 0  # empty range is fine


def test_unrepresentable_state_is_rejected_not_altered(hip_lib):
    s = pa.make_boards(130, seed=8)
    s["board"][77, 4, 4] = 12345678  # no reachable game state holds this
    with BatchEnvironment(130, mode=MODE_ENV) as env:
        with pytest.raises(PomError) as e:
            env.make_game(s)
        assert e.value.code == 3 and "env 77" in str(e.value)
        st = env.status()
        assert st["done"][77] == 1 and st["done"].sum() == 1  # parked as a finished blank board
        got = env.get_state()
        assert _digest(got[:77]) == _digest(s[:77]) and _digest(got[78:]) == _digest(s[78:])


def test_out_of_range_move_values_and_dead_agent_moves(hip_lib, oracle):
    """Move[4] entries of dead agents and values outside 0..5 are inputs: same result as the oracle."""
    n = 256
    start = pa.make_boards(n, seed=12)
    for e in range(0, n, 3):
        pa.kill(start[e], 1)
    rng = np.random.default_rng(0)
    ref = start.copy()
    with BatchEnvironment(n, mode=MODE_RAW) as env:
        env.make_game(start)
        for _ in range(40):
            mv = rng.integers(-2, 9, size=(n, 4), dtype=np.int32)
            env.step(mv)
            oracle.step_batch(ref, mv)
        assert _digest(env.get_state()) == _digest(ref)


def test_lost_agent_is_flagged_and_counted(hip_lib):
    s = pa.new_states(64)
    for e in range(64):
        pa.put_agent(s[e], 0, 0, 0)
        pa.put_agent(s[e], 2, 0, 1)
        pa.put_agent(s[e], 1, 0, 2)
        pa.put_agent(s[e], 1, 1, 3)
    mv = np.tile(np.array([Move.RIGHT, Move.LEFT, Move.DOWN, Move.DOWN], dtype=np.int32), (64, 1))
    with BatchEnvironment(64, mode=MODE_RAW) as env:
        env.make_game(s)
        env.step(mv)
        st = env.status()
        assert np.all(st["ubflags"] & UB_LOST_AGENT)
        assert env.counters()[CNT_UB_TICKS] == 64
        got = env.get_state()
        assert np.all(got["agents"]["x"][:, 2] == 1) and np.all(got["agents"]["y"][:, 2] == 1)


def test_single_state_drop_in_and_status_api(hip_lib, oracle):
    from pomcpp_amd.batch import step_one
    s = pa.make_boards(1, seed=4)
    ref = s.copy()
    rng = np.random.default_rng(4)
    for _ in range(25):
        mv = rng.integers(0, 6, size=4, dtype=np.int32)
        step_one(s, mv)
        oracle.step(ref, mv)
    assert _digest(s) == _digest(ref)
    with BatchEnvironment(3, mode=MODE_ENV, max_steps=5) as env:
        env.make_game(pa.make_boards(3, seed=1))
        for _ in range(7):
            env.step(np.zeros((3, 4), dtype=np.int32))
        st = env.status()
        assert st["time_step"].tolist() == [5, 5, 5] and st["done"].tolist() == [1, 1, 1]  # frozen at the cap
        assert st["winner"].tolist() == [-1, -1, -1] and env.is_done().all() and not env.is_draw().any()
        env.snapshot()  # current state becomes the restart point, status cleared there


def test_one_state_path_random_play_rejection_and_env_bookkeeping(hip_lib, oracle):
    """pom_step / pom_env_step (one pinned State, one launch): stress play with arbitrary Move values against the oracle every
    tick, UB flags included; a State the record cannot hold is refused and left alone; max_steps ends a game."""
    from pomcpp_amd.batch import step_one, env_step_one
    rng = np.random.default_rng(11)
    for kind, seed in (("stress", 3), ("ffa", 5), ("stress", 8)):
        s = pa.make_boards(1, seed=seed, kind=kind)
        ref = s.copy()
        status = dict(done=0, winner=-1, draw=0)
        for t in range(150):
            mv = rng.integers(-1, 8, size=4, dtype=np.int32) if t % 7 == 0 else rng.integers(0, 6, size=4, dtype=np.int32)
            if status["done"]:
                break
            r = env_step_one(s, mv)
            ub = oracle.env_step(ref, mv, status)
            assert _digest(s) == _digest(ref), (kind, t)
            assert (r["done"], r["draw"], r["ubflags"]) == (status["done"], status["draw"], ub), (kind, t)
            assert r["winner"] == (status["winner"] if status["done"] and not status["draw"] and r["winner"] >= 0 else -1)
    # RAW: the bare Step never touches timeStep
    s = pa.make_boards(1, seed=9, kind="stress")
    ref = s.copy()
    for t in range(60):
        mv = rng.integers(0, 6, size=4, dtype=np.int32)
        step_one(s, mv)
        oracle.step(ref, mv)
        assert _digest(s) == _digest(ref), t
    assert int(s["timeStep"][0]) == 0
    # refused, not altered — every kind of field the record cannot hold
    def spoil(k):
        bad = pa.make_boards(1, seed=2)
        if k == 0:
            bad["board"][0, 4, 4] = 12345678
        elif k == 1:
            bad["aliveAgents"][0] = 1000
        elif k == 2:
            bad["bombs_count"][0] = 21
        elif k == 3:
            bad["flames_queue"]["strength"][0, 17] = 300  # a stale slot: checked as well
        elif k == 4:
            bad["agents"]["x"][0, 2] = 11
        elif k == 5:  # a live bomb owned by agent 7 would index agents[7]
            bad["bombs_queue"][0, 0] = 3 | (3 << 4) | (7 << 8) | (2 << 12) | (9 << 16)
            bad["bombs_count"][0] = 1
        return bad
    for k in range(6):
        bad = spoil(k)
        before = bad.copy()
        with pytest.raises(PomError) as e:
            step_one(bad, np.zeros(4, dtype=np.int32))
        assert e.value.code == 3 and bad.tobytes() == before.tobytes(), k
        with BatchEnvironment(1, mode=MODE_RAW) as env:  # the batch path refuses the same States
            with pytest.raises(PomError):
                env.make_game(before)
    # the step cap
    s = pa.make_boards(1, seed=1)
    for t in range(5):
        r = env_step_one(s, np.zeros(4, dtype=np.int32), max_steps=5)
        assert r["done"] == (1 if t == 4 else 0) and r["winner"] == -1 and r["draw"] == 0
    assert int(s["timeStep"][0]) == 5


@pytest.mark.parametrize("epw,lpe", [(16, 4), (16, 1), (32, 1), (64, 1)])
@pytest.mark.parametrize("kind,dist", [("ffa", DIST_RANDOM), ("stress", DIST_STRESS)])
def test_every_kernel_variant_matches_oracle(hip_lib, oracle, epw, lpe, kind, dist):
    """The four kernel instantiations (quad per env; one lane per env at 16 / 32 / 64 envs per wavefront) are the same
    function of the input."""
    n, ticks, seed = 3000 + epw + lpe, 80, 4242
    start = pa.make_boards(n, seed=31, kind=kind)
    want = start.copy()
    oracle.run_random(want, start, ticks, seed, 0, 0, dist, 800)
    for tpl in (1, 5):
        with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, envs_per_wave=epw, lanes_per_env=lpe) as env:
            assert env.launch_shape()[:2] == (epw, lpe)
            env.make_game(start)
            env.step_random(seed, dist, ticks=ticks, ticks_per_launch=tpl)
            assert _digest(env.get_state()) == _digest(want)
            assert env.counters()[CNT_STEPS] == n * ticks
    rng = np.random.default_rng(epw)
    ref = start.copy()
    with BatchEnvironment(n, mode=MODE_RAW, envs_per_wave=epw, lanes_per_env=lpe) as env:
        env.make_game(start)
        for _ in range(25):
            mv = rng.integers(0, 6, size=(n, 4), dtype=np.int32)
            env.step(mv)
            oracle.step_batch(ref, mv)
        assert _digest(env.get_state()) == _digest(ref)
    with pytest.raises(PomError):
        BatchEnvironment(64, envs_per_wave=48)
    with pytest.raises(PomError):
        BatchEnvironment(64, envs_per_wave=32, lanes_per_env=4)


@pytest.mark.parametrize("streams", [1, 2, 3, 8])
def test_split_steps_over_streams_give_identical_results(hip_lib, oracle, streams):
    """A step may be issued as several launches on internal streams; states, status and counters must not depend on it,
    with the synthetic stream, with host moves, and when uploads / downloads are interleaved with steps."""
    n, ticks, seed = 5000, 60, 808
    start = pa.make_boards(n, seed=5)
    want = start.copy()
    oracle.run_random(want, start, ticks, seed, 0, 0, DIST_RANDOM, 800)
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=streams) as env:
        # one stream: plain launches in a row; more: chained launches (one launch over all tiles per tick) rotating over them
        assert env.issue_info() == (("chain", streams) if streams > 1 else ("threads", 1)) and env.launch_shape()[2] == 1
        env.make_game(start)
        env.step_random(seed, DIST_RANDOM, ticks=ticks // 2)
        mid = env.get_state()          # joins the sub-streams
        env.make_game(mid)             # re-upload: snapshot becomes `mid`, but nothing restarts in the next 30 ticks unless done
        env.make_game(start)           # back to the original snapshot ...
        env.set_tick(0)
        env.step_random(seed, DIST_RANDOM, ticks=ticks // 3)   # ... and replay the whole run,
        env.set_streams(1 + streams % 3)                       # changing the split on the way
        env.step_random(seed, DIST_RANDOM, ticks=ticks - ticks // 3)
        assert _digest(env.get_state()) == _digest(want)
        assert env.counters()[CNT_STEPS] == n * (ticks + ticks // 2)
    rng = np.random.default_rng(streams)
    ref = start.copy()
    with BatchEnvironment(n, mode=MODE_RAW, streams=streams) as env:
        env.make_game(start)
        for t in range(20):
            mv = rng.integers(0, 6, size=(n, 4), dtype=np.int32)
            env.step(mv)
            mv[:] = 0                  # the host buffer may be reused immediately
        # recompute the same moves for the oracle
        rng = np.random.default_rng(streams)
        for t in range(20):
            oracle.step_batch(ref, rng.integers(0, 6, size=(n, 4), dtype=np.int32))
        assert _digest(env.get_state()) == _digest(ref)
    with pytest.raises(PomError):
        BatchEnvironment(64, streams=9)


@pytest.mark.parametrize("mode", ["direct", "threads", "graph", "chain"])
@pytest.mark.parametrize("streams", [1, 3])
def test_issue_modes_give_identical_results(hip_lib, oracle, mode, streams):
    """How the launches of a several-tick call are issued (PomBatchOptions.issue_mode: by the calling thread, by one helper thread
    per sub-stream, as replayed HIP graphs of 20 ticks) must not show in states, counters or the tick that keys the move stream —
    random play, ticks_per_launch 1 and 3, the fused SimpleAgent kernel, calls shorter and longer than a graph chunk."""
    from pomcpp_amd.batch import ISSUE_CHAIN, ISSUE_DIRECT, ISSUE_GRAPH, ISSUE_THREADS
    im = {"direct": ISSUE_DIRECT, "threads": ISSUE_THREADS, "graph": ISSUE_GRAPH, "chain": ISSUE_CHAIN}[mode]
    n, seed = 4000, 99
    start = pa.make_boards(n, seed=12)
    want = start.copy()
    oracle.run_random(want, start, 47 + 5 + 63, seed, 0, 0, DIST_RANDOM, 800)
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=streams, issue_mode=im) as env:
        env.make_game(start)
        env.step_random(seed, DIST_RANDOM, ticks=47)                      # two chunks + 7
        env.step_random(seed, DIST_RANDOM, ticks=5)                       # shorter than a chunk
        env.step_random(seed, DIST_RANDOM, ticks=63, ticks_per_launch=3)  # 21 launches of 3 ticks
        assert _digest(env.get_state()) == _digest(want)
        assert env.counters()[CNT_STEPS] == n * (47 + 5 + 63)
    ref, mems = start.copy(), np.zeros((n, 4, 16), dtype=np.int32)
    oracle.run_simple(ref, start, mems, 45, seed, 0, 0, 800)
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=streams, issue_mode=im) as env:
        env.make_game(start)
        env.step_simple(seed, 45)
        assert _digest(env.get_state()) == _digest(ref)
        assert np.array_equal(env.policy_memory(), mems)
    with pytest.raises(PomError):
        BatchEnvironment(64, issue_mode=7)


def test_one_state_path_from_several_threads(hip_lib, oracle, tmp_path):
    """pom_step is re-entrant over distinct States like the reference's Step (performance_test.cpp:71-94 steps one env per
    std::thread): every calling thread has a pinned page of its own and launches are combined (one kernel serves every request
    pending at that moment).  Eight native threads (tests/cpp/step_threads.cpp), a State each: every final State equals the
    oracle's, and the threads overlap (together well above one thread's rate)."""
    import json
    import subprocess
    import __graft_entry__ as g
    exe = g.build_step_threads()
    n_threads, ticks = 8, 1500
    states = pa.make_boards(n_threads, seed=40, kind="stress")
    moves = np.random.default_rng(3).integers(0, 6, size=(n_threads, ticks, 4), dtype=np.int32)
    (tmp_path / "s.bin").write_bytes(states.tobytes())
    (tmp_path / "m.bin").write_bytes(moves.tobytes())
    out = subprocess.run([exe, str(n_threads), str(ticks), str(tmp_path / "s.bin"), str(tmp_path / "m.bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])

    def fnv(b):
        h = 1469598103934665603
        for x in b:
            h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % h
    for k in range(n_threads):
        ref = np.ascontiguousarray(states[k:k + 1]).copy()
        for t in range(ticks):
            oracle.step(ref, moves[k, t])
        ref["agents"]["pad"] = 0
        assert r["digests"][k] == fnv(ref.tobytes()), k
    assert r["calls_per_s_all_threads"] > 3.0 * r["calls_per_s_1_thread"], r
    # the Python wrapper from Python threads (the GIL is released inside the call)
    import threading
    from pomcpp_amd.batch import step_one
    sts = [np.ascontiguousarray(states[k:k + 1]).copy() for k in range(4)]
    refs = [s.copy() for s in sts]
    errors = []

    def play(k):
        try:
            for t in range(60):
                step_one(sts[k], moves[k, t])
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)
    th = [threading.Thread(target=play, args=(k,)) for k in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    for k in range(4):
        for t in range(60):
            oracle.step(refs[k], moves[k, t])
        assert _digest(sts[k]) == _digest(refs[k]), k
