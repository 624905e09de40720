"""GPU: chained launches (POM_ISSUE_CHAIN, pomcpp_amd/csrc/pom_chain.h) — every launch covers all tiles, consecutive launches go
to different streams, a ticket word per tile orders the tile's ticks.  Results must equal the oracle's whatever overlaps: calls
back to back without a join, launches of two calls in flight together, other handles alive, stream counts, ragged sizes, the
fused policy / fresh boards / end-of-tick reset twins, the periodic reset of the ticket words."""
import os
import subprocess
import sys

import numpy as np
import pytest

import pomcpp_amd as pa
from pomcpp_amd.batch import (BatchEnvironment, MODE_ENV, DIST_RANDOM, DIST_STRESS, CNT_STEPS, ISSUE_CHAIN, ISSUE_THREADS, ISSUE_GRAPH,
                              RESET_AT_END)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(got, ref):
    g, r = got.copy(), ref.copy()
    g["agents"]["pad"] = 0
    r["agents"]["pad"] = 0
    return g.tobytes() == r.tobytes()


def test_default_issue_mode_is_chained_where_it_applies(hip_lib):
    with BatchEnvironment(65536, auto_reset=True, max_steps=800) as env:
        assert env.issue_info() == ("chain", 2) and env.launch_shape() == (16, 4, 1)
        env.make_game(pa.make_boards(65536, seed=1))
        env.step_random(1, DIST_RANDOM, ticks=60)   # a long call rotates over three streams, a short one over two
        assert env.issue_info() == ("chain", 3)
        env.step_random(1, DIST_RANDOM, ticks=12)
        assert env.issue_info() == ("chain", 2)
        assert env.counters()[CNT_STEPS] == 65536 * 72
    with BatchEnvironment(262144) as env:  # from 196,608 envs up: sub-batches on parallel streams
        assert env.issue_info() == ("threads", 3) and env.launch_shape() == (16, 4, 3)
    with BatchEnvironment(4096, streams=1) as env:  # one stream: launches in a row, nothing to chain
        assert env.issue_info()[0] == "threads" and env.launch_shape()[2] == 1
    with BatchEnvironment(4096, lanes_per_env=1) as env:  # the one-lane-per-env shapes have no chained twin
        assert env.issue_info()[0] == "threads"
    with BatchEnvironment(4096, issue_mode=ISSUE_THREADS) as env:
        assert env.issue_info()[0] == "threads"


@pytest.mark.parametrize("n,kind,dist", [(4000, "ffa", DIST_RANDOM), (1000, "stress", DIST_STRESS), (37, "ffa", DIST_RANDOM)])
def test_calls_back_to_back_overlap_and_match_the_oracle(hip_lib, oracle, n, kind, dist):
    """Many fresh handles, 2 - 4 streams, several calls without a join in between (launches of consecutive calls are in flight
    together), a several-ticks-per-launch call in the middle (sub-batches: the streams are joined on the way in and out), other
    handles with their own streams, helper threads and graphs alive beside the one under test."""
    start = pa.make_boards(n, seed=12, kind=kind)
    plans = [((47, 1), (5, 1), (63, 3), (9, 1)), ((47, 1), (5, 1), (1, 1), (30, 1)), ((30, 1),), ((2, 1), (3, 1), (4, 1), (6, 3), (1, 1))]
    want = []
    for plan in plans:
        ref, done = start.copy(), 0
        for ticks, _ in plan:
            oracle.run_random(ref, start, ticks, 99, 0, done, dist, 800)
            done += ticks
        want.append((ref, done))
    others = []
    try:
        for it in range(48):
            streams, (plan, (ref, done)) = (2, 3, 3, 4)[it % 4], list(zip(plans, want))[it % len(plans)]
            if it % 6 == 0:
                o = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=3, issue_mode=(ISSUE_THREADS, ISSUE_GRAPH)[(it // 6) % 2])
                o.make_game(start)
                o.step_random(1, dist, ticks=25)
                others.append(o)
                if len(others) > 3:
                    others.pop(0).close()
            with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=streams, issue_mode=ISSUE_CHAIN) as env:
                assert env.issue_info() == ("chain", streams)
                env.make_game(start)
                for ticks, tpl in plan:
                    env.step_random(99, dist, ticks=ticks, ticks_per_launch=tpl)
                assert _same(env.get_state(), ref), (it, streams, plan)
                assert env.counters()[CNT_STEPS] == n * done
    finally:
        for o in others:
            o.close()


def test_chained_twins_policy_fresh_boards_reset_at_end(hip_lib, oracle):
    n, seed, bseed = 3000, 31, 77
    start = pa.make_boards(n, seed=3)
    # fused SimpleAgent x4, calls back to back
    ref, mems = start.copy(), np.zeros((n, 4, 16), dtype=np.int32)
    oracle.run_simple(ref, start, mems, 21 + 4 + 30, seed, 0, 0, 800)
    for streams in (2, 3):
        with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=streams, issue_mode=ISSUE_CHAIN) as env:
            env.make_game(start)
            env.step_simple(seed, 21)
            env.step_simple(seed, 4)
            env.step_simple(seed, 30)
            assert _same(env.get_state(), ref) and np.array_equal(env.policy_memory(), mems)
    # fresh device-generated boards, random moves and SimpleAgents
    for streams in (2, 4):
        with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, fresh_boards=True, board_seed=bseed, streams=streams,
                              issue_mode=ISSUE_CHAIN) as env:
            env.generate(bseed)
            st = oracle.boardgen(bseed, np.arange(n), np.zeros(n))
            eps = np.zeros(n, dtype=np.int32)
            for ticks in (60, 7, 90):
                env.step_random(seed, DIST_RANDOM, ticks=ticks)
            oracle.run_random_fresh(st, eps, 157, seed, bseed, 0, 0, DIST_RANDOM, 800)
            assert _same(env.get_state(), st) and np.array_equal(env.episodes(), eps)
    # reset at the END of the finishing tick: same games as the reset at the start of the next one, one restart later
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=800, streams=3, issue_mode=ISSUE_CHAIN) as env, \
         BatchEnvironment(n, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=800, streams=3, issue_mode=ISSUE_THREADS) as plain:
        for e in (env, plain):
            e.make_game(start)
            for ticks in (33, 2, 48):
                e.step_random(seed, DIST_RANDOM, ticks=ticks)
        assert _same(env.get_state(), plain.get_state())
        a, b = env.last_results(), plain.last_results()
        assert all(np.array_equal(a[k], b[k]) for k in a)


def test_changing_streams_and_mixing_step_kinds(hip_lib):
    """Chained calls, explicit-move steps (one plain launch, ordered with the caller's stream) and changes of the stream count
    interleaved, against a twin handle that never chains (one stream, plain launches: compared with the oracle elsewhere)."""
    n, seed = 2500, 5
    start = pa.make_boards(n, seed=9)
    rng = np.random.default_rng(1)
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800) as env, \
         BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, issue_mode=ISSUE_THREADS, streams=1) as twin:
        assert env.issue_info() == ("chain", 2) and twin.issue_info()[0] == "threads"
        for e in (env, twin):
            e.make_game(start)
        for k, ticks in enumerate((13, 1, 8, 21, 2, 40)):
            mv = rng.integers(0, 6, size=(n, 4), dtype=np.int32)
            for e in (env, twin):
                e.step_random(seed, DIST_RANDOM, ticks=ticks)
                e.step(mv)
                e.step_random(seed, DIST_RANDOM, ticks=3)
            env.set_streams(2 + k % 3)
            if k % 2:
                assert _same(env.get_state(), twin.get_state()), k
        assert _same(env.get_state(), twin.get_state())
        assert np.array_equal(env.counters(), twin.counters())


def test_ticket_words_start_over(hip_lib, oracle):
    """After 2^27 visits the tile words are zeroed (their fields must not run into each other).  POM_CHAIN_RESET_AT makes that happen
    every 50 visits in a child process."""
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, ISSUE_CHAIN
from tests.oracle_lib import Oracle
ora = Oracle()
n = 2000
start = pa.make_boards(n, seed=4)
ref = start.copy()
env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=3, issue_mode=ISSUE_CHAIN)
env.make_game(start)
done = 0
for ticks in (20, 20, 20, 7, 45, 49, 3, 30, 30):
    env.step_random(8, 1, ticks=ticks)
    ora.run_random(ref, start, ticks, 8, 0, done, 1, 800)
    done += ticks
got = env.get_state()
got["agents"]["pad"] = 0
assert got.tobytes() == ref.tobytes()
assert env.counters()[0] == n * done
print("ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, POM_CHAIN_RESET_AT="50"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("auto_reset,fresh", [(True, False), (RESET_AT_END, False), (True, True)])
def test_random_api_sequences_chained_against_plain(hip_lib, auto_reset, fresh):
    """Differential: the same seeded random sequence of calls on a handle with chained launches and on a twin that never chains
    (one stream, plain launches; compared with the oracle by the other suites).  Steps of every kind, several-tick-per-launch
    calls, stream changes, snapshots, partial uploads, tick changes and reads in between — states, status, counters, agent memory
    and episode counts must never differ."""
    n = 1777
    start = pa.make_boards(n, seed=21)
    for seq in range(8):
        rng = np.random.default_rng(1000 * seq + (7 if fresh else 0) + int(auto_reset))
        kw = dict(mode=MODE_ENV, auto_reset=auto_reset, max_steps=300, fresh_boards=fresh, board_seed=5)
        with BatchEnvironment(n, issue_mode=ISSUE_CHAIN, streams=int(rng.integers(2, 5)), **kw) as a, \
             BatchEnvironment(n, issue_mode=ISSUE_THREADS, streams=1, **kw) as b:
            for e in (a, b):
                if fresh:
                    e.generate(5)
                else:
                    e.make_game(start)
            for op in range(14):
                kind = int(rng.integers(0, 10))
                seed, ticks = int(rng.integers(1, 1 << 30)), int(rng.integers(1, 50))
                if kind <= 2:
                    for e in (a, b):
                        e.step_random(seed, DIST_RANDOM, ticks=ticks)
                elif kind == 3:
                    tpl = int(rng.integers(2, 5))
                    for e in (a, b):
                        e.step_random(seed, DIST_STRESS, ticks=tpl * (1 + ticks // 8), ticks_per_launch=tpl)
                elif kind == 4:
                    for e in (a, b):
                        e.step_simple(seed, 1 + ticks // 4)
                elif kind == 5:
                    mv = rng.integers(0, 6, size=(n, 4), dtype=np.int32)
                    for e in (a, b):
                        e.step(mv)
                elif kind == 6:
                    a.set_streams(int(rng.integers(1, 6)))
                    tick = int(rng.integers(0, 1000))
                    for e in (a, b):
                        e.set_tick(tick)
                elif kind == 7 and not fresh:
                    for e in (a, b):
                        e.snapshot()
                elif kind == 8 and not fresh:
                    first, count = int(rng.integers(0, n - 200)), int(rng.integers(1, 200))
                    for e in (a, b):
                        e.make_game(np.ascontiguousarray(start[first:first + count]), first=first)
                elif kind == 9 and op % 2:
                    for e in (a, b):  # the two-kernel form of a SimpleAgent tick: policy into the move buffer, then the step
                        e.policy_simple(seed)
                        e.step_policy()
                else:
                    sa, sb = a.status(), b.status()
                    assert all(np.array_equal(sa[k], sb[k]) for k in sa), (seq, op)
                if op % 4 == 3:
                    assert _same(a.get_state(), b.get_state()), (seq, op, kind)
            assert _same(a.get_state(), b.get_state()), seq
            assert np.array_equal(a.counters(), b.counters()), seq
            assert np.array_equal(a.policy_memory(), b.policy_memory()), seq
            assert np.array_equal(a.episodes(), b.episodes()), seq
            if auto_reset == RESET_AT_END:
                ra, rb = a.last_results(), b.last_results()
                assert all(np.array_equal(ra[k], rb[k]) for k in ra), seq


def test_full_size_chained_against_plain(hip_lib, oracle):
    """65,536 envs (the headline's size: every wavefront slot of the chip taken by one launch, so launches really queue behind each
    other): long and short chained calls, stress moves, the fused SimpleAgent kernel, against a twin that never chains — and two
    slices of 2,048 envs against the oracle playing the same sequence of calls."""
    n = 65536
    start = pa.make_boards(n, seed=33, kind="stress")
    kw = dict(mode=MODE_ENV, auto_reset=True, max_steps=800)
    with BatchEnvironment(n, **kw) as a, BatchEnvironment(n, issue_mode=ISSUE_THREADS, streams=1, **kw) as b:
        assert a.issue_info()[0] == "chain" and b.issue_info()[0] == "threads"
        for e in (a, b):
            e.make_game(start)
            e.step_random(5, DIST_STRESS, ticks=130)   # three streams
            e.step_random(5, DIST_STRESS, ticks=9)     # two streams, in flight together with the call before
            e.step_random(6, DIST_RANDOM, ticks=45)    # another seed: waits for the launches before it
            e.step_simple(6, 25)
            e.step_random(6, DIST_RANDOM, ticks=3)
        assert _same(a.get_state(), b.get_state())
        for lo in (0, 40000):
            cnt = 2048
            ini = np.ascontiguousarray(start[lo:lo + cnt])
            ref, mems = ini.copy(), np.zeros((cnt, 4, 16), dtype=np.int32)
            oracle.run_random(ref, ini, 139, 5, lo, 0, DIST_STRESS, 800)
            oracle.run_random(ref, ini, 45, 6, lo, 139, DIST_RANDOM, 800)
            oracle.run_simple(ref, ini, mems, 25, 6, lo, 184, 800)
            oracle.run_random(ref, ini, 3, 6, lo, 209, DIST_RANDOM, 800)
            assert _same(a.get_state(lo, cnt), ref), lo
            assert np.array_equal(a.policy_memory(lo, cnt), mems), lo
        assert np.array_equal(a.counters(), b.counters()) and np.array_equal(a.policy_memory(), b.policy_memory())
        sa, sb = a.status(), b.status()
        assert all(np.array_equal(sa[k], sb[k]) for k in sa)


def _oracle_explicit(oracle, ref, start, status, moves, cap, at_end, last=None, term=None):
    """one tick of Environment::Step with explicit moves on every env of `ref`, with the batch's auto-reset (tests/test_reset_modes.py)"""
    n = ref.size
    fin = np.zeros(n, int)
    for i in range(n):
        if not at_end and (status[i]["done"] or ref["timeStep"][i] >= cap):  # reset at the start of the next tick
            ref[i] = start[i]
            status[i] = dict(done=0, winner=-1, draw=0)
        oracle.env_step(ref[i:i + 1], moves[i], status[i])
        if at_end and (status[i]["done"] or ref["timeStep"][i] >= cap):
            fin[i] = 1
            if last is not None:
                last["winner"][i], last["draw"][i] = status[i]["winner"], status[i]["draw"]
                last["length"][i], last["alive"][i] = ref["timeStep"][i], ref["aliveAgents"][i]
                term[i] = ref[i]
            ref[i] = start[i]
            status[i] = dict(done=0, winner=-1, draw=0)
    return fin


@pytest.mark.parametrize("at_end", [True, False])
def test_move_tape_chained_against_the_oracle(hip_lib, oracle, at_end):
    """pom_batch_step_device_many: K ticks of explicit Move[4] from a tape in device memory, issued as chained launches — every tick
    played by the oracle with the same moves; states, restart marks, outcomes and terminal states after every call (calls of 1 .. 55
    ticks, then forty calls of two: nearly every tick boundary is looked at)."""
    import torch
    n, cap = 400, 60
    start = pa.make_boards(n, seed=41)
    rng = np.random.default_rng(11)
    ref = start.copy()
    status = [dict(done=0, winner=-1, draw=0) for _ in range(n)]
    last = dict(winner=np.full(n, -1), draw=np.zeros(n, int), length=np.zeros(n, int), alive=np.zeros(n, int))
    term = np.zeros(n, dtype=start.dtype)
    plan = [2, 3, 5, 8, 1, 13, 21, 34, 55] + [2] * 40
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=RESET_AT_END if at_end else True, max_steps=cap, issue_mode=ISSUE_CHAIN) as env:
        assert env.issue_info()[0] == "chain"
        env.make_game(start)
        played = 0
        for k in plan:
            tape = rng.integers(0, 6, size=(k, n, 4), dtype=np.int32)
            dev = torch.from_numpy(tape).to("cuda")
            env.step_device_many(dev)
            fin = None
            for t in range(k):
                fin = _oracle_explicit(oracle, ref, start, status, tape[t], cap, at_end, last, term)
            played += k
            got = env.get_state()  # synchronises: the tape may go now
            if at_end:
                assert _same(got, ref), (played, k)
                r = env.last_results()
                assert r["finished"].tolist() == fin.tolist(), (played, k)
                for key in ("winner", "draw", "length", "alive"):
                    assert r[key].tolist() == last[key].tolist(), (played, key)
            else:
                assert _same(got, ref), (played, k)
                st = env.status()
                assert st["done"].tolist() == [int(s["done"] or ref["timeStep"][i] >= cap) for i, s in enumerate(status)], played
        if at_end:
            assert _same(env.get_terminal_state(), term)
        assert env.counters()[CNT_STEPS] == n * played
        stats = env.chain_stats()
        assert stats["launches"] == sum(k for k in plan if k >= 2) and stats["tiles_recovered"] == 0, stats
        with pytest.raises(ValueError):
            env.step_device_many(torch.zeros((3, n, 3), dtype=torch.int32, device="cuda"))


def test_move_tape_full_size_equals_one_launch_per_tick(hip_lib, oracle):
    """65,536 envs: a 60-tick tape through chained launches against the same moves handed over tick by tick (pom_batch_step_device,
    one plain launch per tick), and an oracle slice of 4,096 envs"""
    import torch
    n, k = 65536, 60
    start = pa.make_boards(n, seed=5)
    gen = torch.Generator(device="cuda").manual_seed(3)
    tape = torch.randint(0, 6, (k, n, 4), dtype=torch.int32, device="cuda", generator=gen)
    kw = dict(mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=800)
    with BatchEnvironment(n, **kw) as a, BatchEnvironment(n, issue_mode=ISSUE_THREADS, streams=1, **kw) as b:
        for e in (a, b):
            e.make_game(start)
        a.step_random(7, DIST_RANDOM, ticks=25)  # chained launches of another kind in flight when the tape call arrives
        b.step_random(7, DIST_RANDOM, ticks=25)
        a.step_device_many(tape)
        for t in range(k):
            b.step_device(tape[t])
        ga, gb = a.get_state(), b.get_state()
        assert _same(ga, gb)
        assert np.array_equal(a.counters(), b.counters())
        ra, rb = a.last_results(), b.last_results()
        assert all(np.array_equal(ra[x], rb[x]) for x in ra)
    # the oracle on a slice: random ticks first, then the tape (reset at the end of the finishing tick)
    m = 4096
    ref = np.ascontiguousarray(start[:m]).copy()
    with BatchEnvironment(m, **kw) as c:
        c.make_game(ref)
        c.step_random(7, DIST_RANDOM, ticks=25)
        mid = c.get_state()
    ref = mid.copy()
    status = [dict(done=0, winner=-1, draw=0) for _ in range(m)]
    host_tape = tape[:, :m].cpu().numpy()
    for t in range(k):
        _oracle_explicit(oracle, ref, start[:m], status, host_tape[t], 800, True)
    assert _same(ga[:m], ref)


@pytest.mark.parametrize("wait_us", ["0", "6"])
def test_forced_give_ups_are_replayed(hip_lib, wait_us):
    """POM_CHAIN_WAIT_US=0 (and 6: only the slower predecessors are given up on, so chained and replayed ticks mix): a wavefront that
    finds its tile not ready gives up at once and poisons the tile; nobody steps a poisoned
    tile, and the next call that reads the batch replays the missing ticks from the log of chained calls.  Random moves, the fused
    SimpleAgent kernel, a move tape and the end-of-tick reset; results must equal the oracle's and tiles must really have been left
    behind (otherwise the test tests nothing)."""
    code = r'''
import numpy as np, sys, torch
sys.path.insert(0, %r)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, ISSUE_CHAIN, RESET_AT_END, CNT_STEPS
from tests.oracle_lib import Oracle
from tests.test_gpu_chain import _oracle_explicit, _same
ora = Oracle()
n = 16384
start = pa.make_boards(n, seed=4)
recovered = 0
# random moves + SimpleAgent, calls back to back, reads in between
ref, mems = start.copy(), np.zeros((n, 4, 16), dtype=np.int32)
env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=3, issue_mode=ISSUE_CHAIN)
env.make_game(start)
done = 0
for ticks in (30, 45, 7):
    env.step_random(8, 1, ticks=ticks)
    ora.run_random(ref, start, ticks, 8, 0, done, 1, 800)
    done += ticks
assert _same(env.get_state(), ref)
# the ordinary RL loop step_random(K) -> observe(): the planes are those of the state a download returns, for every element type — a
# tile left behind by the chained launches is caught up before it is looked at (pom_batch_observe settles; advisor, round 4)
env.step_random(8, 1, ticks=9)
ora.run_random(ref, start, 9, 8, 0, done, 1, 800)
done += 9
first = {dt: env.observe(dtype=dt, attrs=False)[0].cpu().numpy() for dt in ("uint8",)}
assert _same(env.get_state(), ref)
from tests.test_observe import _oracle as _observe_oracle
assert np.array_equal(first["uint8"], _observe_oracle().observe(env.get_state())[0])
for dt in ("uint8", "float16", "float32", "codes"):
    again = env.observe(dtype=dt, attrs=False)[0].cpu().numpy()
    if dt in first:
        assert np.array_equal(first[dt], again), dt
    if dt != "codes":
        assert np.array_equal(again.astype(np.uint8), first["uint8"]), dt
env.step_random(8, 1, ticks=12)
env.step_simple(8, 20)
env.step_random(9, 1, ticks=6)
ora.run_random(ref, start, 12, 8, 0, done, 1, 800)
ora.run_simple(ref, start, mems, 20, 8, 0, done + 12, 800)
ora.run_random(ref, start, 6, 9, 0, done + 32, 1, 800)
done += 38
assert _same(env.get_state(), ref) and np.array_equal(env.policy_memory(), mems)
assert env.counters()[CNT_STEPS] == n * done
st = env.chain_stats()
recovered += st["tiles_recovered"]
assert st["ticks_replayed"] >= st["tiles_recovered"]
env.close()
# a move tape with the end-of-tick reset
m, cap = 2048, 50
s2 = np.ascontiguousarray(start[:m])
ref = s2.copy()
status = [dict(done=0, winner=-1, draw=0) for _ in range(m)]
rng = np.random.default_rng(2)
env = BatchEnvironment(m, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=cap, streams=3, issue_mode=ISSUE_CHAIN)
env.make_game(s2)
for k in (40, 25):
    tape = rng.integers(0, 6, size=(k, m, 4), dtype=np.int32)
    env.step_device_many(torch.from_numpy(tape).to("cuda"))
    for t in range(k):
        _oracle_explicit(ora, ref, s2, status, tape[t], cap, True)
    assert _same(env.get_state(), ref)
recovered += env.chain_stats()["tiles_recovered"]
env.close()
assert recovered > 0, "no wavefront ever had to wait: nothing was tested"
print("ok", recovered)
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, POM_CHAIN_WAIT_US=wait_us), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


@pytest.mark.parametrize("streams", [2, 3, 4])
def test_hand_off_litmus(hip_lib, streams):
    """The hand-off by itself (pom_chain_litmus): 4,096 records, 300 launches on 2 - 4 streams = 1.2 M hand-offs, every visit checking
    all 1,792 dwords of its record against what the visit before left (tagged per tile and dword: a stale, torn or foreign record
    cannot pass) — while another handle steps 65,536 envs with chained launches of its own on the same device."""
    from pomcpp_amd.batch import chain_litmus
    with BatchEnvironment(65536, mode=MODE_ENV, auto_reset=True, max_steps=800) as other:
        other.make_game(pa.make_boards(65536, seed=2))
        other.step_random(3, DIST_RANDOM, ticks=400)  # queued: runs while the litmus does
        r = chain_litmus(4096, 300, streams)
        other.sync()
        assert other.chain_stats()["tiles_recovered"] == 0
    assert r["visits"] == r["visits_expected"] == 4096 * 300, r
    assert r["bad_records"] == 0 and r["bad_dwords"] == 0 and r["tiles_wrong"] == 0 and r["flags"] == 0, r
    small = chain_litmus(64, 2000, streams)  # few tiles, long chains: every launch waits for the one before on nearly every tile
    assert small["visits"] == 64 * 2000 and small["bad_records"] == 0 and small["tiles_wrong"] == 0 and small["flags"] == 0, small
