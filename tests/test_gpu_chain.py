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


def test_full_size_chained_against_plain(hip_lib):
    """65,536 envs (the headline's size: every wavefront slot of the chip taken by one launch, so launches really queue behind each
    other): long and short chained calls, stress moves, the fused SimpleAgent kernel, against a twin that never chains."""
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
        assert np.array_equal(a.counters(), b.counters()) and np.array_equal(a.policy_memory(), b.policy_memory())
        sa, sb = a.status(), b.status()
        assert all(np.array_equal(sa[k], sb[k]) for k in sa)
