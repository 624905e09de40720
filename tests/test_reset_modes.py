"""PomBatchOptions.auto_reset = POM_RESET_AT_END: the tick that finishes an episode leaves the env on its next start state, so
that what a caller observes is the state its next move is applied to (with the start-of-tick reset a supplied move
was chosen looking at the finished game).  The sequence of states stepped is the same in both modes."""
import numpy as np
import pytest

import pomcpp_amd as pa
from pomcpp_amd.batch import (BatchEnvironment, MODE_ENV, DIST_RANDOM, DIST_STRESS, RESET_AT_END, RESET_AT_START, CNT_STEPS,
                              CNT_EPISODES, CNT_RESETS, PomError)

pytestmark = pytest.mark.gpu


def _same(got, want):
    want = want.copy()
    want["agents"]["pad"] = 0
    return got.tobytes() == want.tobytes()


def test_explicit_moves_every_tick_against_the_oracle(hip_lib, oracle):
    """moves for tick t are drawn after looking at the state of tick t; every tick: states, restart marks, outcomes, terminal states"""
    n, cap = 500, 60
    start = pa.make_boards(n, seed=31)
    rng = np.random.default_rng(5)
    ref = start.copy()
    status = [dict(done=0, winner=-1, draw=0) for _ in range(n)]
    last = dict(winner=np.full(n, -1), draw=np.zeros(n, int), length=np.zeros(n, int), alive=np.zeros(n, int))
    term = np.zeros(n, dtype=start.dtype)
    total_fin = 0
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=cap) as env:
        env.make_game(start)
        assert not env.last_results()["finished"].any()
        for t in range(150):
            shown = env.get_state()
            assert _same(shown, ref), f"tick {t}: the state shown is not the state the move will be applied to"
            moves = rng.integers(0, 6, size=(n, 4), dtype=np.int32)
            env.step(moves)
            fin = np.zeros(n, int)
            for i in range(n):
                oracle.env_step(ref[i:i + 1], moves[i], status[i])
                if status[i]["done"] or ref["timeStep"][i] >= cap:
                    fin[i] = 1
                    last["winner"][i], last["draw"][i] = status[i]["winner"], status[i]["draw"]
                    last["length"][i], last["alive"][i] = ref["timeStep"][i], ref["aliveAgents"][i]
                    term[i] = ref[i]
                    ref[i] = start[i]
                    status[i] = dict(done=0, winner=-1, draw=0)
            total_fin += fin.sum()
            r = env.last_results()
            assert r["finished"].tolist() == fin.tolist(), f"tick {t}"
            for k in ("winner", "draw", "length", "alive"):
                assert r[k].tolist() == last[k].tolist(), (t, k)
            st = env.status()
            assert not st["done"].any()  # nobody is ever left finished
        assert _same(env.get_state(), ref)
        assert _same(env.get_terminal_state(), term)
        cnt = env.counters()
    assert total_fin > n and cnt[CNT_EPISODES] == cnt[CNT_RESETS] == total_fin and cnt[CNT_STEPS] == n * 150


@pytest.mark.parametrize("dist,kind,tpl", [(DIST_RANDOM, "ffa", 1), (DIST_RANDOM, "ffa", 8), (DIST_STRESS, "stress", 1)])
def test_random_stream_same_games_as_the_start_of_tick_mode(hip_lib, oracle, dist, kind, tpl):
    n, ticks, seed = 4096 + 21, 96, 17
    start = pa.make_boards(n, seed=9, kind=kind)
    want = start.copy()
    steps = oracle.run_random(want, start, ticks, seed, 0, 0, dist, 800)
    done = (want["aliveAgents"] <= 1) | (want["timeStep"] >= 800)
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=800) as env:
        env.make_game(start)
        env.step_random(seed, dist, ticks=ticks, ticks_per_launch=tpl)
        got, term, res, cnt = env.get_state(), env.get_terminal_state(), env.last_results(), env.counters()
    assert done.sum() > 20
    assert res["finished"].astype(bool).tolist() == done.tolist()
    expect = want.copy()
    expect[done] = start[done]  # finished with the last tick: already on the start state again
    assert _same(got, expect)
    assert _same(term[done], want[done])
    assert cnt[CNT_STEPS] == steps == n * ticks


def test_fresh_boards_and_device_policy(hip_lib, oracle):
    """SimpleAgent x4 with a new board per game: the same games as the start-of-tick mode (checked against the oracle elsewhere),
    one restart ahead"""
    n, seed, bseed, cap = 2000, 3, 77, 40
    out = {}
    for mode in (RESET_AT_START, RESET_AT_END):
        with BatchEnvironment(n, mode=MODE_ENV, auto_reset=mode, max_steps=cap, fresh_boards=True, board_seed=bseed) as env:
            env.generate(bseed)
            env.step_simple(seed, 120)  # a multiple of the cap: every game that never ended early times out now
            out[mode] = (env.get_state(), env.episodes(), env.policy_memory(), env.status()["done"].astype(bool), env.counters())
    s1, ep1, mem1, done1, c1 = out[RESET_AT_START]
    s2, ep2, mem2, done2, c2 = out[RESET_AT_END]
    assert done1.sum() > 10 and not done2.any()
    keep = ~done1
    assert s1[keep].tobytes() == s2[keep].tobytes() and np.array_equal(ep1[keep], ep2[keep]) and np.array_equal(mem1[keep], mem2[keep])
    # the games that finished with the last tick: the end-of-tick mode already stands on the next board, with fresh agents
    assert np.array_equal(ep2[done1], ep1[done1] + 1)
    nxt = oracle.boardgen(bseed, np.nonzero(done1)[0], ep2[done1])
    assert s2[done1].tobytes() == nxt.tobytes() and not mem2[done1].any()
    assert c1[CNT_STEPS] == c2[CNT_STEPS] and c1[CNT_EPISODES] == c2[CNT_EPISODES]


def test_results_calls_need_the_mode(hip_lib):
    with BatchEnvironment(64, mode=MODE_ENV, auto_reset=True) as env:
        with pytest.raises(PomError):
            env.last_results()
    with pytest.raises(PomError):
        BatchEnvironment(64, mode=MODE_ENV, auto_reset=3)


def test_moves_from_a_device_tensor_equal_moves_from_the_host(hip_lib):
    """pom_batch_step_device (what an RL loop calls every tick) against pom_batch_step with the same moves; the Python wrapper
    checks what it is handed"""
    import torch
    n = 3000
    start = pa.make_boards(n, seed=12)
    rng = np.random.default_rng(8)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        a = BatchEnvironment(n, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=50, stream=stream.cuda_stream)
        b = BatchEnvironment(n, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=50, stream=stream.cuda_stream)
        a.make_game(start)
        b.make_game(start)
        for t in range(120):
            mv = rng.integers(0, 6, size=(n, 4), dtype=np.int32)
            a.step(mv)
            b.step_device(torch.from_numpy(mv).to("cuda", non_blocking=False))
        assert a.get_state().tobytes() == b.get_state().tobytes()
        assert all(np.array_equal(x, y) for x, y in zip(a.last_results().values(), b.last_results().values()))
        with pytest.raises(ValueError):
            b.step_device(torch.zeros((n, 3), dtype=torch.int32, device="cuda"))
        with pytest.raises(ValueError):
            b.step_device(torch.zeros((n, 4), dtype=torch.int64, device="cuda"))
        with pytest.raises(ValueError):
            b.step_device(torch.zeros((n, 4), dtype=torch.int32))  # a host tensor
        a.close()
        b.close()
