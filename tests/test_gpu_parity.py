"""GPU parity: the HIP path through the C-ABI vs the oracle on identical seeded inputs (bit-exact)."""
import numpy as np
import pytest

import pomcpp_amd as pa
from pomcpp_amd.batch import (BatchEnvironment, MODE_ENV, MODE_RAW, DIST_HARMLESS, DIST_RANDOM, DIST_STRESS,
                              CNT_STEPS, CNT_RESETS)

pytestmark = pytest.mark.gpu


def _assert_same(got, want, what):
    want = want.copy()
    want["agents"]["pad"] = 0
    if got.tobytes() == want.tobytes():
        return
    g = got.view(np.uint8).reshape(got.size, -1)
    w = want.view(np.uint8).reshape(want.size, -1)
    bad = np.nonzero((g != w).any(axis=1))[0]
    e = int(bad[0])
    fields = [f for f in got.dtype.names if got[e][f].tobytes() != want[e][f].tobytes()]
    raise AssertionError(f"{what}: {bad.size}/{got.size} envs differ; first env {e}, fields {fields}\n"
                         f"gpu board:\n{got[e]['board']}\noracle board:\n{want[e]['board']}")


@pytest.mark.parametrize("kind,n,ticks,move_hi", [("ffa", 1000, 60, 6), ("ffa", 64, 200, 5), ("stress", 777, 40, 6)])
def test_explicit_moves_raw_step_matches_oracle_every_tick(hip_lib, oracle, kind, n, ticks, move_hi):
    rng = np.random.default_rng(11)
    ref = pa.make_boards(n, seed=5, kind=kind)
    with BatchEnvironment(n, mode=MODE_RAW) as env:
        env.make_game(ref)
        for t in range(ticks):
            moves = rng.integers(0, move_hi, size=(n, 4), dtype=np.int32)
            env.step(moves)
            flags = oracle.step_batch(ref, moves)
            got = env.get_state()
            _assert_same(got, ref, f"{kind} tick {t}")
            st = env.status()
            # flags are sticky on the device; every flag the oracle raised this tick must be present
            assert np.all((st["ubflags"] & flags) == flags)


@pytest.mark.parametrize("dist,kind", [(DIST_RANDOM, "ffa"), (DIST_HARMLESS, "ffa"), (DIST_STRESS, "stress")])
def test_random_stream_env_mode_autoreset_matches_oracle(hip_lib, oracle, dist, kind):
    n, ticks, seed = 4096 + 37, 120, 99
    start = pa.make_boards(n, seed=21, kind=kind)
    want = start.copy()
    steps = oracle.run_random(want, start, ticks, seed, 0, 0, dist, 800)
    for tpl in (1, 8):
        with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800) as env:
            env.make_game(start)
            env.step_random(seed, dist, ticks=ticks, ticks_per_launch=tpl)
            got = env.get_state()
            cnt = env.counters()
        _assert_same(got, want, f"dist {dist} ticks_per_launch {tpl}")
        assert cnt[CNT_STEPS] == steps == n * ticks
        if dist != DIST_HARMLESS:
            assert cnt[CNT_RESETS] > 0


def test_env_mode_freezes_finished_envs_and_reports_winner(hip_lib, oracle):
    n = 300
    start = pa.make_boards(n, seed=2)
    rng = np.random.default_rng(3)
    ref = start.copy()
    status = [dict(done=0, winner=-1, draw=0) for _ in range(n)]
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=False) as env:
        env.make_game(start)
        for t in range(150):
            moves = rng.integers(0, 6, size=(n, 4), dtype=np.int32)
            env.step(moves)
            for i in range(n):
                oracle.env_step(ref[i:i + 1], moves[i], status[i])
        got = env.get_state()
        st = env.status()
    _assert_same(got, ref, "env mode")
    assert st["done"].tolist() == [s["done"] for s in status]
    assert st["winner"].tolist() == [s["winner"] for s in status]
    assert st["draw"].tolist() == [s["draw"] for s in status]
    assert st["done"].sum() > n // 2
