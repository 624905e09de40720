"""the SimpleAgent policy.  CPU: the device policy body (host build) vs the policy oracle under play.
GPU: pom_batch_policy_simple / step_simple through the C-ABI vs the oracle — moves, agent memory and states, bit-exact."""
import os
import subprocess

import numpy as np
import pytest

import pomcpp_amd as pa
from pomcpp_amd.state import Move

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build")
INC = ["-I" + os.path.join(ROOT, p) for p in ("include", "pomcpp_amd/csrc", "oracle")]


@pytest.mark.parametrize("floods", ["one-lane", "quad"])
@pytest.mark.parametrize("scenario", [1, 2])
def test_device_policy_body_matches_oracle_under_play(scenario, floods):
    """floods: the two searches of an act() as one lane runs them (four registers per cell set), or through the quad-word level
    functions the kernels' wave-cooperative floods are made of (pom_quad_*_level, pom_window_word)"""
    os.makedirs(BUILD, exist_ok=True)
    run = lambda *a: subprocess.run(list(a), check=True, cwd=ROOT)
    run("g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", *INC, "-c", "tests/emul/pom_policy_emul.cpp", "-o", "build/pom_policy_emul.o")
    run("gcc", "-O2", "-std=c11", *INC, "-c", "oracle/pom_policy_oracle.c", "-o", "build/pom_policy_oracle.o")
    run("gcc", "-O2", "-std=c11", *INC, "-c", "oracle/pom_oracle.c", "-o", "build/pom_oracle_p.o")
    run("g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", *INC, "tests/emul/emul_policy_fuzz.cpp", "build/pom_policy_emul.o",
        "build/pom_policy_oracle.o", "build/pom_oracle_p.o", "-o", "build/emul_policy_fuzz")
    out = subprocess.run([os.path.join(BUILD, "emul_policy_fuzz"), str(scenario), "100000", "3"] + (["quad"] if floods == "quad" else []),
                         capture_output=True, text=True)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout[-2000:]


def test_window_word_by_word_equals_the_window():
    import ctypes as C
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libpom_policy_emul.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-fPIC", "-shared", *INC, "tests/emul/pom_policy_emul.cpp", "-o", so],
                   check=True, cwd=ROOT)
    assert C.CDLL(so).pom_emul_window_check() == 0  # every agent cell x radius 0..15, all four words


def test_oracle_policy_basics(oracle):
    """Hand-checkable situations: an agent next to a wood bombs it; an agent on a ticking bomb walks away."""
    s = pa.new_states(1)
    pa.put_agents_in_corners(s[0])
    pa.put_item(s[0], 1, 0, pa.Item.WOOD)
    mem = np.zeros((1, 4, 16), dtype=np.int32)
    mv = oracle.simple_policy(s, mem, seed=1, first_env=0, tick=0)
    assert mv[0, 0] == Move.BOMB                      # IsAdjacentItem(WOOD) -> BOMB (simple_agent.cpp:100-103)
    assert mem[0, 0, 9] == 1 and mem[0, 0, 0:2].tolist() == [0, 0]   # recentPositions got its cell
    pa.plant_bomb(s[0], 10, 10, 2, True, 3)           # a bomb under agent 2, 3 ticks left
    mv = oracle.simple_policy(s, mem, seed=1, first_env=0, tick=1)
    assert mv[0, 2] in (Move.UP, Move.LEFT)           # in danger: MoveTowardsSafePlace / a safe direction


@pytest.mark.gpu
@pytest.mark.parametrize("kind,streams", [("ffa", 1), ("ffa", 2), ("stress", 1)])
def test_gpu_policy_moves_memory_and_states_match_oracle(hip_lib, oracle, kind, streams):
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    n, ticks, seed = 1500, 120, 77
    start = pa.make_boards(n, seed=13, kind=kind)
    # make agents meet early in a third of the envs so that the enemy / loop-breaking branches run
    for e in range(0, n, 3):
        if start[e]["board"][5, 4] == 0 and start[e]["board"][5, 6] == 0:
            start[e]["board"][0, 0] = 0
            start[e]["board"][0, 10] = 0
            pa.put_agent(start[e], 4, 5, 0)
            pa.put_agent(start[e], 6, 5, 1)
    ref = start.copy()
    mem = np.zeros((n, 4, 16), dtype=np.int32)
    status = [dict(done=0, winner=-1, draw=0) for _ in range(n)]
    hist = np.zeros(6, dtype=np.int64)
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=False, streams=streams) as env:
        env.make_game(start)
        for t in range(ticks):
            got_mv = env.policy_simple(seed, want_moves=True)
            done = np.array([s["done"] for s in status], dtype=np.int32)
            want_mv = oracle.simple_policy(ref, mem, seed, 0, t, done)
            assert np.array_equal(got_mv, want_mv), f"tick {t}: moves differ in envs {np.nonzero((got_mv != want_mv).any(1))[0][:8]}"
            assert np.array_equal(env.policy_memory(), mem), f"tick {t}: agent memory differs"
            hist += np.bincount(want_mv.ravel(), minlength=6)
            env.step_policy()
            for e in range(n):
                oracle.env_step(ref[e:e + 1], want_mv[e], status[e])
        got = env.get_state()
    ref["agents"]["pad"] = 0
    assert got.tobytes() == ref.tobytes()
    assert hist.min() > 0  # every kind of move was played


@pytest.mark.gpu
@pytest.mark.parametrize("n,streams,shape", [(4096 + 5, 1, {}), (4096 + 5, 3, {}),
                                             # one lane per env: the policy kernel and the tick stay two launches
                                             (2048 + 3, 2, dict(lanes_per_env=1, envs_per_wave=16)), (1000, 1, dict(envs_per_wave=64))])
def test_gpu_step_simple_with_autoreset_matches_oracle(hip_lib, oracle, n, streams, shape):
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, CNT_STEPS, CNT_RESETS
    ticks, seed = 150, 5
    start = pa.make_boards(n, seed=4)
    want = start.copy()
    mem = np.zeros((n, 4, 16), dtype=np.int32)
    steps = oracle.run_simple(want, start, mem, ticks, seed, 0, 0, 800)
    want["agents"]["pad"] = 0
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=streams, **shape) as env:
        env.make_game(start)
        env.step_simple(seed, ticks)
        got = env.get_state()
        assert got.tobytes() == want.tobytes()
        assert np.array_equal(env.policy_memory(), mem)
        cnt = env.counters()
        assert cnt[CNT_STEPS] == steps == n * ticks and cnt[CNT_RESETS] > 0
        # uploading envs again gives them fresh agents
        env.make_game(start[:10], first=0)
        assert not env.policy_memory(0, 10).any() and env.policy_memory(10, 5).any()


@pytest.mark.gpu
def test_mixed_play_through_the_device_move_buffer(hip_lib, oracle):
    """agent 0 follows an outside policy (here: always IDLE unless dead) written into the device move buffer, agents 1..3 are
    SimpleAgents: policy_simple -> overwrite column 0 on the handle's stream -> step_policy, against the oracle"""
    import torch
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    n, seed = 600, 4
    start = pa.make_boards(n, seed=8)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=False, stream=stream.cuda_stream)
        env.make_game(start)
        ref = start.copy()
        mems = np.zeros((n, 4, 16), dtype=np.int32)
        for t in range(40):
            env.policy_simple(seed)
            mv = env.moves_tensor()
            mv[:, 0] = int(Move.IDLE)
            env.step_policy()
            done = (ref["aliveAgents"] <= 1).astype(np.int32)
            want = oracle.simple_policy(ref, mems, seed, 0, t, done)
            want[:, 0] = int(Move.IDLE)
            for e in range(n):
                if not done[e]:
                    oracle.step(ref[e:e + 1], want[e])
                    ref["timeStep"][e] += 1
        assert env.get_state().tobytes() == ref.tobytes()
        assert np.array_equal(env.policy_memory(), mems)
        with pytest.raises(RuntimeError):
            env.close()  # a view of the handle's move buffer is still alive
        del mv
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 1000])
def test_gpu_policy_every_agent_of_a_wavefront_searches(hip_lib, oracle, n):
    """The searches of all agents of a wavefront run together, 16 at a time (pom_coop_forward / pom_coop_backward): here EVERY
    live agent of every env stands on a ticking bomb (64 forward jobs per wavefront: four rounds) and most then need a path
    (several backward rounds) — moves, memory and the tick that follows must equal the oracle's; n = 1000 leaves the last
    wavefront partly empty."""
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    start = pa.make_boards(n, seed=99)
    rng = np.random.default_rng(3)
    for e in range(n):
        for a in range(4):
            start[e]["agents"][a]["maxBombCount"] = 3
            x, y = int(start[e]["agents"][a]["x"]), int(start[e]["agents"][a]["y"])
            pa.plant_bomb(start[e], x, y, a, False, int(rng.integers(2, 9)))  # under the agent: the board keeps showing him
        for c in rng.integers(0, 121, size=30):  # open the boards up a little: longer searches
            yy, xx = divmod(int(c), 11)
            if start[e]["board"][yy, xx] in (1, 2 << 8) or (int(start[e]["board"][yy, xx]) >> 8) == 2:
                start[e]["board"][yy, xx] = 0
    ref = start.copy()
    mem = np.zeros((n, 4, 16), dtype=np.int32)
    status = [dict(done=0, winner=-1, draw=0) for _ in range(n)]
    with BatchEnvironment(n, mode=MODE_ENV, auto_reset=False) as env:
        env.make_game(start)
        for t in range(12):
            done = np.array([st["done"] for st in status], dtype=np.int32)
            want = oracle.simple_policy(ref, mem, 5, 0, t, done)
            got = env.policy_simple(5, want_moves=True)
            assert np.array_equal(got, want), (t, np.argwhere(got != want)[:5])
            assert np.array_equal(env.policy_memory(), mem), t
            if t == 0:  # the first call must really have loaded every quad: all agents are in danger and walk away
                assert (want != 0).mean() > 0.5
            env.step_policy()
            for e in range(n):
                oracle.env_step(ref[e:e + 1], want[e], status[e])
        got_state = env.get_state()
    ref["agents"]["pad"] = 0
    assert got_state.tobytes() == ref.tobytes()
