"""Golden vectors recorded from the compiled reference (tests/golden/gen_golden.py): every recorded Step
replayed in isolation, and whole random-play episodes replayed tick by tick against the per-tick
state hashes, the checkpoints and the final state."""
import hashlib
import os

import numpy as np
import pytest

from pomcpp_amd.state import STATE_DTYPE

HERE = os.path.join(os.path.dirname(__file__), "golden")
CASES = np.load(os.path.join(HERE, "step_cases.npz"))
TRAJ = np.load(os.path.join(HERE, "trajectories.npz"))


def _hash(buf: bytes) -> int:
    return int.from_bytes(hashlib.blake2b(buf, digest_size=8).digest(), "little")


def _all_steps():
    names = sorted({k.split("__")[0] for k in CASES.files})
    before = np.concatenate([CASES[f"{n}__before"] for n in names])
    moves = np.concatenate([CASES[f"{n}__moves"] for n in names])
    after = np.concatenate([CASES[f"{n}__after"] for n in names])
    return before, moves, after


def test_oracle_reproduces_every_recorded_step(oracle):
    before, moves, after = _all_steps()
    st = np.frombuffer(before.tobytes(), dtype=STATE_DTYPE).copy()
    oracle.step_batch(st, moves)
    st["agents"]["pad"] = 0
    assert st.tobytes() == after.tobytes()


def _replay_episodes(step_all):
    """step_all(states[E], moves[E,4], active[E]) advances the active episodes by one tick in place."""
    start, off = TRAJ["start"], TRAJ["offsets"]
    E = len(start)
    st = np.frombuffer(start.tobytes(), dtype=STATE_DTYPE).copy()
    lengths = np.diff(off)
    ck = {(int(e), int(t)): i for i, (e, t) in enumerate(zip(TRAJ["ck_episode"], TRAJ["ck_tick"]))}
    compared = 0
    for t in range(int(lengths.max())):
        active = lengths > t
        mv = np.zeros((E, 4), dtype=np.int32)
        mv[active] = TRAJ["moves"][off[:-1][active] + t]
        step_all(st, mv, active)
        st["agents"]["pad"] = 0
        for e in np.nonzero(active)[0]:
            assert _hash(st[e:e + 1].tobytes()) == int(TRAJ["hashes"][off[e] + t]), f"episode {e} tick {t}"
            compared += 1
            if (int(e), t + 1) in ck:
                assert st[e:e + 1].tobytes() == TRAJ["ck_state"][ck[(int(e), t + 1)]].tobytes()
    assert st.tobytes() == TRAJ["final"].tobytes()
    assert compared == int(off[-1])


def test_oracle_replays_reference_episodes(oracle):
    def step_all(st, mv, active):
        for e in np.nonzero(active)[0]:
            oracle.step(st[e:e + 1], mv[e])
    _replay_episodes(step_all)


@pytest.mark.gpu
def test_gpu_reproduces_every_recorded_step(hip_lib):
    from pomcpp_amd.batch import BatchEnvironment, MODE_RAW
    before, moves, after = _all_steps()
    st = np.frombuffer(before.tobytes(), dtype=STATE_DTYPE)
    with BatchEnvironment(len(st), mode=MODE_RAW) as env:
        env.make_game(st)
        env.step(moves)
        got = env.get_state()
    assert got.tobytes() == after.tobytes()


@pytest.mark.gpu
def test_gpu_replays_reference_episodes(hip_lib):
    from pomcpp_amd.batch import BatchEnvironment, MODE_RAW
    E = len(TRAJ["start"])
    with BatchEnvironment(E, mode=MODE_RAW) as env:
        def step_all(st, mv, active):
            # finished episodes keep receiving IDLE moves on the device; only active ones are read back
            env.make_game(st)
            env.step(mv)
            got = env.get_state()
            st[active] = got[active]
        _replay_episodes(step_all)


# ---- Environment::Step: games recorded from the compiled reference Environment -----------------------
ENVT = np.load(os.path.join(HERE, "env_traces.npz"))


def _replay_env_games(step_all):
    """step_all(states[E], moves[E,4], active[E]) -> (done[E], winner[E], draw[E]) after one Environment::Step of every game;
    games past their recorded end keep receiving the call (finished ones must stay frozen, the others are ignored)"""
    start, off = ENVT["start"], ENVT["offsets"]
    E = len(start)
    st = np.frombuffer(start.tobytes(), dtype=STATE_DTYPE).copy()
    lengths = np.diff(off)
    compared = 0
    for t in range(int(lengths.max())):
        active = lengths > t
        mv = np.zeros((E, 4), dtype=np.int32)
        mv[active] = ENVT["moves"][off[:-1][active] + t]
        alive_before = ~st["agents"]["dead"].astype(bool)
        done, winner, draw = step_all(st, mv, active)
        st["agents"]["pad"] = 0
        for e in np.nonzero(active)[0]:
            rec = ENVT["status"][off[e] + t]
            assert _hash(st[e:e + 1].tobytes()) == int(ENVT["hashes"][off[e] + t]), f"game {e} tick {t}"
            assert (int(done[e]), int(winner[e]), int(draw[e])) == (int(rec[0]), int(rec[1]), int(rec[2])), f"game {e} tick {t}"
            # the reference asked exactly the live agents; the moves of the others are IDLE in the fixture
            assert int(rec[3]) == sum(int(alive_before[e, i]) << i for i in range(4))
            compared += 1
    final = np.frombuffer(ENVT["final"].tobytes(), dtype=STATE_DTYPE)
    ended = ENVT["status"][off[1:] - 1][:, 0] == 1  # games the reference finished: frozen ever since
    assert st[ended].tobytes() == final[ended].tobytes()
    assert compared == int(off[-1]) and ended.sum() > len(ended) // 2


def test_oracle_replays_reference_environment_games(oracle):
    E = len(ENVT["start"])
    status = [dict(done=0, winner=-1, draw=0) for _ in range(E)]

    def step_all(st, mv, active):
        for e in range(E):
            if active[e] or status[e]["done"]:
                oracle.env_step(st[e:e + 1], mv[e], status[e])
        return ([s["done"] for s in status], [s["winner"] for s in status], [s["draw"] for s in status])
    _replay_env_games(step_all)


@pytest.mark.gpu
def test_gpu_replays_reference_environment_games(hip_lib):
    from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
    E = len(ENVT["start"])
    with BatchEnvironment(E, mode=MODE_ENV, auto_reset=False) as env:
        env.make_game(np.frombuffer(ENVT["start"].tobytes(), dtype=STATE_DTYPE))

        def step_all(st, mv, active):
            env.step(mv)  # the device keeps the games; finished ones are not stepped (environment.cpp:125-128)
            st[:] = env.get_state()
            s = env.status()
            return s["done"], s["winner"], s["draw"]
        _replay_env_games(step_all)


@pytest.mark.gpu
def test_gpu_one_state_env_step_replays_reference_environment_games(hip_lib):
    """The same games through pom_env_step: one host State per call, the tick and Environment::Step's bookkeeping in one launch
    (what bboard::Environment::Step of include/pom_bboard.hpp calls).  The caller keeps finished games away from it
    (environment.cpp:125-128)."""
    from pomcpp_amd.batch import env_step_one
    E = len(ENVT["start"])
    status = [dict(done=0, winner=-1, draw=0) for _ in range(E)]

    def step_all(st, mv, active):
        for e in range(E):
            if active[e] and not status[e]["done"]:
                r = env_step_one(st[e:e + 1], mv[e])
                status[e].update(done=r["done"], draw=r["draw"])
                if r["winner"] >= 0:
                    status[e]["winner"] = r["winner"]
        return ([s["done"] for s in status], [s["winner"] for s in status], [s["draw"] for s in status])
    _replay_env_games(step_all)
