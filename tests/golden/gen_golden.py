#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED, UNMODIFIED reference (oracle/_ref/libpomref.so).

Runs in the build container only (needs /root/reference to have been compiled by `make -C oracle ref`).
The fixtures are data — input states, Move[4] and the reference's output states / state hashes —
never reference source.

  step_cases.npz    every Step of the 32 restated [step function] leaf runs and of the directed
                    quirk vectors (tests/step_cases.py): state before, moves, state after.
  trajectories.npz  random-play episodes (configs 1/2/5 distributions): start state, per-tick moves,
                    per-tick blake2b-64 of the reference's state, full state every 32 ticks + final.
                    The reference is never stepped on a tick for which the oracle predicts one of
                    its crashing UBs (SURVEY §8c guard); such an episode ends there.
  env_traces.npz    games played by the reference's Environment::Step (environment.cpp:123-169, row a12) with
                    play-back agents: start state, per-tick moves, per-tick done / winner / draw, which agents
                    were asked, per-tick state hash, final state.  A game ends before a tick on which the
                    reference's uninitialised Move entries (dead agents; moves[-1]) could matter.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import pomcpp_amd as pa  # noqa: E402
from pomcpp_amd.state import STATE_DTYPE  # noqa: E402
from tests.case_api import RefAPI  # noqa: E402
from tests.oracle_lib import Oracle  # noqa: E402
from tests.step_cases import ALL_CASES  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FATAL = 2 | 4 | 8 | 16  # NULL_BOMB | QUEUE_OVERFLOW | REVERT_LOOP | BAD_INDEX


def state_hash(buf: bytes) -> int:
    return int.from_bytes(hashlib.blake2b(buf, digest_size=8).digest(), "little")


def gen_cases():
    out = {}
    total = 0
    for name, fn in ALL_CASES.items():
        api = RefAPI()
        fn(api)
        k = len(api.trace)
        total += k
        out[f"{name}__before"] = np.frombuffer(b"".join(t[0] for t in api.trace), dtype=np.uint8).reshape(k, 1004)
        out[f"{name}__moves"] = np.array([t[1] for t in api.trace], dtype=np.int32).reshape(k, 4)
        out[f"{name}__after"] = np.frombuffer(b"".join(t[2] for t in api.trace), dtype=np.uint8).reshape(k, 1004)
        print(f"  case {name}: {k} steps, {api.checks} assertions hold on the reference")
    np.savez_compressed(os.path.join(OUT, "step_cases.npz"), **out)
    print(f"step_cases.npz: {len(ALL_CASES)} cases, {total} steps")


def gen_trajectories():
    ref = RefAPI().lib
    oracle = Oracle()
    rng = np.random.default_rng(20261003)
    plan = [("ffa", 6, 48, 300), ("ffa", 5, 12, 400), ("stress", "stress", 48, 120)]
    starts, moves_all, hashes_all, offsets, ck_ep, ck_tick, ck_state, finals, kinds = [], [], [], [0], [], [], [], [], []
    for kind, dist, episodes, max_ticks in plan:
        boards = pa.make_boards(episodes, seed=int(rng.integers(1 << 30)), kind=kind)
        for e in range(episodes):
            s = boards[e:e + 1].copy()
            starts.append(s.tobytes())
            kinds.append(0 if kind == "ffa" else 1)
            n = 0
            for t in range(max_ticks):
                if dist == "stress":
                    mv = rng.choice(6, size=4, p=[.10, .15, .15, .15, .15, .30]).astype(np.int32)
                else:
                    mv = rng.integers(0, dist, size=4, dtype=np.int32)
                probe = s.copy()
                if oracle.step(probe, mv) & FATAL:
                    break  # the reference would crash on this tick
                ref.ref_step(s.ctypes.data, mv.ctypes.data)
                s["agents"]["pad"] = 0
                moves_all.append(mv)
                hashes_all.append(state_hash(s.tobytes()))
                n += 1
                if n % 32 == 0:
                    ck_ep.append(len(starts) - 1)
                    ck_tick.append(n)
                    ck_state.append(s.tobytes())
                if int(s["aliveAgents"][0]) <= 1:
                    break
            finals.append(s.tobytes())
            offsets.append(offsets[-1] + n)
    E = len(starts)
    np.savez_compressed(
        os.path.join(OUT, "trajectories.npz"),
        start=np.frombuffer(b"".join(starts), dtype=np.uint8).reshape(E, 1004),
        final=np.frombuffer(b"".join(finals), dtype=np.uint8).reshape(E, 1004),
        moves=np.array(moves_all, dtype=np.int32).reshape(-1, 4),
        hashes=np.array(hashes_all, dtype=np.uint64),
        offsets=np.array(offsets, dtype=np.int64),
        kind=np.array(kinds, dtype=np.int8),
        ck_episode=np.array(ck_ep, dtype=np.int32), ck_tick=np.array(ck_tick, dtype=np.int32),
        ck_state=np.frombuffer(b"".join(ck_state), dtype=np.uint8).reshape(-1, 1004),
    )
    print(f"trajectories.npz: {E} episodes, {offsets[-1]} reference steps, {len(ck_state)} checkpoints")


def gen_env_traces():
    import ctypes as C
    import itertools
    ref = RefAPI().lib
    ref.ref_env_new.restype = C.c_void_p
    ref.ref_env_new.argtypes = [C.c_void_p]
    ref.ref_env_delete.argtypes = [C.c_void_p]
    ref.ref_env_step.argtypes = [C.c_void_p] * 3 + [C.POINTER(C.c_int)] * 3
    oracle = Oracle()
    rng = np.random.default_rng(20261004)
    plan = [("ffa", 6, 40, 200), ("ffa", 5, 8, 300), ("stress", "stress", 32, 100)]
    starts, finals, moves_all, hashes_all, status_all, offsets = [], [], [], [], [], [0]
    for kind, dist, games, max_ticks in plan:
        boards = pa.make_boards(games, seed=int(rng.integers(1 << 30)), kind=kind)
        for e in range(games):
            s = boards[e:e + 1].copy()
            starts.append(s.tobytes())
            g = ref.ref_env_new(s.ctypes.data)
            n = 0
            for t in range(max_ticks):
                if dist == "stress":
                    mv = rng.choice(6, size=4, p=[.10, .15, .15, .15, .15, .30]).astype(np.int32)
                else:
                    mv = rng.integers(0, dist, size=4, dtype=np.int32)
                dead = [i for i in range(4) if s["agents"]["dead"][0, i]]
                mv[dead] = 0  # what the restatement and the device define for agents that are not asked
                probe = s.copy()
                if oracle.step(probe, mv):
                    break  # any UB flag: the reference's moves[-1] / null deref would be in play
                sensitive = False
                for combo in itertools.product(range(5), repeat=len(dead)):
                    m2 = mv.copy()
                    m2[dead] = combo
                    p2 = s.copy()
                    oracle.step(p2, m2)
                    if p2.tobytes() != probe.tobytes():
                        sensitive = True
                        break
                if sensitive:
                    break
                d, w, dr = C.c_int(), C.c_int(), C.c_int()
                asked = ref.ref_env_step(g, mv.ctypes.data, s.ctypes.data, C.byref(d), C.byref(w), C.byref(dr))
                s["agents"]["pad"] = 0
                moves_all.append(mv)
                hashes_all.append(state_hash(s.tobytes()))
                status_all.append((d.value, w.value, dr.value, asked))
                n += 1
                if d.value:
                    break
            ref.ref_env_delete(g)
            finals.append(s.tobytes())
            offsets.append(offsets[-1] + n)
    E = len(starts)
    st = np.array(status_all, dtype=np.int32).reshape(-1, 4)
    np.savez_compressed(
        os.path.join(OUT, "env_traces.npz"),
        start=np.frombuffer(b"".join(starts), dtype=np.uint8).reshape(E, 1004),
        final=np.frombuffer(b"".join(finals), dtype=np.uint8).reshape(E, 1004),
        moves=np.array(moves_all, dtype=np.int32).reshape(-1, 4),
        hashes=np.array(hashes_all, dtype=np.uint64),
        status=st, offsets=np.array(offsets, dtype=np.int64),
    )
    print(f"env_traces.npz: {E} games, {offsets[-1]} Environment::Step calls, {int(st[:, 0].sum())} finished "
          f"({int(st[:, 2].sum())} draws)")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "env":  # only that fixture; the others are byte-stable and stay as committed
        gen_env_traces()
    elif len(sys.argv) > 1 and sys.argv[1] == "cases":  # after adding cases to tests/step_cases.py
        gen_cases()
    else:
        gen_cases()
        gen_trajectories()
        gen_env_traces()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
