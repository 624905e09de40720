#!/usr/bin/env python3
"""Generate tests/golden/policy_traces.npz from the COMPILED, UNMODIFIED reference agent (oracle/_ref/libpomref.so:
src/agents/simple_agent.cpp + src/bboard/strategy.cpp behind oracle/ref_shim.cpp's ref_simple_* window).

Runs in the build container only (needs /root/reference compiled by `make -C oracle ref`).  The fixture is data — start states,
the draw each act() was handed, the Move it returned and the agent memory it left — never reference source.

  games      four reference SimpleAgents play `E` games over four kinds of start state (reference board distribution; the same with
             two agents dropped next to each other; the kick / chain stress boards; power-up rich boards with mixed agents), ticks by
             the reference's own bboard::Step.  The one random draw of an act() (simple_agent.cpp:19-21,47: a mt19937_64 member) is the
             value the synthetic stream of include/pom_rng.h hands agent i of env e on tick t — the shim reseeds the agent's public `rng`
             until its next draw is that value — so that the DEVICE policy, which draws from that stream, can be replayed against the
             fixture as well as the oracle.  Agent memory starts zeroed (`new SimpleAgent()` leaves its queues' raw slots to the heap).
             A game ends when one agent is left, after `CAP` ticks, or before a tick on which the reference's Step would crash
             (SURVEY §8c guard).
  strategy   the helpers the reference's [strategy] tests call (unit_test/bboard/strategy_test.cpp), on hand-made boards:
             IsAdjacentEnemy for distances 0..4 (incl. the test's own two placements), FillRMap's raw map and MoveTowardsPosition
             towards every reachable cell (incl. the placements of "Move Towards Methods").
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import pomcpp_amd as pa  # noqa: E402
from pomcpp_amd.state import Item, new_states  # noqa: E402
from tests.oracle_lib import Oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "policy_traces.npz")
SEED, CAP = 20261005, 200
FATAL = 2 | 4 | 8 | 16  # NULL_BOMB | QUEUE_OVERFLOW | REVERT_LOOP | BAD_INDEX


def ref_lib():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpomref.so"))
    lib.ref_simple_new.restype = C.c_void_p
    lib.ref_simple_new.argtypes = [C.c_int, C.c_ulonglong]
    lib.ref_simple_delete.argtypes = [C.c_void_p]
    lib.ref_simple_peek_draw.argtypes = [C.c_void_p]
    lib.ref_simple_act.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_simple_memory.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_simple_set_memory.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_step.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_is_adjacent_enemy.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.ref_fill_rmap.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def start_states(rng) -> np.ndarray:
    """52 games of each kind"""
    a = pa.make_boards(52, seed=int(rng.integers(1 << 30)))
    b = pa.make_boards(52, seed=int(rng.integers(1 << 30)))
    for e in range(52):  # let them meet: two agents next to each other in the middle
        s = b[e]
        s["board"][0, 0] = s["board"][0, 10] = Item.PASSAGE
        for i, (x, y) in enumerate(((4, 5), (6, 5))):
            s["board"][y, x] = Item.AGENT0 + i
            s["agents"]["x"][i], s["agents"]["y"][i] = x, y
        s["board"][5, 5] = Item.PASSAGE
    c = pa.make_boards(52, seed=int(rng.integers(1 << 30)), kind="stress")
    d = pa.make_boards(52, seed=int(rng.integers(1 << 30)))
    for e in range(52):  # power-up rich, mixed agents
        s = d[e]
        cells = s["board"].reshape(-1)
        wood = (cells >> 8) == 2
        cells[wood] = Item.WOOD + rng.integers(1, 5, size=int(wood.sum()))
        extra = (cells == Item.PASSAGE) & (rng.random(121) < 0.25)
        cells[extra] = Item.WOOD + rng.integers(1, 4, size=int(extra.sum()))
        for i in range(4):
            s["agents"]["canKick"][i] = rng.integers(0, 2)
            s["agents"]["maxBombCount"][i] = rng.integers(1, 4)
            s["agents"]["bombStrength"][i] = rng.integers(1, 6)
    return np.concatenate([a, b, c, d])


def gen_games(lib, oracle):
    rng = np.random.default_rng(SEED)
    start = start_states(rng)
    n = start.size
    length = np.zeros(n, dtype=np.int32)
    draws = np.zeros((n, CAP, 4), dtype=np.int8)
    moves = np.zeros((n, CAP, 4), dtype=np.int8)
    asked = np.zeros((n, CAP, 4), dtype=np.int8)
    mem = np.zeros((n, CAP, 4, 16), dtype=np.int8)
    acts = 0
    hist = np.zeros(6, dtype=np.int64)
    for e in range(n):
        s = start[e:e + 1].copy()
        agents = [lib.ref_simple_new(i, 1) for i in range(4)]
        zero = np.zeros(16, dtype=np.int32)
        for a in agents:
            lib.ref_simple_set_memory(a, zero.ctypes.data)
        t = 0
        while t < CAP and s["aliveAgents"][0] > 1:
            mv = np.zeros(4, dtype=np.int32)
            for i in range(4):
                if s["agents"]["dead"][0, i]:
                    continue
                want = int(oracle.lib.pom_oracle_policy_draw(SEED, e, t, i))
                # the reference draws from its own generator: reseed it until the draw it is about to make is the stream's
                a_obj = C.cast(agents[i], C.c_void_p)
                keep = np.zeros(16, dtype=np.int32)
                lib.ref_simple_memory(a_obj, keep.ctypes.data)
                k = 0
                while True:
                    fresh = lib.ref_simple_new(i, 7919 * (acts + 1) + k)
                    if lib.ref_simple_peek_draw(fresh) == want:
                        break
                    lib.ref_simple_delete(fresh)
                    k += 1
                lib.ref_simple_set_memory(fresh, keep.ctypes.data)
                lib.ref_simple_delete(agents[i])
                agents[i] = fresh
                m = lib.ref_simple_act(fresh, s.ctypes.data)
                out = np.zeros(16, dtype=np.int32)
                lib.ref_simple_memory(fresh, out.ctypes.data)
                assert np.abs(out).max() < 128 and 0 <= m <= 5
                draws[e, t, i], moves[e, t, i], asked[e, t, i], mem[e, t, i] = want, m, 1, out
                mv[i] = m
                hist[m] += 1
                acts += 1
            probe = s.copy()
            if oracle.step(probe, mv) & FATAL:
                asked[e, t] = 0  # this tick is not part of the game: the reference's Step would crash on it
                break
            lib.ref_step(s.ctypes.data, mv.ctypes.data)  # (the shim pads the array: Step reads moves[-1] on lost-agent ticks, Q-UB1)
            s["timeStep"][0] += 1
            t += 1
        length[e] = t
        for a in agents:
            lib.ref_simple_delete(a)
    total = int(asked[np.arange(CAP)[None, :] < length[:, None]].sum())
    print(f"games: {n}, ticks {int(length.sum())}, act() calls {total}; moves idle/up/down/left/right/bomb {hist.tolist()}")
    return dict(game_start=start.view(np.uint8).reshape(n, 1004), game_length=length, game_draws=draws, game_moves=moves,
                game_asked=asked, game_memory=mem, game_seed=np.array([SEED], dtype=np.int64))


def gen_strategy(lib):
    rng = np.random.default_rng(SEED + 1)
    states, adj, rmaps, move_to = [], [], [], []

    def record(s):
        a = np.zeros((4, 5), dtype=np.int8)
        for i in range(4):
            for d in range(5):
                a[i, d] = lib.ref_is_adjacent_enemy(s.ctypes.data, i, d)
        m, mt = np.zeros(121, dtype=np.int32), np.zeros(121, dtype=np.int32)
        lib.ref_fill_rmap(s.ctypes.data, 0, m.ctypes.data, mt.ctypes.data)
        states.append(s.copy().view(np.uint8).reshape(1004))
        adj.append(a)
        rmaps.append(m)
        move_to.append(mt.astype(np.int8))

    def place(s, x, y, i):  # State::PutAgent
        s["board"][0, y, x] = Item.AGENT0 + i
        s["agents"]["x"][0, i], s["agents"]["y"][0, i] = x, y

    # strategy_test.cpp "IsAdjacent": (5,5) / (4,4) and (5,5) / (3,2), on the empty board of a fresh State
    for other in ((4, 4), (3, 2)):
        s = new_states(1)
        place(s, 5, 5, 0)
        place(s, *other, 1)
        record(s)
    # "Fill RMap" / "Move Towards Methods": agents 1..3 killed, agent 0 at (0,0) / (4,5), on hand-made boards (the tests' own boards
    # come from InitBoardItems' libstdc++ stream, which is not reproduced anywhere)
    for k in range(60):
        s = pa.make_boards(1, seed=int(rng.integers(1 << 30)), kind="stress" if k % 3 == 2 else "ffa")
        for i in (1, 2, 3):
            if k % 2 == 0:
                s["agents"]["dead"][0, i] = 1
                s["aliveAgents"][0] -= 1
        x, y = ((0, 0), (4, 5), (int(rng.integers(0, 11)), int(rng.integers(0, 11))))[k % 3]
        s["board"][0, 0, 0] = Item.PASSAGE
        s["board"][0, y, x] = Item.PASSAGE
        place(s, x, y, 0)
        if k % 5 == 0:  # power-ups on the way (FillRMap walks over them)
            cells = s["board"].reshape(-1)
            free = np.nonzero(cells == Item.PASSAGE)[0]
            cells[rng.choice(free, size=min(6, free.size), replace=False)] = Item.EXTRABOMB + rng.integers(0, 3, size=min(6, free.size))
        record(s)
    print(f"strategy vectors: {len(states)} states")
    return dict(strat_state=np.stack(states), strat_adjacent=np.stack(adj), strat_rmap=np.stack(rmaps), strat_move_to=np.stack(move_to))


def main():
    lib, oracle = ref_lib(), Oracle()
    out = {}
    out.update(gen_games(lib, oracle))
    out.update(gen_strategy(lib))
    np.savez_compressed(OUT, **out)
    print(f"{OUT}: {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
