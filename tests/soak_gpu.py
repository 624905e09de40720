#!/usr/bin/env python3
"""Long GPU soak (not collected by pytest; run by hand on the GPU box): 65,536 envs for tens of thousands of ticks, slices
compared bit for bit with the oracle every few ticks (a difference must not get the chance to be wiped out by a restart).  usage: python tests/soak_gpu.py [--ticks 20000] [--policy-ticks 3000]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, RESET_AT_END
from tests.oracle_lib import Oracle

ap = argparse.ArgumentParser()
ap.add_argument("--ticks", type=int, default=20000)
ap.add_argument("--policy-ticks", type=int, default=3000)
ap.add_argument("--chunk", type=int, default=16, help="ticks between comparisons (random-move configurations)")
ap.add_argument("--policy-chunk", type=int, default=20)
ap.add_argument("--seed", type=int, default=101, help="move-stream seed (board seeds follow from it)")
a = ap.parse_args()
ora = Oracle()
N, cap = 65536, 800
slices = [(0, 384), (21845 - 100, 384), (43690 - 7, 384), (N - 384, 384)]  # around the sub-batch boundaries of a 3-way split
t_all = time.time()
for name, fresh, kind, dist, at_end in (("replay, ffa, random moves", False, "ffa", 1, False),
                                        ("fresh boards, ffa, random moves", True, "ffa", 1, False),
                                        ("replay, stress boards, stress moves", False, "stress", 2, False),
                                        ("replay, reset at the END of the tick, ffa, random moves", False, "ffa", 1, True),
                                        ("replay, reset at the END of the tick, stress", False, "stress", 2, True)):
    seed, bseed = a.seed, 2 * a.seed
    env = BatchEnvironment(N, mode=MODE_ENV, auto_reset=RESET_AT_END if at_end else True, max_steps=cap, fresh_boards=fresh,
                           board_seed=bseed, streams=(2, 3)[len(name) % 2])
    if fresh:
        env.generate(bseed)
    else:
        start = pa.make_boards(N, seed=5, kind=kind)
        env.make_game(start)
    # compare every `chunk` ticks — shorter than a game, so that a difference cannot be wiped out by the restart that follows it
    chunk, t0, gpu_s = a.chunk, time.time(), 0.0
    refs = []
    for first, m in slices:
        if fresh:
            refs.append([ora.boardgen(bseed, first + np.arange(m), np.zeros(m)), np.zeros(m, dtype=np.int32), None])
        else:
            init = np.ascontiguousarray(start[first:first + m])
            refs.append([init.copy(), None, init])
    for tick in range(0, a.ticks, chunk):
        t1 = time.time()
        env.step_random(seed, dist, ticks=chunk - chunk // 3)  # two calls back to back: with chained launches their launches
        env.step_random(seed, dist, ticks=chunk // 3)          # are in flight together
        got = [env.get_state(first, m) for first, m in slices]
        gpu_s += time.time() - t1
        for (first, m), r, g in zip(slices, refs, got):
            if fresh:
                ora.run_random_fresh(r[0], r[1], chunk, seed, bseed, first, tick, dist, cap)
            else:
                ora.run_random(r[0], r[2], chunk, seed, first, tick, dist, cap)
            if at_end:  # the oracle restarts a finished env at the start of the next tick, the device already has: same games
                fin = (r[0]["aliveAgents"] <= 1) | (r[0]["timeStep"] >= cap)
                r[0][fin] = r[2][fin]
            assert g.tobytes() == r[0].tobytes(), (name, first, tick)
    eps = env.episodes()
    ub = env.status()["ubflags"]
    if fresh:
        for (first, m), r in zip(slices, refs):
            assert np.array_equal(eps[first:first + m], r[1]), (name, first)
    cnt = env.counters()
    assert cnt[0] == N * a.ticks
    print(f"{name}: {N} envs x {a.ticks} ticks in {gpu_s:.1f} s, {int(cnt[1])} episodes, {int(cnt[3])} ticks with UB flags "
          f"({int((ub != 0).sum())} envs flagged now); 4 slices of 384 envs = oracle every {a.chunk} ticks", flush=True)
    env.close()
# SimpleAgent games with fresh boards
seed, bseed = (7, 9) if a.seed == 101 else (a.seed + 6, a.seed + 8)
env = BatchEnvironment(N, mode=MODE_ENV, auto_reset=True, max_steps=cap, fresh_boards=True, board_seed=bseed)
env.generate(bseed)
pslices = [(0, 192), (30000, 192), (N - 192, 192)]
prefs = [[ora.boardgen(bseed, first + np.arange(m), np.zeros(m)), np.zeros(m, dtype=np.int32), np.zeros((m, 4, 16), dtype=np.int32)]
         for first, m in pslices]
gpu_s, chunk = 0.0, a.policy_chunk
for tick in range(0, a.policy_ticks, chunk):
    t1 = time.time()
    env.step_simple(seed, chunk - chunk // 3)
    env.step_simple(seed, chunk // 3)
    got = [(env.get_state(first, m), env.policy_memory(first, m)) for first, m in pslices]
    gpu_s += time.time() - t1
    for (first, m), r, (g, gm) in zip(pslices, prefs, got):
        ora.run_simple_fresh(r[0], r[1], r[2], chunk, seed, bseed, first, tick, cap)
        assert g.tobytes() == r[0].tobytes() and np.array_equal(gm, r[2]), (first, tick)
eps = env.episodes()
for (first, m), r in zip(pslices, prefs):
    assert np.array_equal(eps[first:first + m], r[1]), first
print(f"SimpleAgent x4, fresh boards: {N} envs x {a.policy_ticks} ticks in {gpu_s:.1f} s, {int(env.counters()[1])} episodes; "
      f"3 slices of 192 envs (states, agent memory every {a.policy_chunk} ticks, episode counts) = oracle", flush=True)
env.close()
print(f"soak ok in {time.time() - t_all:.0f} s")
