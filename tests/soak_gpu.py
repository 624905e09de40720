#!/usr/bin/env python3
"""Long GPU soak (not collected by pytest; run by hand on the GPU box): 65,536 envs for tens of thousands of ticks, slices
compared bit for bit with the oracle.  usage: python tests/soak_gpu.py [--ticks 20000] [--policy-ticks 3000]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
from tests.oracle_lib import Oracle

ap = argparse.ArgumentParser()
ap.add_argument("--ticks", type=int, default=20000)
ap.add_argument("--policy-ticks", type=int, default=3000)
a = ap.parse_args()
ora = Oracle()
N, cap = 65536, 800
slices = [(0, 384), (21845 - 100, 384), (43690 - 7, 384), (N - 384, 384)]  # around the sub-batch boundaries of a 3-way split
t_all = time.time()
for name, fresh, kind, dist in (("replay, ffa, random moves", False, "ffa", 1), ("fresh boards, ffa, random moves", True, "ffa", 1),
                                ("replay, stress boards, stress moves", False, "stress", 2)):
    seed, bseed = 101, 202
    env = BatchEnvironment(N, mode=MODE_ENV, auto_reset=True, max_steps=cap, fresh_boards=fresh, board_seed=bseed, streams=3)
    if fresh:
        env.generate(bseed)
    else:
        start = pa.make_boards(N, seed=5, kind=kind)
        env.make_game(start)
    t0 = time.time()
    env.step_random(seed, dist, ticks=a.ticks)
    got = env.get_state()
    eps = env.episodes()
    ub = env.status()["ubflags"]
    gpu_s = time.time() - t0
    for first, m in slices:
        if fresh:
            ref = ora.boardgen(bseed, first + np.arange(m), np.zeros(m))
            e = np.zeros(m, dtype=np.int32)
            ora.run_random_fresh(ref, e, a.ticks, seed, bseed, first, 0, dist, cap)
            assert np.array_equal(eps[first:first + m], e), (name, first)
        else:
            init = np.ascontiguousarray(start[first:first + m])
            ref = init.copy()
            ora.run_random(ref, init, a.ticks, seed, first, 0, dist, cap)
        assert got[first:first + m].tobytes() == ref.tobytes(), (name, first)
    cnt = env.counters()
    assert cnt[0] == N * a.ticks
    print(f"{name}: {N} envs x {a.ticks} ticks in {gpu_s:.1f} s, {int(cnt[1])} episodes, {int(cnt[3])} ticks with UB flags "
          f"({int((ub != 0).sum())} envs flagged now); 4 slices of 384 envs = oracle", flush=True)
    env.close()
# SimpleAgent games with fresh boards
seed, bseed = 7, 9
env = BatchEnvironment(N, mode=MODE_ENV, auto_reset=True, max_steps=cap, fresh_boards=True, board_seed=bseed)
env.generate(bseed)
t0 = time.time()
env.step_simple(seed, a.policy_ticks)
got, eps = env.get_state(), env.episodes()
gpu_s = time.time() - t0
for first, m in [(0, 192), (30000, 192), (N - 192, 192)]:
    ref = ora.boardgen(bseed, first + np.arange(m), np.zeros(m))
    e, mems = np.zeros(m, dtype=np.int32), np.zeros((m, 4, 16), dtype=np.int32)
    ora.run_simple_fresh(ref, e, mems, a.policy_ticks, seed, bseed, first, 0, cap)
    assert got[first:first + m].tobytes() == ref.tobytes() and np.array_equal(eps[first:first + m], e), first
    assert np.array_equal(env.policy_memory(first, m), mems), first
print(f"SimpleAgent x4, fresh boards: {N} envs x {a.policy_ticks} ticks in {gpu_s:.1f} s, {int(env.counters()[1])} episodes; "
      f"3 slices of 192 envs (states, episode counts, agent memory) = oracle", flush=True)
env.close()
print(f"soak ok in {time.time() - t_all:.0f} s")
