#!/usr/bin/env python3
"""Experiment: one handle, streams=K (library-internal split) — timing without torch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
start = pa.make_boards(n, seed=1)
for K in (1, 2, 3, 4, 6, 8):
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=K)
    env.make_game(start)
    env.step_random(1, 1, ticks=40)
    env.sync()
    t0 = time.perf_counter()
    T = 300
    for _ in range(T):
        env.step_random(1, 1, ticks=1)
    env.sync()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    env.step_random(1, 1, ticks=T)   # one API call, T ticks: launches back to back from C
    env.sync()
    dt2 = time.perf_counter() - t1
    print(f"envs {n} streams {K}: {dt / T * 1e6:7.2f} us/tick (one call per tick)   {dt2 / T * 1e6:7.2f} us/tick (one call, {T} ticks)  HWQ={os.environ.get('GPU_MAX_HW_QUEUES','default')}")
    env.close()
