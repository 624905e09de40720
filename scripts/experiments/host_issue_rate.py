#!/usr/bin/env python3
"""Diagnostic: how long does the HOST need to issue a step (3 launches) compared with the GPU's step time?  GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV

for n, streams in ((65536, 3), (65536, 1), (4096, 1)):
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=streams)
    env.make_game(pa.make_boards(n, seed=1))
    env.step_random(1, 1, ticks=100)
    env.sync()
    t0 = time.perf_counter()
    env.step_random(1, 1, ticks=2000)   # one C call: the library's own launch loop
    t1 = time.perf_counter()
    env.sync()
    t2 = time.perf_counter()
    print(f"envs {n} streams {streams}: host issued 2000 steps in {(t1 - t0) * 1e3:.1f} ms ({(t1 - t0) / 2000 * 1e6:.2f} us/step), "
          f"GPU finished {(t2 - t1) * 1e3:.1f} ms later; total {(t2 - t0) / 2000 * 1e6:.2f} us/step")
    env.close()
