#!/usr/bin/env python3
"""Chained launches: two against three streams by the length of a call (each call from an idle, synchronised device)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, ISSUE_CHAIN
N = 65536
start = pa.make_boards(N, seed=5, kind="ffa")
envs = {}
for s in (2, 3):
    e = BatchEnvironment(N, mode=MODE_ENV, auto_reset=True, max_steps=800, issue_mode=ISSUE_CHAIN, streams=s)
    e.make_game(start)
    e.step_random(101, 1, ticks=300)
    e.sync()
    envs[s] = e
for ticks in (10, 20, 30, 40, 60, 80, 120, 200, 400):
    out = []
    for s in (2, 3):
        e = envs[s]
        ts = []
        for rep in range(9):
            e.fork()
            e.sync()
            t0 = time.perf_counter()
            e.step_random(101, 1, ticks=ticks)
            e.sync()
            ts.append((time.perf_counter() - t0) / ticks * 1e6)
        out.append(f"{s} streams: median {np.median(ts):.2f} min {min(ts):.2f} max {max(ts):.2f}")
    print(f"{ticks:4d} ticks per call   " + "   ".join(out), flush=True)
