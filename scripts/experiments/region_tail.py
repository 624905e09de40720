#!/usr/bin/env python3
"""Diagnostic: where a driver-shaped timed region (20 steps between two device synchronisations) spends its time:
the steps themselves, the counter read-back that joins the sub-streams, the final synchronisation.
usage (GPU box): python scripts/region_tail.py [--steps 20] [--reps 15]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--reps", type=int, default=15)
ap.add_argument("--envs", type=int, default=65536)
a = ap.parse_args()
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
env = BatchEnvironment(a.envs, device=0, mode=MODE_ENV, auto_reset=True, max_steps=800, stream=stream.cuda_stream)
env.make_game(pa.make_boards(a.envs, seed=1000003, kind="ffa"))
counters = torch.zeros(4, dtype=torch.int64, device=dev)
env.step_random(1, 1, ticks=300)
env.sync()
rows = {"steps+sync": [], "steps+counters+sync": [], "steps only (host returns)": [], "counters (host returns)": []}
for rep in range(a.reps):
    for mode in ("steps+sync", "steps+counters+sync"):
        env.step_random(1, 1, ticks=5)
        env.counters_into(counters.data_ptr())
        env.fork()
        torch.cuda.synchronize()
        time.sleep(0.05)  # the driver's region starts from an idle device
        t0 = time.perf_counter()
        env.step_random(1, 1, ticks=a.steps)
        t1 = time.perf_counter()
        if mode == "steps+counters+sync":
            env.counters_into(counters.data_ptr())
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        rows[mode].append((t3 - t0) * 1e6)
        if mode == "steps+counters+sync":
            rows["steps only (host returns)"].append((t1 - t0) * 1e6)
            rows["counters (host returns)"].append((t2 - t1) * 1e6)
for k, v in rows.items():
    v = np.array(v)
    print(f"{k:28s} median {np.median(v):8.1f} us  min {v.min():8.1f}  max {v.max():8.1f}   per step {np.median(v) / a.steps:6.2f} us")
