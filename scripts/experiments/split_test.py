#!/usr/bin/env python3
"""Experiment: one batch vs K independent sub-batches on K streams (memory phases of one overlap compute of the others)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
start = pa.make_boards(n, seed=1)
for K in (1, 2, 4, 8):
    for epw in (32, 64):
        q = n // K
        envs = [BatchEnvironment(q, mode=MODE_ENV, auto_reset=True, max_steps=800, env_offset=k * q, envs_per_wave=epw) for k in range(K)]
        for k, e in enumerate(envs):
            e.make_game(start[k * q:(k + 1) * q])
        for _ in range(40):
            for e in envs:
                e.step_random(1, 1, ticks=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T = 300
        for _ in range(T):
            for e in envs:
                e.step_random(1, 1, ticks=1)
        for e in envs:
            e.sync()
        dt = time.perf_counter() - t0
        print(f"envs {n} split {K} x {q} EPW {epw}: {dt / T * 1e6:7.2f} us/tick  {n * T / dt / 1e9:6.3f} G env-steps/s")
        for e in envs:
            e.close()
