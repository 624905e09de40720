#!/usr/bin/env python3
"""Time pom_batch_observe (row f4) on mid-game states: HIP events around repeated calls, bytes moved over time.
usage (on the GPU box): python scripts/observe_bench.py [--envs N]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
env = BatchEnvironment(a.envs, mode=MODE_ENV, auto_reset=True, max_steps=800, stream=stream.cuda_stream)
env.make_game(pa.make_boards(a.envs, seed=1))
env.step_simple(1, 150)
rec = 448 * a.envs
for per_agent in (False, True):
    for dtype, esz in (("uint8", 1), ("float16", 2), ("float32", 4)):
        out, _, _ = env.observe(per_agent=per_agent, dtype=dtype)
        for _ in range(5):
            env.observe(per_agent=per_agent, dtype=dtype, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(a.reps):
            env.observe(per_agent=per_agent, dtype=dtype, out=out)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        nbytes = out.numel() * esz + rec + a.envs * (128 + 16)
        print(f"envs {a.envs} per_agent {int(per_agent)} {dtype:8s}: {ms * 1e3:8.1f} us/call  {nbytes / ms / 1e6:8.1f} GB/s "
              f"({nbytes / 1e6:.0f} MB moved, {a.envs / ms / 1e3:.1f} M env-obs/s)", flush=True)
        del out
