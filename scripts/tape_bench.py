#!/usr/bin/env python3
"""scripts/tape_bench.py [--envs N] [--ticks K] — explicit Move[4] as a K-tick tape (pom_batch_step_device_many, chained launches) at
65,536 envs, POM_RESET_AT_END: microseconds per tick between HIP events.  POM_LIB selects an experimental build."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, RESET_AT_END

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--ticks", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
gen = torch.Generator(device=dev).manual_seed(1)
tape = torch.randint(0, 6, (a.ticks, a.envs, 4), dtype=torch.int32, device=dev, generator=gen)
env = BatchEnvironment(a.envs, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=800, stream=stream.cuda_stream)
env.make_game(pa.make_boards(a.envs, seed=1000003))
env.step_device_many(tape)
env.step_device_many(tape)
env.sync()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = None
for rep in range(3):
    e0.record(stream)
    env.step_device_many(tape)
    env.flush()
    e1.record(stream)
    env.sync()
    us = e0.elapsed_time(e1) / a.ticks * 1e3
    best = us if best is None else min(best, us)
print(os.environ.get("POM_LIB", "default"), "tape: %.3f us per tick (best of 3 x %d)" % (best, a.ticks))
env.close()
