#!/usr/bin/env python3
"""scripts/observe_only.py [--dtype uint8|codes|float32] [--fused] — 65,536 played envs, 20 observation exports (or fused step + export
launches): the workload for rocprofv3 runs on the observation kernels."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, RESET_AT_END

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--dtype", default="uint8")
ap.add_argument("--fused", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
gen = torch.Generator(device=dev).manual_seed(1)
mv = torch.randint(0, 6, (8, a.envs, 4), dtype=torch.int32, device=dev, generator=gen)
env = BatchEnvironment(a.envs, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=800, stream=stream.cuda_stream)
env.make_game(pa.make_boards(a.envs, seed=1000003))
env.step_random(1, 1, ticks=300)
out, _, _ = env.observe(dtype=a.dtype, attrs=False)
for t in range(20):
    if a.fused:
        env.step_device_observe(mv[t % 8], dtype=a.dtype, out=out, attrs=False)
    else:
        env.observe(dtype=a.dtype, out=out, attrs=False)
env.sync()
env.close()
