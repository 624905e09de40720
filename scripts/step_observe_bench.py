#!/usr/bin/env python3
"""scripts/step_observe_bench.py [--envs N] — an RL tick at 65,536 envs: explicit moves in, uint8 global planes out, as two launches
(pom_batch_step_device + pom_batch_observe) and as one (pom_batch_step_device_observe).  POM_LIB selects an experimental build."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, RESET_AT_END

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
gen = torch.Generator(device=dev).manual_seed(1)
mv = torch.randint(0, 6, (8, a.envs, 4), dtype=torch.int32, device=dev, generator=gen)
env = BatchEnvironment(a.envs, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=800, stream=stream.cuda_stream)
env.make_game(pa.make_boards(a.envs, seed=1000003))
for t in range(300):
    env.step_device(mv[t % 8].data_ptr())
planes, _, _ = env.observe(attrs=False)
codes, _, _ = env.observe(dtype="codes", attrs=False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {}
for name in ("step only", "two launches", "one launch", "codes: two launches", "codes: one launch", "observe only", "codes: observe only"):
    for rep in range(2):
        e0.record(stream)
        for t in range(a.steps):
            if name == "one launch":
                env.step_device_observe(mv[t % 8], out=planes, attrs=False)
            elif name == "codes: one launch":
                env.step_device_observe(mv[t % 8], dtype="codes", out=codes, attrs=False)
            elif name == "observe only":
                env.observe(out=planes, attrs=False)
            elif name == "codes: observe only":
                env.observe(dtype="codes", out=codes, attrs=False)
            elif name == "codes: two launches":
                env.step_device(mv[t % 8].data_ptr())
                env.observe(dtype="codes", out=codes, attrs=False)
            else:
                env.step_device(mv[t % 8].data_ptr())
                if name == "two launches":
                    env.observe(out=planes, attrs=False)
        e1.record(stream)
        env.sync()
    out[name] = e0.elapsed_time(e1) / a.steps * 1e3
print(os.environ.get("POM_LIB", "default"), " ".join(f"{k}: {v:.2f} us" for k, v in out.items()))
env.close()
