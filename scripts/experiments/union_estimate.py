#!/usr/bin/env python3
"""What would fewer envs per wavefront buy at small batches?  Same kernel, 4,096 envs, but only `live` of every 16 envs play a
real game (the others hold a lone agent: a trivial tick and a restart every time) — the wavefront then runs the union of the
paths of `live` envs instead of 16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
n = 4096
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
for live in (16, 8, 4, 2, 1):
    b = pa.make_boards(n, seed=3)
    idle = (np.arange(n) % 16) >= live
    for i in (1, 2, 3):
        b["agents"]["dead"][idle, i] = 1
    b["aliveAgents"][idle] = 1
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, stream=stream.cuda_stream, streams=1)
    env.make_game(b)
    env.step_random(1, 1, ticks=300); env.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); env.step_random(1, 1, ticks=400); env.flush(); e1.record(stream); env.sync()
    print(f"{live:2d} real games per wavefront: {e0.elapsed_time(e1) / 400 * 1e3:6.2f} us per step", flush=True)
    env.close()
