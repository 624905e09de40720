#!/usr/bin/env python3
"""One fresh process: PRE steps (one call), sync, optional idle, 5 warm-up steps, then a timed 20-step burst (bench.py's driver shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
pre, idle = int(sys.argv[1]), float(sys.argv[2])
n = 65536
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, stream=stream.cuda_stream, streams=3)
env.make_game(pa.make_boards(n, seed=1000003))
if pre:
    env.step_random(1, 1, ticks=pre)
    env.sync()
if idle:
    time.sleep(idle)
env.step_random(1, 1, ticks=5)
res = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(stream)
    env.step_random(1, 1, ticks=20)
    env.flush()
    e1.record(stream)
    torch.cuda.synchronize()
    res.append("wall %.2f ev %.2f" % ((time.perf_counter() - t0) / 20 * 1e6, e0.elapsed_time(e1) / 20 * 1e3))
print(f"pre {pre:5d} idle {idle:4.2f}: 20-step bursts: " + " | ".join(res), flush=True)
