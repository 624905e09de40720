#!/usr/bin/env python3
"""How does the time of a 20-step burst depend on what the GPU did just before (clock / power state)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV

n = 65536
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
start = pa.make_boards(n, seed=1000003)
env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, stream=stream.cuda_stream, streams=3)
env.make_game(start)
env.step_random(1, 1, ticks=5)
env.sync()

def burst(k=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(stream)
    env.step_random(1, 1, ticks=k)
    env.flush()
    e1.record(stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6, e0.elapsed_time(e1) / k * 1e3

for pre, sleep in ((0, 0), (0, 0), (100, 0), (100, 0.02), (1000, 0), (1000, 0.02), (1000, 0.2), (5000, 0), (5000, 0.2), (0, 0)):
    if pre:
        env.step_random(1, 1, ticks=pre)
        env.sync()
    if sleep:
        time.sleep(sleep)
    w, g = burst()
    print(f"after {pre:5d} steps + {sleep:4.2f} s idle: 20-step burst wall {w:6.2f} us/step, events {g:6.2f} us/step", flush=True)
w, g = burst(400)
print(f"400-step run: wall {w:6.2f} events {g:6.2f}")
for k in (1, 2):
    env.set_streams(k); env.step_random(1, 1, ticks=50); env.sync()
env.set_streams(3)
for i in range(3):
    w, g = burst()
    print(f"after set_streams 1,2,3 cycle: burst wall {w:6.2f} events {g:6.2f}")
