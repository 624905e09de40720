#!/usr/bin/env python3
"""explicit-moves path (pom_batch_step_device, joined with the caller's stream every tick) by sub-batches per step"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
start = pa.make_boards(n, seed=1000003)
mv = torch.randint(0, 6, (8, n, 4), dtype=torch.int32, device=dev)
for parts in (1, 2, 3):
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, stream=stream.cuda_stream, streams=parts)
    env.make_game(start)
    for t in range(40):
        env.step_device(mv[t % 8].data_ptr())
    env.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for t in range(300):
        env.step_device(mv[t % 8].data_ptr())
    e1.record(stream)
    env.sync()
    print(f"envs {n} parts {parts}: {e0.elapsed_time(e1) / 300 * 1e3:.2f} us per explicit-moves step", flush=True)
    env.close()
