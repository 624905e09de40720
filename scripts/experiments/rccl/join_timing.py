#!/usr/bin/env python3
"""host time of flush / counters_into right behind a 20-step chained call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
n = 65536
env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, stream=stream.cuda_stream)
env.make_game(pa.make_boards(n, seed=3))
c = torch.zeros(4, dtype=torch.int64, device=dev)
side = torch.cuda.Stream(device=dev)
env.step_random(1, 1, ticks=300); env.sync()
ev = torch.cuda.Event()
ev.record(stream)
for what in ("flush", "counters_into", "side.wait_stream", "side.wait_stream", "event made before", "event made before", "new event kept", "nothing"):
    env.fork()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.step_random(1, 1, ticks=20)
    t1 = time.perf_counter()
    if what == "flush": env.flush()
    elif what == "counters_into": env.counters_into(c.data_ptr())
    elif what == "side.wait_stream": side.wait_stream(stream)
    elif what == "event made before":
        ev.record(stream); side.wait_event(ev)
    elif what == "new event kept":
        keep = torch.cuda.Event(); keep.record(stream); side.wait_event(keep)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"{what:18s}: steps queued {1e6*(t1-t0):6.0f} us, call {1e6*(t2-t1):6.0f} us, sync {1e6*(t3-t2):6.0f} us, total {1e6*(t3-t0):6.0f} us")
env.close()
