#!/usr/bin/env python3
"""Step time as the batch ages: fresh games at tick 0 -> steady mix (episodes of every length up to the 800-tick cap)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kind = sys.argv[3] if len(sys.argv) > 3 else "ffa"
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, stream=stream.cuda_stream, streams=3)
env.make_game(pa.make_boards(n, seed=1000003, kind=kind))
chunk = 100
for c in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    env.step_random(1, dist, ticks=chunk)
    env.flush()
    e1.record(stream)
    env.sync()
    line = f"ticks {c * chunk:5d}..{(c + 1) * chunk:5d}: {e0.elapsed_time(e1) / chunk * 1e3:6.2f} us/step"
    if c % 4 == 3 or c < 4:
        st = env.get_state()
        ts = st["timeStep"]
        line += (f"   mean timeStep {ts.mean():6.1f}  bombs {st['bombs_count'].mean():5.2f}  flames {st['flames_count'].mean():5.2f}  "
                 f"alive {st['aliveAgents'].mean():4.2f}  strength {st['agents']['bombStrength'].mean():4.2f}  timeStep>100: {(ts > 100).mean() * 100:4.1f} %")
    print(line, flush=True)
