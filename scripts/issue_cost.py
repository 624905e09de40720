#!/usr/bin/env python3
"""Host time to QUEUE a 20-tick call (the call returns when everything is queued) and the GPU time it then takes, from an idle
device — the driver's shape.  POM_GRAPH_TICKS=0: direct launches from the calling thread; default: one HIP graph per part."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, stream=stream.cuda_stream)
env.make_game(pa.make_boards(n, seed=3))
env.step_random(1, 1, ticks=300); env.sync()
q, tot = [], []
for rep in range(30):
    env.fork(); torch.cuda.synchronize(); time.sleep(0.002)
    t0 = time.perf_counter()
    env.step_random(1, 1, ticks=K)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    q.append((t1 - t0) * 1e6); tot.append((t2 - t0) * 1e6)
q, tot = np.array(q[5:]), np.array(tot[5:])
print(f"{n} envs, {K} ticks per call, graph chunk {os.environ.get('POM_GRAPH_TICKS', 'default')}: queued in {np.median(q):.0f} us (min {q.min():.0f}), "
      f"done after {np.median(tot):.0f} us (min {tot.min():.0f}) = {np.median(tot) / K:.2f} us per step")
