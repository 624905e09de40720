#!/usr/bin/env python3
"""Tuning sweep: us/tick of pom_step_kernel by envs-per-wavefront (POM_EPW) and batch size; checks results agree."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
for n in (4096, 16384, 32768, 65536, 98304, 131072, 262144):
    start = pa.make_boards(n, seed=1)
    ref = None
    for epw in (64, 32, 16):
        os.environ["POM_EPW"] = str(epw)
        st = torch.cuda.Stream(); torch.cuda.set_stream(st)
        env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, stream=st.cuda_stream)
        env.make_game(start)
        env.step_random(1, 1, ticks=40)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        env.step_random(1, 1, ticks=200)
        e1.record(st); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        d = hashlib.blake2b(env.get_state().tobytes(), digest_size=8).hexdigest()
        ref = ref or d
        print(f"envs {n:7d} EPW {epw:2d}: {us:8.2f} us/tick  {n/us/1e3:7.3f} G env-steps/s  {'same' if d == ref else 'DIFFERENT'}")
        env.close()
