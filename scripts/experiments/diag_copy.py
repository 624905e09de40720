#!/usr/bin/env python3
"""Diagnostic: time pom_step_kernel with ticks=0 (pure record round trip) vs ticks=1, HIP events via torch."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "build", "libpom_batch_diag.so")
subprocess.run(["hipcc", "-Os", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DPOM_DIAG", "-I" + ROOT + "/include",
                "-I" + ROOT + "/pomcpp_amd/csrc", "-o", lib, ROOT + "/pomcpp_amd/csrc/pom_batch.hip"], check=True)
import torch
import pomcpp_amd.batch as B
B.library_path = lambda: lib
import pomcpp_amd as pa
for n in ([int(x) for x in sys.argv[1:]] or [65536, 262144]):
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    env = B.BatchEnvironment(n, mode=B.MODE_ENV, auto_reset=True, max_steps=800, stream=st.cuda_stream)
    env.make_game(pa.make_boards(n, seed=1))
    L = B.load_library(); L.pom_diag_copy_only.argtypes = [C.c_void_p]
    env.step_random(1, 1, ticks=30)
    for name, fn in (("ticks=0 copy only", lambda: L.pom_diag_copy_only(env._h)), ("ticks=1", lambda: env.step_random(1, 1, ticks=1))):
        for _ in range(20): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(200): fn()
        e1.record(st); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        print(f"envs {n}: {name:20s} {us:8.2f} us/launch   record bytes moved {n*448*2/1e6:.1f} MB -> {n*448*2/us/1e6:.2f} TB/s")
    env.close()
