#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime shares of pom_policy_kernel (POM_DIAG build, never shipped; its run time is not quoted).
usage (on the GPU box): python scripts/policy_stamps.py [--envs N] [--ticks T]"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--ticks", type=int, default=200)
ap.add_argument("--warm", type=int, default=100)
a = ap.parse_args()
lib = os.path.join(ROOT, "build", "libpom_batch_diag.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.run(["hipcc", "-Os", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DPOM_DIAG", "-I" + ROOT + "/include",
                "-I" + ROOT + "/pomcpp_amd/csrc", "-o", lib, ROOT + "/pomcpp_amd/csrc/pom_batch.hip"], check=True)
import pomcpp_amd.batch as B
B.library_path = lambda: lib
import pomcpp_amd as pa
env = B.BatchEnvironment(a.envs, mode=B.MODE_ENV, auto_reset=True, max_steps=800, streams=1)
env.make_game(pa.make_boards(a.envs, seed=1, kind="ffa"))
env.step_simple(1, ticks=a.warm)
out = (C.c_longlong * 7)()
L = B.load_library()
L.pom_diag_policy_read.argtypes = [C.c_void_p, C.c_void_p]
L.pom_diag_policy_read(env._h, out)
env.step_simple(1, ticks=a.ticks)
L.pom_diag_policy_read(env._h, out)
names = ["load", "prepare (danger map, sets)", "predicates", "target (reach + window)", "path (backward fill)", "tail (safe step, memory)", "store"]
v = np.array(list(out), dtype=np.float64)
waves = (a.envs + 15) // 16
per = v / (waves * a.ticks)
print(f"envs {a.envs}: s_memtime ticks per wavefront-act (100 MHz), total {per.sum():.0f}")
for n, x in zip(names, per):
    print(f"  {n:28s} {x:10.1f}  {100 * x / per.sum():5.1f} %")
