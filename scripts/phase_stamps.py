#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime shares of pom_step_kernel (POM_DIAG build, never shipped; its run time is not quoted).
usage (on the GPU box): python scripts/phase_stamps.py [--envs N] [--kind ffa|stress] [--dist 0|1|2] [--ticks T]"""
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
ap.add_argument("--kind", default="ffa")
ap.add_argument("--dist", type=int, default=1)
ap.add_argument("--ticks", type=int, default=200)
a = ap.parse_args()
lib = os.path.join(ROOT, "build", "libpom_batch_diag.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.run(["hipcc", "-Os", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DPOM_DIAG", "-I" + ROOT + "/include",
                "-I" + ROOT + "/pomcpp_amd/csrc", "-o", lib, ROOT + "/pomcpp_amd/csrc/pom_batch.hip"], check=True)
import pomcpp_amd.batch as B
B.library_path = lambda: lib
import pomcpp_amd as pa
env = B.BatchEnvironment(a.envs, mode=B.MODE_ENV, auto_reset=True, max_steps=800, streams=1)
env.make_game(pa.make_boards(a.envs, seed=1, kind=a.kind))
env.step_random(1, a.dist, ticks=50)
out = (C.c_longlong * 17)()
L = B.load_library()
L.pom_diag_read.argtypes = [C.c_void_p, C.c_void_p]
L.pom_diag_read(env._h, out)
env.step_random(1, a.dist, ticks=a.ticks)
L.pom_diag_read(env._h, out)
names = ["load", "tick_flames", "agent_prep", "agent_loop", "bomb reset/classify pass", "bomb_loop_A", "bomb_loop_B", "tick_bombs+explosions", "epilogue", "store",
         "  long blast: look", "  long blast: commit", "  long blast: bookkeeping", "  long blast: nest", "  short blast", "restarts + move draw (before the tick)", "flames: timers (the rest of tick_flames = pops)"]
v = np.array(list(out), dtype=np.float64)
epw = env.launch_shape()[0]; waves = (a.envs + epw - 1) // epw
per = v / (waves * a.ticks)
print(f"envs {a.envs} kind {a.kind} dist {a.dist}: s_memtime ticks per wavefront-tick (100 MHz clock ticks if constant clock), total {per.sum():.0f}")
for n, x in zip(names, per):
    print(f"  {n:24s} {x:10.1f}  {100 * x / per.sum():5.1f} %")
