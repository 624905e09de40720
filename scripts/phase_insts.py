#!/usr/bin/env python3
"""Diagnostic: dynamic instruction counts per phase of the tick.  A POM_TRUNC build (never shipped: its results are wrong by
design) plays `--burn` ticks whole, then ONE launch of one tick that stops after phase POM_TRUNC_AT; run under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES` once per cut, the last pom_step_kernel dispatch of each
run is that launch, and the difference between consecutive cuts is a phase (scripts/phase_insts.sh drives it).
usage (GPU box): POM_TRUNC_CUT=k python scripts/phase_insts.py [--envs N] [--kind ffa|stress] [--dist 1|2] [--burn 300]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--kind", default="ffa")
ap.add_argument("--dist", type=int, default=1)
ap.add_argument("--burn", type=int, default=300)
ap.add_argument("--build-only", action="store_true")
ap.add_argument("--policy", action="store_true", help="4x SimpleAgent fused with the tick (fresh boards)")
a = ap.parse_args()
lib = os.path.join(ROOT, "build", "libpom_batch_trunc.so")
if a.build_only or not os.path.exists(lib):
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    subprocess.run(["hipcc", "-Os", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DPOM_TRUNC", "-I" + ROOT + "/include",
                    "-I" + ROOT + "/pomcpp_amd/csrc", "-o", lib, ROOT + "/pomcpp_amd/csrc/pom_batch.hip"], check=True)
    if a.build_only:
        sys.exit(0)
import pomcpp_amd.batch as B
B.library_path = lambda: lib
import pomcpp_amd as pa
if a.policy:
    env = B.BatchEnvironment(a.envs, mode=B.MODE_ENV, auto_reset=True, max_steps=800, streams=1, fresh_boards=True, board_seed=1)
    env.generate(1)
    env.step_simple(1, a.burn)
else:
    env = B.BatchEnvironment(a.envs, mode=B.MODE_ENV, auto_reset=True, max_steps=800, streams=1)
    env.make_game(pa.make_boards(a.envs, seed=1000003, kind=a.kind))
    env.step_random(1, a.dist, ticks=a.burn)
env.sync()
os.environ["POM_TRUNC_AT"] = os.environ.get("POM_TRUNC_CUT", "990")
if a.policy:
    env.step_simple(1, 1)
else:
    env.step_random(1, a.dist, ticks=1)
env.sync()
