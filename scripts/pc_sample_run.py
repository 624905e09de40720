#!/usr/bin/env python3
"""Diagnostic: a plain run of the tick for rocprofv3's PC sampling (scripts/pc_sample.sh): `--envs N --kind ffa|stress --dist D --ticks K`,
one launch per tick on one stream, nothing else on the device."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--kind", default="ffa")
ap.add_argument("--dist", type=int, default=1)
ap.add_argument("--ticks", type=int, default=2000)
ap.add_argument("--policy", action="store_true")
a = ap.parse_args()
import pomcpp_amd.batch as B
import pomcpp_amd as pa
if a.policy:
    env = B.BatchEnvironment(a.envs, mode=B.MODE_ENV, auto_reset=True, max_steps=800, streams=1, fresh_boards=True, board_seed=1)
    env.generate(1)
    env.step_simple(1, a.ticks)
else:
    env = B.BatchEnvironment(a.envs, mode=B.MODE_ENV, auto_reset=True, max_steps=800, streams=1)
    env.make_game(pa.make_boards(a.envs, seed=1000003, kind=a.kind))
    env.step_random(1, a.dist, ticks=a.ticks)
env.sync()
print("steps", env.counters()[0])
