#!/usr/bin/env python3
"""per-tile load (sum over the tile's 16 envs of bombs.count + flames.count) after the bench's burn-in: mean, sd, percentiles"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV
kind = sys.argv[1] if len(sys.argv) > 1 else "ffa"
n = 65536
env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800)
env.make_game(pa.make_boards(n, seed=1000003, kind=kind))
env.step_random(1, 2 if kind == "stress" else 1, ticks=300)
s = env.get_state()
l = np.minimum(s["bombs_count"].astype(np.int64) + s["flames_count"].astype(np.int64), 31).reshape(-1, 16).sum(axis=1)
print(kind, "mean %.1f sd %.1f" % (l.mean(), l.std()), "pct 50/90/99/99.9:", np.percentile(l, [50, 90, 99, 99.9]))
print("BASE=%d STEP=%d" % (round(l.mean() + l.std()), max(1, round(l.std()))))
