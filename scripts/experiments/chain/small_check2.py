#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, ISSUE_CHAIN, PomError
from tests.oracle_lib import Oracle
ora = Oracle()
n = 4000
start = pa.make_boards(n, seed=12)
for streams in (2, 3):
    for plan in (((47, 1), (5, 1), (63, 3)), ((47, 1), (63, 3)), ((5, 1), (63, 3)), ((47, 1), (5, 1)), ((5, 1), (6, 3), (5, 1)), ((6, 3), (5, 1))):
        ref = start.copy()
        env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=streams, issue_mode=ISSUE_CHAIN)
        env.make_game(start)
        done = 0
        for ticks, tpl in plan:
            env.step_random(99, 1, ticks=ticks, ticks_per_launch=tpl)
            ora.run_random(ref, start, ticks, 99, 0, done, 1, 800)
            done += ticks
        try:
            got = env.get_state()
            bad = np.nonzero([got[i].tobytes() != ref[i].tobytes() for i in range(n)])[0]
            print(streams, plan, len(bad), sorted(set((bad // 16).tolist()))[:12], flush=True)
        except PomError as e:
            print(streams, plan, str(e)[:100])
        env.close()
