#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, ISSUE_CHAIN, PomError
from tests.oracle_lib import Oracle
ora = Oracle()
for n in (4000, 4096, 250, 16 * 8 * 5):
    for streams in (2, 3, 4):
        start = pa.make_boards(n, seed=12)
        ref = start.copy()
        env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=streams, issue_mode=ISSUE_CHAIN)
        env.make_game(start)
        done = 0
        res = []
        try:
            for ticks, tpl in ((47, 1), (5, 1), (1, 1), (63, 3), (20, 1)):
                env.step_random(99, 1, ticks=ticks, ticks_per_launch=tpl)
                ora.run_random(ref, start, ticks, 99, 0, done, 1, 800)
                done += ticks
                got = env.get_state()
                bad = np.nonzero([got[i].tobytes() != ref[i].tobytes() for i in range(n)])[0]
                res.append((ticks, tpl, len(bad), (bad[:6] // 16).tolist()))
        except PomError as e:
            res.append(str(e)[:80])
        print(n, streams, res, flush=True)
        env.close()
