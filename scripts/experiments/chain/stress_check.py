#!/usr/bin/env python3
"""Hunt for a rare mismatch of chained launches: many fresh handles, several stream counts, other handles alive beside them."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, ISSUE_CHAIN, ISSUE_THREADS, ISSUE_GRAPH, PomError
from tests.oracle_lib import Oracle
ora = Oracle()
n = 4000
start = pa.make_boards(n, seed=12)
plans = {}
def want(plan):
    key = tuple(plan)
    if key not in plans:
        ref = start.copy(); done = 0
        for ticks, tpl in plan:
            ora.run_random(ref, start, ticks, 99, 0, done, 1, 800); done += ticks
        plans[key] = ref
    return plans[key]
others = []
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 60
it = 0; fails = 0
while time.time() < t_end:
    it += 1
    streams = (2, 3, 3, 4)[it % 4]
    plan = (((47, 1), (5, 1), (63, 3)), ((47, 1), (5, 1)), ((30, 1),))[it % 3]
    if it % 5 == 0:  # keep some other handles (other streams, helper threads, graphs) alive beside the one under test
        o = BatchEnvironment(4000, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=3, issue_mode=(ISSUE_THREADS, ISSUE_GRAPH)[it % 2])
        o.make_game(start); o.step_random(1, 1, ticks=25); others.append(o)
        if len(others) > 3: others.pop(0).close()
    env = BatchEnvironment(n, mode=MODE_ENV, auto_reset=True, max_steps=800, streams=streams, issue_mode=ISSUE_CHAIN)
    env.make_game(start)
    for ticks, tpl in plan:
        env.step_random(99, 1, ticks=ticks, ticks_per_launch=tpl)
    try:
        got = env.get_state()
        ref = want(plan)
        bad = np.nonzero([got[i].tobytes() != ref[i].tobytes() for i in range(n)])[0]
        if len(bad):
            fails += 1
            print("MISMATCH it", it, "streams", streams, plan, len(bad), "envs in tiles", sorted(set((bad // 16).tolist()))[:20], flush=True)
    except PomError as e:
        fails += 1
        print("ERROR it", it, streams, plan, str(e)[:100], flush=True)
    env.close()
print("iterations", it, "failures", fails)
