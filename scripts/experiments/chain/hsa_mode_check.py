#!/usr/bin/env python3
"""POM_ISSUE_CHAIN_HSA against POM_ISSUE_CHAIN: parity with the oracle, then timing (calls of 20 and 500 ticks from an idle device)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, ISSUE_CHAIN, ISSUE_CHAIN_HSA
from tests.oracle_lib import Oracle
os.environ["POM_CHAIN_VERBOSE"] = "1"
ora = Oracle()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
start = pa.make_boards(N, seed=5, kind="ffa")
for mode, name in ((ISSUE_CHAIN_HSA, "chain-hsa"), (ISSUE_CHAIN, "chain")):
    env = BatchEnvironment(N, mode=MODE_ENV, auto_reset=True, max_steps=800, issue_mode=mode)
    env.make_game(start)
    ref = np.ascontiguousarray(start[:512]).copy(); init = ref.copy(); done = 0
    for ticks in (2, 5, 16, 64, 213):
        env.step_random(101, 1, ticks=ticks)
        ora.run_random(ref, init, ticks, 101, 0, done, 1, 800); done += ticks
        assert env.get_state(0, 512).tobytes() == ref.tobytes(), (name, ticks)
    assert env.counters()[0] == N * done
    print(name, "parity ok over", done, "ticks", env.issue_info(), flush=True)
    for steps in (20, 20, 20, 500, 500):
        env.fork(); env.sync()
        t0 = time.perf_counter()
        env.step_random(101, 1, ticks=steps)
        env.sync()
        print(f"  {name}: {steps} steps {(time.perf_counter() - t0) / steps * 1e6:.2f} us per step", flush=True)
    env.close()
