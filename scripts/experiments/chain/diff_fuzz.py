#!/usr/bin/env python3
"""Differential fuzz of chained launches: random sequences of calls on a chained handle and on a twin that never chains, batch
sizes from 37 to 65,536 envs, for a given number of seconds.  usage: diff_fuzz.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, DIST_RANDOM, DIST_STRESS, ISSUE_CHAIN, ISSUE_THREADS, RESET_AT_END
def same(g, r):
    g, r = g.copy(), r.copy(); g["agents"]["pad"] = 0; r["agents"]["pad"] = 0
    return g.tobytes() == r.tobytes()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
boards = {}
t_end, seqs, calls, fails = time.time() + secs, 0, 0, 0
while time.time() < t_end:
    n = int(rng.choice([37, 1000, 1777, 4096, 16384, 65536], p=[0.15, 0.2, 0.25, 0.2, 0.12, 0.08]))
    if n not in boards:
        boards[n] = pa.make_boards(n, seed=21, kind="ffa")
    start = boards[n]
    auto_reset = [True, RESET_AT_END, True][int(rng.integers(0, 3))]
    fresh = bool(rng.integers(0, 4) == 0)
    kw = dict(mode=MODE_ENV, auto_reset=auto_reset, max_steps=int(rng.choice([60, 300, 800])), fresh_boards=fresh, board_seed=5)
    a = BatchEnvironment(n, issue_mode=ISSUE_CHAIN, streams=int(rng.integers(0, 5)) if rng.integers(0, 2) else 0, **kw)
    b = BatchEnvironment(n, issue_mode=ISSUE_THREADS, streams=1, **kw)
    for e in (a, b):
        e.generate(5) if fresh else e.make_game(start)
    log = []
    ok = True
    for op in range(int(rng.integers(4, 16))):
        kind = int(rng.integers(0, 10)); seed = int(rng.integers(1, 1 << 30)) if rng.integers(0, 3) == 0 else 77; ticks = int(rng.integers(1, 70))
        calls += 1
        if kind <= 3:
            for e in (a, b): e.step_random(seed, DIST_RANDOM if kind < 3 else DIST_STRESS, ticks=ticks)
            log.append(f"random {ticks} seed {seed}")
        elif kind == 4:
            tpl = int(rng.integers(2, 5))
            for e in (a, b): e.step_random(seed, DIST_RANDOM, ticks=tpl * (1 + ticks // 8), ticks_per_launch=tpl)
            log.append(f"random tpl {tpl}")
        elif kind == 5:
            for e in (a, b): e.step_simple(seed, 1 + ticks // 4)
            log.append(f"simple {1 + ticks // 4}")
        elif kind == 6:
            mv = rng.integers(0, 6, size=(n, 4), dtype=np.int32)
            for e in (a, b): e.step(mv)
            log.append("moves")
        elif kind == 7:
            a.set_streams(int(rng.integers(1, 6))); t2 = int(rng.integers(0, 1000))
            for e in (a, b): e.set_tick(t2)
            log.append("streams/tick")
        elif kind == 8 and not fresh:
            if rng.integers(0, 2):
                for e in (a, b): e.snapshot()
                log.append("snapshot")
            else:
                first = int(rng.integers(0, max(1, n - 30))); count = int(rng.integers(1, min(200, n - first) + 1))
                for e in (a, b): e.make_game(np.ascontiguousarray(start[first:first + count]), first=first)
                log.append(f"upload {first}+{count}")
        else:
            sa, sb = a.status(), b.status()
            ok = ok and all(np.array_equal(sa[k], sb[k]) for k in sa)
            log.append("status")
        if rng.integers(0, 5) == 0:
            ok = ok and same(a.get_state(), b.get_state())
            log.append("compare")
        if not ok:
            break
    ok = ok and same(a.get_state(), b.get_state()) and np.array_equal(a.counters(), b.counters()) and np.array_equal(a.policy_memory(), b.policy_memory()) \
        and np.array_equal(a.episodes(), b.episodes())
    seqs += 1
    if not ok:
        fails += 1
        print("MISMATCH n", n, "auto_reset", auto_reset, "fresh", fresh, log, flush=True)
    a.close(); b.close()
print("sequences", seqs, "calls", calls, "failures", fails)
