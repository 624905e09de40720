#!/usr/bin/env python3
"""Replay of tests/test_gpu_chain.py::test_random_api_sequences_chained_against_plain with a comparison after every call."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, DIST_RANDOM, DIST_STRESS, ISSUE_CHAIN, ISSUE_THREADS, RESET_AT_END
def same(g, r):
    g, r = g.copy(), r.copy(); g["agents"]["pad"] = 0; r["agents"]["pad"] = 0
    bad = np.nonzero([g[i].tobytes() != r[i].tobytes() for i in range(len(g))])[0]
    return bad
n = 1777
start = pa.make_boards(n, seed=21)
auto_reset, fresh = True, False
for seq in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    rng = np.random.default_rng(1000 * seq + (7 if fresh else 0) + int(auto_reset))
    kw = dict(mode=MODE_ENV, auto_reset=auto_reset, max_steps=300, fresh_boards=fresh, board_seed=5)
    sa = int(rng.integers(2, 5))
    a = BatchEnvironment(n, issue_mode=ISSUE_CHAIN, streams=sa, **kw); b = BatchEnvironment(n, issue_mode=ISSUE_THREADS, streams=1, **kw)
    for e in (a, b): e.make_game(start)
    print("seq", seq, "streams", sa)
    for op in range(14):
        kind = int(rng.integers(0, 10)); seed, ticks = int(rng.integers(1, 1 << 30)), int(rng.integers(1, 50))
        desc = ""
        if kind <= 2:
            for e in (a, b): e.step_random(seed, DIST_RANDOM, ticks=ticks)
            desc = f"step_random {ticks}"
        elif kind == 3:
            tpl = int(rng.integers(2, 5))
            for e in (a, b): e.step_random(seed, DIST_STRESS, ticks=tpl * (1 + ticks // 8), ticks_per_launch=tpl)
            desc = f"step_random stress {tpl * (1 + ticks // 8)} tpl {tpl}"
        elif kind == 4:
            for e in (a, b): e.step_simple(seed, 1 + ticks // 4)
            desc = f"step_simple {1 + ticks // 4}"
        elif kind == 5:
            mv = rng.integers(0, 6, size=(n, 4), dtype=np.int32)
            for e in (a, b): e.step(mv)
            desc = "step(moves)"
        elif kind == 6:
            s2 = int(rng.integers(1, 6)); a.set_streams(s2); t2 = int(rng.integers(0, 1000))
            for e in (a, b): e.set_tick(t2)
            desc = f"set_streams {s2} set_tick {t2}"
        elif kind == 7:
            for e in (a, b): e.snapshot()
            desc = "snapshot"
        elif kind == 8:
            first, count = int(rng.integers(0, n - 200)), int(rng.integers(1, 200))
            for e in (a, b): e.make_game(np.ascontiguousarray(start[first:first + count]), first=first)
            desc = f"upload {first}+{count}"
        else:
            a.status(); b.status()
            desc = "status"
        if "--every" in sys.argv or ("--fourth" in sys.argv and op % 4 == 3):
            bad = same(a.get_state(), b.get_state())
            print("  op", op, desc, "->", len(bad), "envs differ", sorted(set((bad // 16).tolist()))[:10], a.issue_info(), flush=True)
        else:
            print("  op", op, desc, flush=True)
    bad = same(a.get_state(), b.get_state())
    print("  end:", len(bad), "envs differ", np.array_equal(a.counters(), b.counters()))
    a.close(); b.close()
