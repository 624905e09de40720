#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — child process of bench.py's cpu_baseline leg: times the compiled, unmodified reference
(oracle/_ref/libpomref_bench.so, see oracle/ref_baseline.c) on the host cores and prints one JSON line.  A separate process
because the reference has undefined behaviour on reachable states; should it crash despite the guard, bench.py falls back to
the restatement's figure.  usage: ref_baseline_run.py boards.npy seed dist_id max_steps budget_s
       ref_baseline_run.py boards.npy seed config1      (BASELINE config 1: one env, harmless moves, one thread, 10 x 1000 ticks)"""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
start = np.load(sys.argv[1])
lib = C.CDLL(os.path.join(here, "_ref", "libpomref_bench.so"))
if len(sys.argv) > 3 and sys.argv[3] == "config1":
    lib.ref_run_single_timed.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.ref_run_single_timed.restype = C.c_int64
    one = np.ascontiguousarray(start[:1])
    secs = C.c_double(0)
    ticks, reps = 1000, 10
    lib.ref_run_single_timed(one.ctypes.data, ticks, int(sys.argv[2]), 0, 2, C.byref(secs))  # warm the caches
    secs = C.c_double(0)
    n = lib.ref_run_single_timed(one.ctypes.data, ticks, int(sys.argv[2]), 0, reps, C.byref(secs))
    print(json.dumps({"value": n / secs.value if n > 0 and secs.value > 0 else None, "cores": 1, "steps": int(n), "timed_s": secs.value,
                      "ticks": ticks, "reps": reps}))
    sys.exit(0 if n > 0 else 1)
seed, dist_id, max_steps, budget = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
lib.ref_run_random_timed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_double), C.POINTER(C.c_int64)]
lib.ref_run_random_timed.restype = C.c_int64
try:
    cores = len(os.sched_getaffinity(0))
except AttributeError:
    cores = os.cpu_count() or 1
per_thread = 2048
cores = max(1, min(cores, start.size // per_thread))
steps, secs, skipped = [0] * cores, [C.c_double(0) for _ in range(cores)], [C.c_int64(0) for _ in range(cores)]
deadline = time.perf_counter() + budget


def work(k: int) -> None:
    init = np.ascontiguousarray(start[k * per_thread:(k + 1) * per_thread])
    cur = init.copy()
    tick = 0
    while time.perf_counter() < deadline:
        steps[k] += lib.ref_run_random_timed(cur.ctypes.data, init.ctypes.data, per_thread, 25, seed, k * per_thread, tick, dist_id,
                                             max_steps, C.byref(secs[k]), C.byref(skipped[k]))
        tick += 25


t0 = time.perf_counter()
threads = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
for th in threads:
    th.start()
for th in threads:
    th.join()
rate = sum(s / t.value for s, t in zip(steps, secs) if t.value > 0)
print(json.dumps({"value": rate, "cores": cores, "steps": int(sum(steps)), "skipped_ub_ticks": int(sum(x.value for x in skipped)),
                  "timed_s_per_thread": sum(t.value for t in secs) / cores, "wall_s": time.perf_counter() - t0, "per_thread": per_thread}))
