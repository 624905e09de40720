#!/usr/bin/env python3
"""Closed-loop stepping (pom_batch_step_device_range + a stand-in device policy per tick) at 65,536 envs: how the per-tick time depends on
the number of ranges and on how the launches are issued — a replayed HIP graph, one Python thread issuing every range's launches in turn,
or a thread per range.  usage (GPU box): python scripts/closed_loop_bench.py [--envs N] [--obs]"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pomcpp_amd as pa
from pomcpp_amd.batch import BatchEnvironment, MODE_ENV, RESET_AT_END, bench_policy

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--obs", action="store_true")
ap.add_argument("--ticks", type=int, default=25)
ap.add_argument("--reps", type=int, default=8)
a = ap.parse_args()
n = a.envs
start = pa.make_boards(n, seed=1000003)
dev = torch.device("cuda", 0)
for n_ranges in (1, 2, 4, 8):
    per = n // n_ranges // 16 * 16
    ranges = [(i * per, per if i < n_ranges - 1 else n - i * per) for i in range(n_ranges)]
    for how in ("graph", "one thread", "thread per range"):
        if how == "thread per range" and n_ranges == 1:
            continue
        env = BatchEnvironment(n, device=0, mode=MODE_ENV, auto_reset=RESET_AT_END, max_steps=800)
        env.make_game(start)
        env.step_random(1, 1, ticks=300)
        codes = env.observe(dtype="codes", attrs=False)[0] if a.obs else None
        moves = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        env.sync()
        torch.cuda.synchronize()
        side = [torch.cuda.Stream(device=dev) for _ in ranges]
        main = torch.cuda.Stream(device=dev)

        def chain(i, ticks):
            (f, c), s = ranges[i], side[i]
            for t in range(ticks):
                bench_policy(codes, moves, f, c, t, s)
                env.step_device_range(f, c, moves, s, codes=codes)

        def issue_all(ticks):
            for s in side:
                s.wait_stream(main)
            if how == "thread per range":
                th = [threading.Thread(target=chain, args=(i, ticks)) for i in range(n_ranges)]
                for t_ in th:
                    t_.start()
                for t_ in th:
                    t_.join()
            else:
                for t in range(ticks):
                    for i, ((f, c), s) in enumerate(zip(ranges, side)):
                        bench_policy(codes, moves, f, c, t, s)
                        env.step_device_range(f, c, moves, s, codes=codes)
            for s in side:
                main.wait_stream(s)

        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if how == "graph":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main):
                issue_all(a.ticks)
            g.replay()
            torch.cuda.synchronize()
            with torch.cuda.stream(main):
                ev0.record(main)
                for _ in range(a.reps):
                    g.replay()
                ev1.record(main)
        else:
            issue_all(a.ticks)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(main):
                ev0.record(main)
                issue_all(a.ticks * a.reps)
                ev1.record(main)
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 1e3 / (a.ticks * a.reps)
        print(f"{n} envs, {'codes observation' if a.obs else 'no observation'}, {n_ranges} range(s), {how:16s}: {us:7.2f} us per tick = {n / us / 1e3:.2f} G env-steps/s", flush=True)
        env.close()
