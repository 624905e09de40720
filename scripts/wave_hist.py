#!/usr/bin/env python3
"""Diagnostic (POM_DIAG build): distribution of the per-wavefront tick time within ONE launch — a launch ends with its slowest
wavefront.  usage: python scripts/wave_hist.py [--envs N] [--kind ffa|stress] [--dist 0|1|2]"""
import argparse, ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--kind", default="ffa")
ap.add_argument("--dist", type=int, default=1)
ap.add_argument("--lib", default="", help="a prebuilt POM_DIAG library (default: build one from the tree)")
a = ap.parse_args()
lib = a.lib or os.path.join(ROOT, "build", "libpom_batch_diag.so")
if not a.lib:
    subprocess.run(["hipcc", "-Os", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DPOM_DIAG", "-I" + ROOT + "/include",
                    "-I" + ROOT + "/pomcpp_amd/csrc", "-o", lib, ROOT + "/pomcpp_amd/csrc/pom_batch.hip"], check=True)
import pomcpp_amd.batch as B
B.library_path = lambda: lib
import pomcpp_amd as pa
env = B.BatchEnvironment(a.envs, mode=B.MODE_ENV, auto_reset=True, max_steps=800, streams=1)
env.make_game(pa.make_boards(a.envs, seed=1, kind=a.kind))
env.step_random(1, a.dist, ticks=100)
L = B.load_library()
nw = (a.envs + 63) // 64 * 64 // 16
buf = np.zeros((nw, 17), dtype=np.int64)
L.pom_diag_read_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
assert L.pom_diag_read_raw(env._h, buf.ctypes.data, nw) == 0
names = ["load", "flames", "prep", "agents", "bomb pass", "loop A", "loop B", "explosions", "epilogue", "store", "x look", "x commit", "x bookkeeping", "x nest", "x short", "restart+draw", "flame timers"]
tot = []
per = []
for t in range(20):
    env.step_random(1, a.dist, ticks=1)
    assert L.pom_diag_read_raw(env._h, buf.ctypes.data, nw) == 0
    w = buf[: a.envs // 16]
    tot.append(w.sum(axis=1).copy())
    per.append(w.copy())
tot = np.concatenate(tot)
per = np.concatenate(per)
q = lambda x, p: np.percentile(x, p)
print(f"envs {a.envs} {a.kind} dist {a.dist}: per-wavefront tick cycles over 20 launches: mean {tot.mean():.0f} p50 {q(tot,50):.0f} p90 {q(tot,90):.0f} "
      f"p99 {q(tot,99):.0f} p99.9 {q(tot,99.9):.0f} max {tot.max()}")
comp = per[:, 1:9].sum(axis=1) + per[:, 10:].sum(axis=1)
print(f"  compute only (flames..epilogue): mean {comp.mean():.0f} p50 {q(comp,50):.0f} p90 {q(comp,90):.0f} p99 {q(comp,99):.0f} p99.9 {q(comp,99.9):.0f} max {comp.max()}")
for k, n in enumerate(names):
    c = per[:, k]
    print(f"  {n:12s} mean {c.mean():8.0f} p50 {q(c,50):8.0f} p90 {q(c,90):8.0f} p99 {q(c,99):8.0f} max {c.max():8d}")
slow = per[tot >= q(tot, 99)]
print("  the slowest 1 % of wavefront-ticks spend, on average:", {n: int(slow[:, k].mean()) for k, n in enumerate(names)})
order = np.argsort(-tot)[:12]
print("  the 12 slowest wavefront-ticks (cycles per phase):")
for o in order:
    print("   ", int(tot[o]), {n: int(per[o, k]) for k, n in enumerate(names) if per[o, k] > 600})
