#!/usr/bin/env python3
"""Diagnostic: what each section of pom_policy_kernel costs on the real workload.  Builds variants that run one section twice
(POM_POL_DUP=k, results unchanged, never shipped) and times config 3 with each; cost(k) = time(k) - time(0).
usage (on the GPU box): python scripts/policy_dup.py [--envs N]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--warmup", type=int, default=100)
a = ap.parse_args()
names = {0: "baseline", 1: "prepare (danger map + sets)", 2: "forward_reach", 3: "safe_place window", 4: "move_towards (backward fill)",
         5: "one_safe_step", 6: "predicates"}
res = {}
for k in names:
    lib = os.path.join(ROOT, "build", f"libpom_dup{k}.so")
    subprocess.run(["hipcc", "-Os", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", f"-DPOM_POL_DUP={k}", "-I" + ROOT + "/include",
                    "-I" + ROOT + "/pomcpp_amd/csrc", "-o", lib, ROOT + "/pomcpp_amd/csrc/pom_batch.hip"], check=True)
    env = dict(os.environ, POM_LIB=lib, POM_STREAMS="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--policy", "simple", "--envs", str(a.envs), "--steps", str(a.steps),
                          "--warmup", str(a.warmup), "--no-cpu-baseline", "--streams", "2"], env=env, capture_output=True, text=True, check=True).stdout
    res[k] = json.loads(out.strip().splitlines()[-1])["ms_per_step"]
    print(f"{names[k]:32s} {res[k] * 1e3:9.1f} us/step   +{(res[k] - res[0]) * 1e3:7.1f} us", flush=True)
