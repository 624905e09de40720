#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs of scripts/profile.sh into one text summary (per-kernel time + PMC means)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
kernels = ("pom_step_kernel", "pom_policy_kernel")


def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None


f = find("trace", "*kernel_stats.csv")
if f:
    print("== kernel-trace --stats ==")
    for row in csv.DictReader(open(f)):
        print(f"{row['Name'][:70]:70s} calls {row['Calls']:>6s} avg_ns {float(row['AverageNs']):>12.1f} pct {row['Percentage']}")
for kernel in kernels:
    f = find("trace", "*kernel_trace.csv")
    if f:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
        d = d[20:] if len(d) > 40 else d
        if d:
            d.sort()
            print(f"{kernel}: n={len(d)} mean {sum(d)/len(d)/1e3:.2f} us median {d[len(d)//2]/1e3:.2f} us min {d[0]/1e3:.2f} max {d[-1]/1e3:.2f}")
    for sub in ("fetch", "write", "sq1", "sq2", "ic1", "ic2"):
        f = find(sub, "*counter_collection.csv")
        if not f:
            print(f"== {sub}: no counter csv ==")
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if not acc:
            continue
        print(f"== pmc {sub} ({kernel}, mean per dispatch) ==")
        for k, v in acc.items():
            v = v[20:] if len(v) > 40 else v
            print(f"{k:28s} {sum(v)/len(v):18.1f}   (n={len(v)})")
