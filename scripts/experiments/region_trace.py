#!/usr/bin/env python3
"""Diagnostic: the kernel timeline of bench.py's timed region out of a rocprofv3 kernel trace
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-traffic).
usage: python scripts/region_trace.py DIR/*/*_kernel_trace.csv [steps=20] [parts=3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r["Queue_Id"]) for r in rows)
idx = [i for i, k in enumerate(ks) if "reduce_counters" in k[2]]
j = idx[1]  # the first one warms the communicator; the second follows the timed region
stepk = [k for k in ks[:j] if "pom_step_kernel" in k[2]][-steps * parts:]
t0 = stepk[0][0]
print(f"timed region: {len(stepk)} step kernels, first start -> last end {(max(k[1] for k in stepk) - t0) / 1e3:.1f} us; "
      f"counter reduction starts {(ks[j][0] - max(k[1] for k in stepk)) / 1e3:.1f} us after the last step kernel and takes {(ks[j][1] - ks[j][0]) / 1e3:.1f} us")
for q in sorted({k[3] for k in stepk}):
    mine = [k for k in stepk if k[3] == q]
    gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(mine, mine[1:])]
    print(f"  queue {q}: {len(mine)} kernels, first start {(mine[0][0] - t0) / 1e3:6.1f} us, last end {(mine[-1][1] - t0) / 1e3:6.1f} us, "
          f"mean duration {sum(k[1] - k[0] for k in mine) / len(mine) / 1e3:5.1f} us, gaps between consecutive kernels: max {max(gaps):.1f} us, sum {sum(gaps):.1f} us")
