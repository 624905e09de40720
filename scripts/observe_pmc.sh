#!/bin/bash
# scripts/observe_pmc.sh [dtypes...] — instructions per wavefront of pom_observe_kernel (rocprofv3 --pmc over scripts/observe_only.py, last dispatch)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/observe_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for DT in ${*:-codes uint8}; do
  rm -rf $OUT/$DT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d $OUT/$DT -- python3 $REPO/scripts/observe_only.py --dtype $DT > $OUT/$DT.log 2>&1 || { echo "$DT failed"; tail -3 $OUT/$DT.log; continue; }
  python3 - $OUT/$DT $DT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "pom_observe_kernel" in r["Kernel_Name"]:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
t = per[sorted(per)[-1]]
w = t["SQ_WAVES"]
print(f"{sys.argv[2]:8s} {sys.argv[0] and ''}per wavefront: VALU {t['SQ_INSTS_VALU']/w:8.1f}  SALU {t['SQ_INSTS_SALU']/w:8.1f}  LDS {t['SQ_INSTS_LDS']/w:6.1f}  branches {t['SQ_INSTS_BRANCH']/w:6.1f}  wave-cycles x4 {4*t['SQ_WAVE_CYCLES']/w:9.0f}  ({w:.0f} wavefronts)")
PY
  rm -rf $OUT/$DT
done
