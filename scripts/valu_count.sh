#!/bin/bash
# scripts/valu_count.sh [configs...] — dynamic instructions per wavefront-tick (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
# SQ_WAVES over scripts/pc_sample_run.py: plain launches, one tick each, steady mix after the burn-in) for head / stress / policy.
# One PMC pass per config, no trace domains.  Run through gpurun; prints one line per config.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/valu_count
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CFG in ${*:-head stress policy}; do
  case $CFG in
    head)   ARGS="--ticks 400" ;;
    stress) ARGS="--ticks 300 --kind stress --dist 2" ;;
    policy) ARGS="--ticks 400 --policy" ;;
    c2)     ARGS="--ticks 400 --envs 4096" ;;
  esac
  rm -rf $OUT/$CFG
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d $OUT/$CFG -- python3 $REPO/scripts/pc_sample_run.py $ARGS > $OUT/$CFG.log 2>&1 || { echo "$CFG failed"; tail -3 $OUT/$CFG.log; continue; }
  python3 - $OUT/$CFG $CFG <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "pom_step_kernel" in r["Kernel_Name"]:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(per)[-100:]  # the last 100 launches: the steady mix
tot = collections.Counter()
for i in ids:
    for k, v in per[i].items():
        tot[k] += v
w = tot["SQ_WAVES"]
print(f"{sys.argv[2]:8s} per wavefront-tick: VALU {tot['SQ_INSTS_VALU']/w:8.1f}  SALU {tot['SQ_INSTS_SALU']/w:8.1f}  LDS {tot['SQ_INSTS_LDS']/w:6.1f}  branches {tot['SQ_INSTS_BRANCH']/w:6.1f}  wave-cycles x4 {4*tot['SQ_WAVE_CYCLES']/w:9.0f}  (last {len(ids)} launches, {w/len(ids):.0f} wavefronts each)")
PY
  rm -rf $OUT/$CFG
done
