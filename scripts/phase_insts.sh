#!/bin/bash
# scripts/phase_insts.sh <out dir> [phase_insts.py args] — one rocprofv3 --pmc run per cut; prints the per-phase differences
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
case $OUT in /*) ;; *) OUT=$REPO/$OUT ;; esac
mkdir -p $OUT
python3 $REPO/scripts/phase_insts.py --build-only || exit 1
cd /tmp && export TMPDIR=/tmp
CUTS="0 10 20 24 27 30 40 50 60 70 990"
case "$*" in *--policy*) CUTS="-4 -3 -2 -1 1 2 3 4 990" ;; esac
for CUT in $CUTS; do
  export POM_TRUNC_CUT=$CUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/cut$CUT -- python3 $REPO/scripts/phase_insts.py "$@" > $OUT/cut$CUT.log 2>&1 || { echo "cut $CUT failed"; tail -3 $OUT/cut$CUT.log; exit 1; }
done
python3 - $OUT "$CUTS" <<'PY'
import csv, glob, sys
names = {0: "before the tick (load, restarts, move draw) + epilogue + store", 10: "flame timers", 20: "flame pops", 24: "agent prep: positions, destinations, contact test",
         27: "agent prep: FixSwitchMove / ResolveDependencies (contact only)", 30: "agent prep: bombs under agents", 40: "agent loop", 50: "bomb reset / classify pass",
         60: "bomb loop A", 70: "bomb loop B", 990: "timer epilogue + top explosions"}
prev = None
cuts = [int(c) for c in sys.argv[2].split()]
if cuts[0] < 0:
    names = {-4: "everything but the policy and the tick (load, agent memory, epilogue, store)", -3: "policy: clear the danger map and sets", -2: "policy: fill (bombs, flames, agents)",
             -1: "policy: safe cells", 1: "act: predicates (danger, enemies near, loop) + the tick's flame timers", 2: "act: target (forward flood + safe place, or the enemy)",
             3: "act: path (backward flood)", 4: "act: exits, one safe step, memory", 990: "the tick"}
for cut in cuts:
    f = glob.glob(f"{sys.argv[1]}/cut{cut}/*/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "pom_step_kernel" in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    c = {r["Counter_Name"]: float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == last}
    w = c["SQ_WAVES"]
    cur = {k: c[k] / w for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES")}
    d = cur if prev is None else {k: cur[k] - prev[k] for k in cur}
    print(f"{names[cut]:64s} VALU {d['SQ_INSTS_VALU']:8.1f}  SALU {d['SQ_INSTS_SALU']:8.1f}  LDS {d['SQ_INSTS_LDS']:6.1f}  wave-cycles x4 {d['SQ_WAVE_CYCLES'] * 4:9.0f}   (per wavefront)")
    prev = cur
print(f"{'whole tick':64s} VALU {prev['SQ_INSTS_VALU']:8.1f}  SALU {prev['SQ_INSTS_SALU']:8.1f}  LDS {prev['SQ_INSTS_LDS']:6.1f}  wave-cycles x4 {prev['SQ_WAVE_CYCLES'] * 4:9.0f}")
PY
