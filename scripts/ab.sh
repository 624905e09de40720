#!/bin/bash
# scripts/ab.sh libA.so libB.so [bench args] — A/B two builds of the library in ONE GPU session, alternating (box-to-box and
# run-to-run noise is a few per cent: only same-session, interleaved numbers are comparable).  Prints ms per step per run.
A=$1; B=$2; shift 2
ARGS=${*:---steps 400 --warmup 40 --no-cpu-baseline --no-config3 --streams 3}
for rep in 1 2 3; do
  for L in $A $B; do
    POM_LIB=$L python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', 'ms_per_step %.3f us  hip-events %.3f us' % (r['ms_per_step']*1e3, r['roofline']['step_ms_hip_events']*1e3))"
  done
done
