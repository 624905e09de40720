#!/bin/bash
# scripts/ab2.sh libA.so libB.so — A/B of two builds in ONE GPU session, interleaved: one launch per step, then the default three
A=$1; B=$2
Q="--steps 500 --warmup 50 --no-cpu-baseline --no-config3 --no-traffic"
for S in 1 3; do
  for rep in 1 2 3; do
    for L in $A $B; do
      POM_LIB=$L python3 bench.py $Q --streams $S 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$S stream(s) $L %.3f us' % (r['ms_per_step']*1e3))"
    done
  done
done
