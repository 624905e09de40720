#!/bin/bash
# A launch onto an idle device with its wavefronts started apart (StepParams.stagger): the driver's 20-step shape, short calls, one plain launch per step
run() { python3 bench.py --no-cpu-baseline --no-config3 "$@" 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us %.2f G' % (r['ms_per_step']*1e3, r['value']/1e9))"; }
HW=$((1<<31))
for rep in 1 2 3; do
for v in 0 2 3 4 5 6 8 $((HW+3)) $((HW+4)) $((HW+5)); do
  echo -n "rep $rep chained 20 steps, stagger $v: "; POM_STAGGER=$v run --steps 20 --warmup 5
done
for v in "4 2" "3 2" "5 3"; do set -- $v
  echo -n "rep $rep chained 20 steps, stagger $1 on the first $2 launches: "; POM_STAGGER=$1 POM_STAGGER_LAUNCHES=$2 run --steps 20 --warmup 5
done; done
for K in 5 10 40; do for v in 0 4; do echo -n "chained $K steps, stagger $v: "; POM_STAGGER=$v run --steps $K --warmup 5; done; done
for rep in 1 2; do for v in 0 2 3 4 5 $((HW+3)) $((HW+4)); do
  echo -n "rep $rep one plain launch per step (--streams 1, 200 steps), stagger $v: "; POM_STAGGER_PLAIN=$v run --steps 200 --warmup 20 --streams 1
done; done
