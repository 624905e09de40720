#!/bin/bash
# experiment: de-phasing the wavefronts inside one launch (POM_STAGGER=groups,cycles)
for S in "0,0" "2,4000" "2,8000" "2,12000" "3,4000" "3,8000" "4,3000" "4,6000" "2,16000"; do
  for ST in 1 3; do
    POM_STAGGER=$S python3 bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-config3 --streams $ST 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger $S streams $ST: %.3f us/step' % (r['ms_per_step']*1e3))"
  done
done
