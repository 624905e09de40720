#!/bin/bash
for K in 1 2 3 5 10 20 40 80; do for rep in 1 2; do echo -n "K=$K: "; python3 bench.py --steps $K --warmup 5 --no-cpu-baseline --no-config3 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us per step, region %.1f us' % (r['ms_per_step']*1e3, r['ms_per_step']*1e3*r['steps']))"; done; done
