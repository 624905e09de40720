#!/bin/bash
# A/B of library builds (POM_LIB): usage ab_libs.sh lib1 lib2 ... ; shapes: driver (20 steps), long (500), plain launch, stress, closed loop
run() { python3 bench.py --no-cpu-baseline --no-config3 "$@" 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us %.2f G' % (r['ms_per_step']*1e3, r['value']/1e9))"; }
for rep in 1 2 3; do for lib in "$@"; do
  export POM_LIB=$PWD/$lib
  echo -n "$lib rep $rep  driver shape: "; run --steps 20 --warmup 5
  echo -n "$lib rep $rep  500 steps: "; run --steps 500 --warmup 50
  echo -n "$lib rep $rep  plain launch: "; run --steps 200 --warmup 20 --streams 1
  echo -n "$lib rep $rep  stress: "; run --steps 100 --warmup 20 --kind stress --dist stress
  echo -n "$lib rep $rep  simple: "; run --steps 100 --warmup 20 --policy simple
  echo -n "$lib rep $rep  tape: "; run --steps 200 --warmup 20 --policy tape
  echo -n "$lib rep $rep  262144 envs: "; run --steps 100 --warmup 20 --envs 262144
done; done
