#!/bin/bash
# short chained calls: two streams, a third joining after n launches (POM_CHAIN_THIRD_FROM, experimental build)
run() { python3 bench.py --no-cpu-baseline --no-config3 "$@" 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us %.2f G' % (r['ms_per_step']*1e3, r['value']/1e9))"; }
export POM_LIB=$PWD/build/libpom_third.so
for rep in 1 2 3; do for n in 0 2 3 4 6 8 12; do echo -n "rep $rep 20 steps, third stream from launch $n: "; POM_CHAIN_THIRD_FROM=$n run --steps 20 --warmup 5; done; done
for K in 10 40; do for n in 0 3 6; do echo -n "$K steps, third stream from launch $n: "; POM_CHAIN_THIRD_FROM=$n run --steps $K --warmup 5; done; done
