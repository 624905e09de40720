#!/bin/bash
# chained launches: streams x tile-order rotation x call length (the driver's shape: --steps 20 --warmup 5)
run() { python3 bench.py --no-cpu-baseline --no-config3 "$@" 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us %.2f G' % (r['ms_per_step']*1e3, r['value']/1e9))"; }
for rep in 1 2 3; do
for cfg in "streams2_rot16" "streams2_rot0" "streams2_rot8" "streams2_rot32" "streams3_rot16" "streams3_rot0" "streams4_rot0"; do
  s=${cfg%%_*}; s=${s#streams}; r=${cfg##*rot}
  echo -n "20 steps $cfg: "; POM_CHAIN_ROT_DIV=$r run --steps 20 --warmup 5 --streams $s
done; done
for rep in 1 2; do for s in 2 3 4 5; do for r in 0 16; do echo -n "500 steps streams$s rot$r: "; POM_CHAIN_ROT_DIV=$r run --steps 500 --warmup 20 --streams $s; done; done; done
for K in 10 40 60; do for s in 2 3; do echo -n "$K steps streams$s: "; run --steps $K --warmup 5 --streams $s; done; done
