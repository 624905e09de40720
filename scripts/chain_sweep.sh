#!/bin/bash
# driver shape: --steps 20 --warmup 5; vary streams and rotation
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config3 "$@" 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us %.2f G' % (r['ms_per_step']*1e3, r['value']/1e9))"; }
for rep in 1 2 3; do
for cfg in "streams0_rot16" "streams0_rot0" "streams0_rot8" "streams0_rot32" "streams3_rot16" "streams3_rot0" "streams2_rot16"; do
  s=${cfg%%_*}; s=${s#streams}; r=${cfg##*rot}
  echo -n "$cfg: "; POM_CHAIN_ROT_DIV=$r run --streams $s
done; done
echo "long call (500 steps):"
for s in 0 2 3; do echo -n "streams $s: "; python3 bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-config3 --streams $s 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us %.2f G' % (r['ms_per_step']*1e3, r['value']/1e9))"; done
