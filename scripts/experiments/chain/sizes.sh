#!/bin/bash
# chained launches against the default issue mode over batch sizes and stream counts (run through gpurun)
Q="--no-cpu-baseline --no-config3 --no-traffic"
for n in 4096 16384 32768 65536 131072 262144; do
  for cfg in "threads 0" "chain 2" "chain 3"; do
    set -- $cfg
    POM_ISSUE=$1 python3 bench.py --envs $n --streams $2 --steps 300 --warmup 30 $Q 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('envs $n $1 streams $2: %.3f us per step  %.3f G' % (r['ms_per_step']*1e3, r['value']/1e9))"
  done
done
