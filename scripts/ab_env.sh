#!/bin/bash
# scripts/ab_env.sh VAR "v1 v2 ..." "bench args" [rounds] — one build, an environment knob swept in ONE GPU session, interleaved
VAR=$1; VALS=$2; ARGS=$3; R=${4:-4}
for rep in $(seq $R); do
  for v in $VALS; do
    env $VAR=$v python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', 'ms_per_step %.3f us  %.3f G' % (r['ms_per_step']*1e3, r['value']/1e9))"
  done
done
