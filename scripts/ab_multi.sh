#!/bin/bash
# scripts/ab_multi.sh "bench args" lib1 lib2 ... — several builds in ONE GPU session, interleaved, 3 rounds
ARGS=$1; shift
for rep in 1 2 3; do
  for L in "$@"; do
    POM_LIB=$L python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', 'ms_per_step %.3f us' % (r['ms_per_step']*1e3))"
  done
done
