#!/bin/bash
# the driver's 20-step shape with the chained launches issued by one thread / with the sub-streams' helper threads (POM_CHAIN_HELPERS=0 / 1), alternating
for i in $(seq 1 ${1:-16}); do for v in 1 0; do
  export POM_CHAIN_HELPERS=$v
  POM_BENCH_TRACE=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config3 2> /tmp/trace.err | tail -1 | LABEL=$v python3 -c "import sys,json,os; r=json.loads(sys.stdin.read()); print('helpers %s: %.3f G %.2f us' % (os.environ['LABEL'], r['value']/1e9, r['ms_per_step']*1e3), end='  ')"
  grep "\[trace\]" /tmp/trace.err | tail -1
done; done
for v in 1 0; do export POM_CHAIN_HELPERS=$v; for k in 1 2; do python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-config3 2>/dev/null | tail -1 | LABEL=$v python3 -c "import sys,json,os; r=json.loads(sys.stdin.read()); print('helpers %s, 500 steps: %.3f G %.2f us' % (os.environ['LABEL'], r['value']/1e9, r['ms_per_step']*1e3))"; done; done
