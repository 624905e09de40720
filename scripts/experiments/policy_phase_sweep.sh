#!/bin/bash
# diagnostic: config-3 step time against game phase (warm-up ticks) and batch size; run on the GPU box
for W in ${WARMUPS:-100}; do
for N in ${ENVS:-65536}; do
for S in ${STREAMS:-2}; do
  timeout -k 10 120 python bench.py --policy simple --no-cpu-baseline --streams $S --envs $N --warmup $W --steps ${STEPS:-100} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup $W envs $N streams $S:', round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e6,1), 'M env-steps/s, episodes', d['config']['episodes_finished'])" || exit 1
done; done; done
