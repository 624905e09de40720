#!/bin/bash
# the driver's command shape many times, chained launches against sub-batches + helper threads, interleaved: how often does a
# 20-step region (~240 us) catch a stall?
for i in $(seq 1 ${1:-20}); do
  for m in chain threads; do
    POM_ISSUE=$m python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-traffic 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m %.2f %.2f' % (r['ms_per_step']*1e3, r['roofline']['step_ms_hip_events']*1e3))"
  done
done
