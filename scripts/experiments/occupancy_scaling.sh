#!/bin/bash
# diagnostic: step time against resident wavefronts per SIMD (16384 envs = 1 per SIMD ... 65536 = 4), one launch per step
cd "$(dirname "$0")/../.."
for N in 4096 16384 32768 49152 65536 81920 98304 131072 196608 262144; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-config3 --streams ${STREAMS:-1} --envs $N --steps 300 --warmup 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('envs $N:', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M env-steps/s', round(d['ms_per_step']*1e3/($N/16384.0),2), 'us per 16384 envs')" || exit 1
done
