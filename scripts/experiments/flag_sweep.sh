#!/bin/bash
# diagnostic: headline step time of variant builds (compiler flags); GPU box
cd "$(dirname "$0")/../.."
i=0
while IFS= read -r flags; do
  i=$((i+1))
  lib=build/libpom_flags$i.so
  hipcc $flags --offload-arch=gfx950 -std=c++17 -shared -fPIC -Iinclude -Ipomcpp_amd/csrc -o $lib pomcpp_amd/csrc/pom_batch.hip 2>/dev/null || { echo "flags [$flags]: build failed"; continue; }
  for r in 1 2; do
  POM_LIB=$PWD/$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-config3 --streams 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags [$flags]:', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M')"
  done
done <<'FLAGS'
-O3
-O2
-Os
-O3 -mllvm -amdgpu-sched-strategy=max-ilp
-O3 -mllvm -amdgpu-sched-strategy=max-memory-clause
-O3 -mllvm -amdgpu-use-divergent-register-indexing
-O3 -mllvm -amdgpu-early-inline-all=true -mllvm -amdgpu-function-calls=false
-O3 -fno-jump-tables -mllvm -amdgpu-enable-max-ilp-scheduling-strategy
FLAGS
