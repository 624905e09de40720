#!/bin/bash
# scripts/bench_set.sh <tag> — the bench lines kept under profiles/ (run through gpurun): the default run, the driver's shape five
# times, and the other sizes / modes.  Output: gpurun_out/bench_<tag>/.  scripts/bench_set.sh <tag> sizes: only the other sizes / modes.
TAG=$1
OUT=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/bench_$TAG
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-/root/repo}
Q="--no-cpu-baseline --no-config3 --no-traffic"
if [ "${2:-all}" = all ]; then
python3 bench.py 2>/dev/null | tail -1 > $OUT/bench.json && echo default done
for i in 1 2 3 4 5; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1; done > $OUT/bench_driver_shape.jsonl && echo driver-shape done
fi
python3 bench.py --envs 262144 --steps 200 --warmup 20 $Q 2>/dev/null | tail -1 > $OUT/bench_262144_envs.json
python3 bench.py --envs 32768 --steps 300 --warmup 20 $Q 2>/dev/null | tail -1 > $OUT/bench_32768_envs.json
python3 bench.py --fresh-boards --steps 300 --warmup 20 $Q 2>/dev/null | tail -1 > $OUT/bench_fresh_boards.json
python3 bench.py --policy simple --envs 262144 --steps 100 --warmup 20 $Q 2>/dev/null | tail -1 > $OUT/bench_simple_262144_envs.json
python3 bench.py --policy simple --steps 200 --warmup 50 $Q 2>/dev/null | tail -1 > $OUT/bench_simple_65536_envs.json
for n in 4096 16384 131072 524288 1048576; do python3 bench.py --envs $n --steps 100 --warmup 20 $Q 2>/dev/null | tail -1 > $OUT/bench_${n}_envs.json; done
# the headline with sub-batches on parallel streams (the default before chained launches), for comparison
POM_ISSUE=threads python3 bench.py --steps 500 --warmup 50 $Q 2>/dev/null | tail -1 > $OUT/bench_issue_threads.json
python3 bench.py --policy simple --envs 32768 --steps 200 --warmup 50 $Q 2>/dev/null | tail -1 > $OUT/bench_simple_32768_envs.json
for f in $OUT/*.json $OUT/*.jsonl; do python3 - "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    d = json.loads(line)
    extra = ""
    if "other_configs" in d:
        extra = "  " + "  ".join(f"{k}: {v['value'] / 1e9:.3f} G" for k, v in d["other_configs"].items())
    if "config3_simple_agent" in d:
        extra += f"  config3: {d['config3_simple_agent']['value'] / 1e9:.3f} G"
    if "cpu_baseline" in d:
        extra += f"  cpu: {d['cpu_baseline']['value'] / 1e6:.1f} M"
    print(f"{sys.argv[1].split('/')[-1]:34s} {d['value'] / 1e9:.3f} G  {d['ms_per_step'] * 1e3:.2f} us/step  frac {d['roofline']['frac']:.3f} contract_frac {d['roofline'].get('contract_frac')}{extra}")
PY
done
