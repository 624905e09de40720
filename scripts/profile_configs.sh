#!/bin/bash
# scripts/profile_configs.sh <tag> — rocprofv3 summaries for every BASELINE single-GPU config (run through gpurun):
#   head1   config "headline" with ONE launch per step (--streams 1): 65,536 envs in one dispatch, so that
#           bytes / AverageNs / peak is checkable from the kernel-trace alone
#   headc   the headline as bench.py runs it by default: chained launches (one launch over all tiles per tick, two streams):
#           kernel trace + the byte counters (the profiler plays dispatches one at a time while it counts)
#   head3   the headline as 3 sub-batches per step on parallel streams (POM_ISSUE=threads: the default before chained launches): kernel trace only
#   c5      config 5: 65,536 envs, kick / chain-explosion stress boards and move mix
#   c3      config 3: 65,536 envs, 4x SimpleAgent policy fused with the tick
#   c2      config 2: 4,096 envs, random moves
#   tape    explicit Move[4] from a tape in device memory (pom_batch_step_device_many), chained launches
#   big     the headline at 1,048,576 envs: records beyond the 256 MiB memory-side cache, sub-batch launches
# Every config gets the byte counters (FETCH_SIZE / WRITE_SIZE, a pass each): scripts/traffic_json.py turns them into
# profiles/<tag>_traffic.json, one block per config, which bench.py prices each line with.
# Kernel traces and PMC passes are separate runs (never combined with sys/hip traces).  Output: gpurun_out/prof_<tag>/<cfg>/.
set -u
TAG=$1; shift
WHICH=${*:-headc head1 c5 c3 c2 tape big}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TOP=$REPO/gpurun_out/prof_$TAG
mkdir -p $TOP
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-config3 --streams 1"
for CFG in $WHICH; do
  case $CFG in
    head1) ARGS="--steps 200 --warmup 20 $COMMON" ;;
    headc) ARGS="--steps 200 --warmup 20 --no-cpu-baseline --no-config3" ;;
    head3) ARGS="--steps 200 --warmup 20 --no-cpu-baseline --no-config3 --streams 3"; export POM_ISSUE=threads ;;
    c5)    ARGS="--steps 200 --warmup 60 --kind stress --dist stress $COMMON" ;;
    c3)    ARGS="--steps 200 --warmup 200 --policy simple $COMMON" ;;
    c2)    ARGS="--steps 400 --warmup 60 --envs 4096 --no-cpu-baseline --no-config3" ;;
    tape)  ARGS="--steps 200 --warmup 20 --policy tape --no-cpu-baseline --no-config3" ;;
    big)   ARGS="--steps 40 --warmup 10 --envs 1048576 --burn-in 100 --no-cpu-baseline --no-config3" ;;
    *) echo "unknown config $CFG"; exit 2 ;;
  esac
  [ "$CFG" = head3 ] || unset POM_ISSUE
  OUT=$TOP/$CFG
  rm -rf $OUT && mkdir -p $OUT
  echo "== $CFG: bench.py $ARGS"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
  if [ "$CFG" = head3 ]; then
    grep "^{\"metric\"" $OUT/trace.log | tail -1 > $OUT/bench_under_trace.json  # (rocprofv3 prints its own lines after the program's)
    python3 $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
    cat $OUT/summary.txt
    find $OUT -name "*kernel_trace.csv" -size +8M -delete
    continue
  fi
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq1 -- python3 $REPO/bench.py $ARGS > $OUT/sq1.log 2>&1 || { echo "sq1 pass failed"; tail -5 $OUT/sq1.log; exit 1; }
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/sq2 -- python3 $REPO/bench.py $ARGS > $OUT/sq2.log 2>&1 || { echo "sq2 pass failed"; tail -5 $OUT/sq2.log; exit 1; }
  # instruction fetch / branches (round 3): the I-cache counters live in the SQC block, the fetch / branch / SALU-cycle counters in the SQ
  if [ "${POM_PROFILE_ICACHE:-0}" = 1 ]; then
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES --output-format csv -d $OUT/ic1 -- python3 $REPO/bench.py $ARGS > $OUT/ic1.log 2>&1 || { echo "ic1 pass failed"; tail -5 $OUT/ic1.log; }
  rocprofv3 --pmc SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/ic2 -- python3 $REPO/bench.py $ARGS > $OUT/ic2.log 2>&1 || { echo "ic2 pass failed"; tail -5 $OUT/ic2.log; }
  fi
  if [ "${POM_PROFILE_BYTES:-1}" = 1 ]; then
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py $ARGS > $OUT/fetch.log 2>&1 || exit 1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/bench.py $ARGS > $OUT/write.log 2>&1 || exit 1
  fi
  grep "^{\"metric\"" $OUT/trace.log | tail -1 > $OUT/bench_under_trace.json  # (rocprofv3 prints its own lines after the program's)
  python3 $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
  cat $OUT/summary.txt
  # keep the merge-back small: the raw traces are tens of MB
  find $OUT -name "*kernel_trace.csv" -size +8M -delete
  find $OUT -name "*counter_collection.csv" -size +8M -delete
done
