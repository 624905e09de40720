#!/bin/bash
# scripts/profile.sh <tag> [bench args...] — rocprofv3 passes over bench.py on the GPU box (run through gpurun).
# Counters are collected in their own passes (never with sys/hip traces); summaries land in gpurun_out/prof_<tag>/.
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 200 --warmup 20 --no-cpu-baseline --no-config3 ${TRACE_STREAMS:---streams 3} $*"
# kernel durations: the bench as it runs by default (sub-batches on parallel streams).  Counters: ONE launch per step
# (--streams 1) — concurrent dispatches share the counters and would be charged each other's work; bytes and instruction
# counts per env do not depend on how a step is split.
PMC_ARGS="--steps 200 --warmup 20 --no-cpu-baseline --no-config3 --streams 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py $PMC_ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/bench.py $PMC_ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq1 -- python3 $REPO/bench.py $PMC_ARGS > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/sq2 -- python3 $REPO/bench.py $PMC_ARGS > $OUT/sq2.log 2>&1
cp $OUT/trace.log $OUT/bench_under_trace.json 2>/dev/null
python3 $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
