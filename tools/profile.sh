#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box. Usage: tools/profile.sh <tag> [bench args...]
# Writes gpurun_out/<tag>/{stats,pmc_fetch,pmc_write}/... ; counters are collected in their own runs
# (kernel-trace/stats only, never combined with --pmc).
set -u
TAG=${1:-prof}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 40 --warmup 4 --cpu-baseline-frames 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
if [ "${STATS_ONLY:-0}" != "1" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1
fi
find $OUT -type f | head -30
