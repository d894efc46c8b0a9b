#!/bin/bash
TAG=${1:-trace}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 4 --cpu-baseline-frames 0 $* > $OUT/log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/timeline.py $OUT/r_kernel_trace.csv
