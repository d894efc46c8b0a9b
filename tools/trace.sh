#!/bin/bash
# Kernel trace of PRODUCT frames: bench.py without a single event stamp (--no-stamps), 64 timed steps; tools/timeline.py prints the
# median frame of the trace (every complete frame of the process: warm-up, timed region, the moving path's frames are other sequences
# only when they launch other kernels).  Usage: tools/trace.sh <tag> [bench args...]
TAG=${1:-trace}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 4 --cpu-baseline-frames 0 --no-stamps --no-path $* > $OUT/log.txt 2>&1
echo "frames: bench.py --steps 64 --warmup 4 --no-stamps --no-path $* (no event record anywhere in the process)"
python $GRAFT_REPO_ROOT/tools/timeline.py $OUT/r_kernel_trace.csv
