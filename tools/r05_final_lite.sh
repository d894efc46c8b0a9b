#!/bin/bash
# The config-3 evidence and the short-scene bench lines once more after the last kernel change of the round (the HZB cull's eight-lane
# form); the suite ran on this library in the call before (tools/r05_x.sh: 112 passed).  collect_round.py r05 afterwards.
set -u
R=r05
cd $GRAFT_REPO_ROOT
tools/profile.sh ${R}_c3 > /dev/null
bash tools/trace.sh ${R}_trace > gpurun_out/${R}_timeline.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 > gpurun_out/${R}_bench_default_20steps.json 2>/dev/null
python bench.py --workload street_x64_4k_hzb --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c4.json 2>/dev/null
python bench.py --workload atrium_1080p --no-hzb --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c2.json 2>/dev/null
python bench.py --cull hierarchical --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c3_bvh.json 2>/dev/null
python bench.py --workload street_4k_masked --cpu-baseline-frames 0 > gpurun_out/${R}_bench_masked.json 2>/dev/null
python bench.py --workload street_4k_masked_twin --cpu-baseline-frames 0 > gpurun_out/${R}_bench_masked_twin.json 2>/dev/null
