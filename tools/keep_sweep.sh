#!/bin/bash
# How long to keep a tile schedule: two-view bench and the moving path (kept / fresh) per keep value and workload.
#   gpurun -- 'bash tools/keep_sweep.sh TAG "WORKLOAD ..." "KEEP ..." [ENV=VALUE ...]'
TAG=$1; WLS=$2; KEEPS=$3; shift 3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
for wl in $WLS; do for k in $KEEPS; do
  env "$@" python bench.py --workload $wl --steps 100 --cpu-baseline-frames 0 --tile-schedule-keep $k 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['moving_path']
print('%-22s keep %d %s: two views %.4f ms | path kept %.4f fresh %.4f ms' % ('$wl', $k, '$*', d['ms_per_step'], m['ms_per_step'], m['fresh_schedule_ms_per_step']))" | tee -a gpurun_out/$TAG/keep_sweep.txt
done; done
