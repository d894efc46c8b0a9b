#!/bin/bash
# A/B of ONE environment switch of the product library, interleaved on one box, then (optionally) the GPU suite with a per-test timeout.
#   gpurun -- 'bash tools/ab_env.sh TAG NAME "WORKLOAD ..." [REPS] [suite]'      (NAME=0 against NAME=1)
TAG=$1; NAME=$2; WLS=$3; REPS=${4:-2}; SUITE=${5:-}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
for rep in $(seq 1 $REPS); do for e in 1 0; do for wl in $WLS; do
  env $NAME=$e timeout 300 python bench.py --workload $wl --steps 200 --cpu-baseline-frames 0 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d.get('moving_path') or {}; g=d['gpu_ms']
    print('%-20s $NAME=$e  %.4f ms  launches %s  tile %.1f us  path %.4f fresh %.4f' % ('$wl', d['ms_per_step'], d.get('kernel_launches'), g['msRasterChunk']*1e3, m.get('ms_per_step',0), m.get('fresh_schedule_ms_per_step',0)))
except Exception as ex: print('$wl $NAME=$e FAILED', ex)" | tee -a $O/ab.txt
done; done; done
if [ -n "$SUITE" ]; then
  timeout 1500 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $O/pytest.txt 2>&1
  grep -a "passed\|failed\|Timeout" $O/pytest.txt | tail -3; grep -a "^FAILED\|^E  " $O/pytest.txt | head -20
fi
