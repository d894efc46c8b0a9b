#!/bin/bash
# Phase-1 HZB cull: levels 6.. reduced and stored by the grid's last workgroup, by a workgroup with commands only when a command asks
# for one.  The suite, then A/B against the commit before (--tag prev).
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05zz_pytest.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05zz_pytest.txt | tail -3
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_ms']
print('%-44s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us  stage1 %.1f' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3, g['msStage1']*1e3))"; }
for rep in 1 2; do
for v in new prev; do
  if [ $v = prev ]; then export CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_prev.so; else unset CHORDVIS_LIB; fi
  python bench.py --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_hzb"
  python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_hzb 20 steps"
done
done
true
