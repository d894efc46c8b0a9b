#!/bin/bash
# HZB cull of short lists with eight lanes per command (HZB_CULL_OCT): the suite, then A/B against -DHZB_CULL_OCT=0 (--tag nooct).
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05x_pytest.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05x_pytest.txt | tail -3
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_ms']
print('%-44s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us  stage1 %.1f' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3, g['msStage1']*1e3))"; }
for rep in 1 2; do
for v in oct nooct; do
  if [ $v = nooct ]; then export CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_nooct.so; else unset CHORDVIS_LIB; fi
  python bench.py --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_hzb"
  python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_hzb 20 steps"
  [ $rep = 1 ] && python bench.py --workload street_4k_masked --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_masked"
done
done
true
