#!/bin/bash
# The eight-lane HZB cull on LONG lists too (-DHZB_CULL_OCT_LONG=1, --tag octlong) against the default (one thread per command above
# 65 536 commands of capacity): config 4 (67 k commands per frame), its x16 sibling, config 4 parity.
cd $GRAFT_REPO_ROOT
CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_octlong.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4 or hzb_culling_lists" > gpurun_out/r05y_pytest.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05y_pytest.txt | tail -3
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_ms']
print('%-44s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us  stage0 %.1f stage1 %.1f' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3, g['msStage0']*1e3, g['msStage1']*1e3))"; }
for rep in 1 2; do
for v in default octlong; do
  if [ $v = octlong ]; then export CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_octlong.so; else unset CHORDVIS_LIB; fi
  python bench.py --workload street_x64_4k_hzb --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_x64_4k_hzb"
  [ $rep = 1 ] && python bench.py --workload street_x16_4k_hzb --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_x16_4k_hzb"
done
done
true
