#!/bin/bash
# Four-lane group cull: the group's loads ahead of the object phase.  Cull parity, then A/B against the commit before (--tag prev).
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "instance_culling or first_frame or two_pass or moving_camera or nothing_in_view or config2 or hierarchical" > gpurun_out/r05v_pytest.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05v_pytest.txt | tail -3
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_ms']
print('%-44s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3))"; }
for rep in 1 2; do
for v in new prev; do
  if [ $v = prev ]; then export CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_prev.so; else unset CHORDVIS_LIB; fi
  python bench.py --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_hzb"
  python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_hzb 20 steps"
  python bench.py --workload atrium_1080p --no-hzb --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] atrium_1080p"
done
done
true
