#!/bin/bash
# The short-scene group cull with four lanes per group instance (CULL_QUAD): parity, then A/B against -DCULL_QUAD=0 (--tag noquad).
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_depth_views.py tests/test_gpu_errors.py -m gpu -x -q > gpurun_out/r05t_pytest.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05t_pytest.txt | tail -3
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_ms']
print('%-44s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3))"; }
for rep in 1 2; do
for v in quad noquad; do
  if [ $v = noquad ]; then export CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_noquad.so; else unset CHORDVIS_LIB; fi
  python bench.py --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_hzb"
  python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_hzb 20 steps"
  if [ $rep = 1 ]; then
    python bench.py --workload atrium_1080p --no-hzb --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] atrium_1080p"
    python bench.py --workload street_4k_masked --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_masked"
  fi
done
done
true
