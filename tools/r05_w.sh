#!/bin/bash
# The schedule of a first pass kept for TILE_ORDER_KEEP frames (launch_raster): parity (every frame-sequence test renders kept frames),
# then A/B against -DTILE_ORDER_KEEP=0 (--tag nokeep).
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_errors.py -m gpu -x -q > gpurun_out/r05w_pytest.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05w_pytest.txt | tail -3
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_ms']
print('%-44s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us launches %s' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3, d.get('kernel_launches')))"; }
for rep in 1 2; do
for v in keep nokeep; do
  if [ $v = nokeep ]; then export CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_nokeep.so; else unset CHORDVIS_LIB; fi
  python bench.py --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_hzb"
  python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_hzb 20 steps"
  if [ $rep = 1 ]; then
    python bench.py --workload atrium_1080p --no-hzb --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] atrium_1080p"
    python bench.py --workload street_x64_4k_hzb --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_x64_4k_hzb"
    python bench.py --workload street_4k_masked --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_masked"
  fi
done
done
true
