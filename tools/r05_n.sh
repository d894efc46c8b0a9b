#!/bin/bash
# Bins cut finer in passes with fewer tiles than the device has tile-workgroup slots (tile_order_part, RasterParams::tileSlots):
# A/B through CHORDVIS_TILE_SLOTS (0 = the old rule) on the rank-by-rank times of the 8-rank frames and on the 1080p frame.
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q -k "shard or rank or group or config2 or atrium or 1080" > gpurun_out/r05n_pytest.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05n_pytest.txt | tail -3
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_ms']
print('%-40s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3))"; }
for s in 0 512; do
  export CHORDVIS_TILE_SLOTS=$s
  python bench.py --workload atrium_1080p --no-hzb --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[slots $s] atrium_1080p"
  python bench.py --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[slots $s] street_4k_hzb"
  MAP=default RANKS=8 python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks" | sed "s/^/[slots $s] c4 /"
  MAP=default RANKS=8 python tools/shard_time.py subpixel_1g 2>&1 | grep "^ranks" | sed "s/^/[slots $s] c5 /"
done
