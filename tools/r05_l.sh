#!/bin/bash
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['gpu_ms']
print('%-52s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us launches %s' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3, d.get('kernel_launches')))"; }
for f in 0 1048576; do
  for a in "" "--workload street_x64_4k_hzb" "--workload atrium_1080p --no-hzb"; do
    python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 --debug-flags $f $a 2>/dev/null | line "[debug $f] $a"
  done
done
