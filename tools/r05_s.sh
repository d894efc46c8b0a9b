#!/bin/bash
# Where a bin is cut on ONE GPU: the heaviest unsplit tile is the first work item of the tile kernel and runs as long as the kernel
# does (config 3: 4 476 entries, config 4: up to 6 144).  CHORDVIS_TILE_SPLIT_MIN / CHORDVIS_TILE_SLICE against the defaults 6144 / 2048.
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_ms']
print('%-44s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3))"; }
for cfg in "6144 2048" "4096 2048" "3072 2048" "2048 2048" "2048 1024" "1536 1024" "6144 2048"; do
  set -- $cfg
  export CHORDVIS_TILE_SPLIT_MIN=$1 CHORDVIS_TILE_SLICE=$2
  python bench.py --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[split $1 slice $2] street_4k_hzb"
  python bench.py --workload street_x64_4k_hzb --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[split $1 slice $2] street_x64_4k_hzb"
done
