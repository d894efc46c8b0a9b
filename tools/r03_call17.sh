#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03p; mkdir -p $O
for w in subpixel_64m subpixel_64m_hotspot subpixel_64m; do
python bench.py --workload $w --steps 40 --warmup 10 --debug-flags 65536 --cpu-baseline-frames 0 > $O/$w.json 2>/dev/null
python3 -c "
import json
d = json.load(open('$O/$w.json')); g = d['gpu_ms']
print('$w', '%.4f ms/step %.3f Gtri/s setup %.3f tile %.3f' % (d['ms_per_step'], d['value'], g['msRasterCluster'], g['msRasterChunk']))"
done
