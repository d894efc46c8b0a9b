#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hotspot or pixel_blocks or config5_subpixel_reduced" 2>&1 | tail -4
for w in subpixel_64m subpixel_64m_hotspot; do
python bench.py --workload $w --steps 40 --warmup 10 --debug-flags 65536 --cpu-baseline-frames 0 > $O/$w.json 2>/dev/null
done
for w in subpixel_1g subpixel_1g_hotspot; do
python bench.py --workload $w --steps 10 --warmup 2 --cpu-baseline-frames 0 > $O/$w.json 2>/dev/null
done
python3 -c "
import json
for n in ('subpixel_64m','subpixel_64m_hotspot','subpixel_1g','subpixel_1g_hotspot'):
    try:
        d = json.load(open('$O/' + n + '.json')); g = d['gpu_ms']
        print(n, '%.4f ms/step %.3f Gtri/s setup %.3f tile %.3f' % (d['ms_per_step'], d['value'], g['msRasterCluster'], g['msRasterChunk']), d['bin_entries_per_step'], d['pixel_blocks_per_step'])
    except Exception as e: print(n, 'failed', e)"
