#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03x; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
grep -a "passed\|failed\|Error\|assert" $O/pytest.txt | tail -8
for w in street_4k_hzb street_x64_4k_hzb; do
python bench.py --workload $w --cpu-baseline-frames 0 > $O/$w.json 2>/dev/null
done
for w in subpixel_1g subpixel_1g_hotspot; do
python bench.py --workload $w --steps 10 --warmup 4 --cpu-baseline-frames 0 > $O/$w.json 2>/dev/null
done
python3 -c "
import json
for n in ('street_4k_hzb','street_x64_4k_hzb','subpixel_1g','subpixel_1g_hotspot'):
    try:
        d = json.load(open('$O/' + n + '.json')); g = d['gpu_ms']
        print(n, '%.4f ms/step %.3f Gtri/s cull %.1f setup %.1f clip %.1f tile %.1f' % (d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3), d['roofline']['frac'])
    except Exception as e: print(n, 'failed', e)"
