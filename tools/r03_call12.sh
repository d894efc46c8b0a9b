#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03k; mkdir -p $O
for f in 0 2097152 4194304 12582912; do
python bench.py --cpu-baseline-frames 0 --debug-flags $f > $O/c3_$f.json 2>/dev/null
done
python3 -c "
import json
for n in ('c3_0','c3_2097152','c3_4194304','c3_12582912'):
    try:
        d = json.load(open('$O/' + n + '.json')); g = d['gpu_ms']
        print(n, '%.4f ms/step %.3f Gtri/s cull %.1f setup %.1f clip %.1f tile %.1f st1 %.1f' % (d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3, g['msStage1']*1e3), d.get('small_passes_per_step'))
    except Exception as e: print(n, 'failed', e)"
