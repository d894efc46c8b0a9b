#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03y2; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q -k "hzb or two_pass or config4 or config3 or sharded or group or moving" 2>&1 | tail -2
for i in 1 2; do
python bench.py --workload street_x64_4k_hzb --cpu-baseline-frames 0 > $O/c4_$i.json 2>/dev/null
python3 -c "
import json
d = json.load(open('$O/c4_$i.json')); g = d['gpu_ms']
print('c4', '%.4f ms/step %.3f Gtri/s cull %.1f st0 %.1f st1 %.1f setup %.1f clip %.1f tile %.1f' % (d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msStage0']*1e3, g['msStage1']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3))"
done
