#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03h; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
grep -a "passed\|failed" $O/pytest.txt | tail -3
for i in 1 2; do
python bench.py --cpu-baseline-frames 0 > $O/c3_$i.json 2>/dev/null
python bench.py --workload street_x64_4k_hzb --cpu-baseline-frames 0 > $O/c4_$i.json 2>/dev/null
done
python bench.py > $O/default.json 2> $O/default.err
python3 -c "
import json
for n in ('c3_1','c3_2','c4_1','c4_2','default'):
    d = json.load(open('$O/' + n + '.json')); g = d['gpu_ms']
    print(n, '%.4f ms/step %.3f Gtri/s cull %.1f setup %.1f tile %.1f' % (d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterChunk']*1e3), d['roofline']['frac'], d['roofline']['traffic'], (d.get('cpu_baseline') or {}).get('value'))"
bash tools/trace.sh r03h_trace > $O/timeline.txt 2>&1; tail -14 $O/timeline.txt
