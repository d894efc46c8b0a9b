#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03z; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
grep -a "passed\|failed\|Error\|assert" $O/pytest.txt | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > $O/default.json 2> $O/default.err; cat $O/default.json | cut -c1-1500
for w in subpixel_1g_hotspot; do
python bench.py --workload $w --steps 10 --warmup 4 --cpu-baseline-frames 0 > $O/$w.json 2>/dev/null
python3 -c "
import json
d = json.load(open('$O/$w.json')); g = d['gpu_ms']
print('$w', '%.4f ms/step %.3f Gtri/s setup %.3f tile %.3f' % (d['ms_per_step'], d['value'], g['msRasterCluster'], g['msRasterChunk']))"
done
