#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03g; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
grep -a "passed\|failed\|Error\|assert" $O/pytest.txt | tail -5
tools/profile.sh r03g_c4 --workload street_x64_4k_hzb > /dev/null
python bench.py --workload street_x64_4k_hzb --cpu-baseline-frames 0 > $O/c4.json 2>/dev/null
python bench.py --cpu-baseline-frames 0 > $O/c3.json 2>/dev/null
python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks" > $O/shard_c4.txt
PIPELINED=1 RANKS=8 python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks" >> $O/shard_c4.txt
python3 -c "
import json
for n in ('c3','c4'):
    d = json.load(open('$O/' + n + '.json')); g = d['gpu_ms']
    print(n, '%.4f ms/step %.3f Gtri/s cull %.1f setup %.1f clip %.1f tile %.1f' % (d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3), d['roofline']['frac'])"
cat $O/shard_c4.txt | cut -c1-120
