#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "binless" 2>&1 | tail -30 > $O/pytest_binless.txt
tail -15 $O/pytest_binless.txt
for i in 1 2; do
python bench.py --cpu-baseline-frames 0 > $O/c3_$i.json 2>$O/c3_$i.err
python bench.py --cpu-baseline-frames 0 --debug-flags 1048576 > $O/c3_nodirect_$i.json 2>/dev/null
done
python bench.py --workload street_x64_4k_hzb --cpu-baseline-frames 0 > $O/c4.json 2>/dev/null
python3 -c "
import json
for n in ('c3_1','c3_2','c3_nodirect_1','c3_nodirect_2','c4'):
    try:
        d = json.load(open('$O/' + n + '.json')); g = d['gpu_ms']
        print(n, '%.4f ms/step %.3f Gtri/s cull %.1f setup %.1f clip %.1f tile %.1f st1 %.1f' % (d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3, g['msStage1']*1e3), d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('small_passes_per_step'))
    except Exception as e: print(n, 'failed', e)"
tail -3 $O/c3_1.err
bash tools/trace.sh r03i_trace > $O/timeline.txt 2>&1; tail -14 $O/timeline.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
grep -a "passed\|failed" $O/pytest.txt | tail -3
