#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "binless" 2>&1 | tail -30 > $O/pytest_binless.txt
tail -5 $O/pytest_binless.txt
for i in 1 2; do
python bench.py --cpu-baseline-frames 0 > $O/c3_$i.json 2>$O/c3_$i.err
done
python3 -c "
import json
for n in ('c3_1','c3_2'):
    try:
        d = json.load(open('$O/' + n + '.json')); g = d['gpu_ms']
        print(n, '%.4f ms/step %.3f Gtri/s cull %.1f setup %.1f clip %.1f tile %.1f st1 %.1f' % (d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3, g['msStage1']*1e3), d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('small_passes_per_step'))
    except Exception as e: print(n, 'failed', e)"
bash tools/trace.sh r03j_trace > $O/timeline.txt 2>&1; tail -12 $O/timeline.txt
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/root/repo/gpurun_out/r03j_trace/r_kernel_trace.csv')))
d = collections.defaultdict(list)
for r in rows:
    d[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print('%-60s n %3d avg %7.1f min %7.1f med %7.1f max %7.1f' % (k[-60:], len(v), sum(v)/len(v), v2[0], v2[len(v2)//2], v2[-1]))
PY
