#!/bin/bash
# round 5: one rank of the 8-rank config-5 frame under the kernel trace (where the sharded cull's 0.09 ms go), block merge 2 vs 3 groups deep, regression lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05e; export TMPDIR=/tmp
( cd /tmp && FRAMES=10 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05e/rank3 -o r -- python $GRAFT_REPO_ROOT/tools/shard_rank.py subpixel_1g 8 3 > $GRAFT_REPO_ROOT/gpurun_out/r05e/rank3.log 2>&1 )
grep "rank 3 of 8" gpurun_out/r05e/rank3.log
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r05e/rank3/**/r_kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'chord' in r['Name']: print('%-70s calls %4s avg %9.1f us' % (r['Name'].replace('void ','').replace('chord::','').split('(')[0][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_mb2.so FRAMES=10 python tools/shard_rank.py subpixel_1g 8 3 2>&1 | grep "rank 3 of 8" | sed 's/^/[merge 2 groups deep] /'
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['gpu_ms']
print('%-44s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f tile %.1f us launches %s' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterChunk']*1e3, d.get('kernel_launches')))"; }
for a in "" "--workload street_x64_4k_hzb" "--workload street_4k_masked" "--workload subpixel_64m --debug-flags 65536"; do
  python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 $a 2>/dev/null | line "$a"
done
find gpurun_out/r05e/rank3 -name "*kernel_trace*" -delete
