#!/bin/bash
# SQ counter passes for bench.py (separate runs, counters only). Usage: tools/pmc.sh <tag> "<counters>" [bench args]
TAG=$1; CNT=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --cpu-baseline-frames 0 $* > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$OUT/r_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for r in rows:
    k=r['Kernel_Name'].replace('void ','').replace('chord::','').split('(')[0]
    a=agg[k][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k in sorted(agg):
    if 'raster' in k or 'cull' in k or 'hzb' in k:
        print(k, {c: round(v[1]/v[0]) for c,v in agg[k].items()})
PY
