#!/bin/bash
# A rank of the 8-rank config-5 frame under the profiler: kernel stats of the current build, and the fetched bytes of its tile
# kernel (is it bound by the pixel blocks' bytes, as profiles/r05_tile_kernel_experiments.txt item 8 infers from the timings?).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
( FRAMES=10 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05r_rank3 -o r -- python $R/tools/shard_rank.py subpixel_1g 8 3 > $O/r05r_rank3.log 2>&1 ); find $O/r05r_rank3 -name "*kernel_trace*" -delete
( FRAMES=6 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r05r_rank3_fetch -o r -- python $R/tools/shard_rank.py subpixel_1g 8 3 > $O/r05r_rank3_fetch.log 2>&1 )
( FRAMES=6 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r05r_rank3_write -o r -- python $R/tools/shard_rank.py subpixel_1g 8 3 > $O/r05r_rank3_write.log 2>&1 )
python3 - <<PY
import csv, collections, glob
for tag in ("fetch", "write"):
    f = glob.glob("$O/r05r_rank3_%s/**/r_counter_collection.csv" % tag, recursive=True)
    if not f: print(tag, "no counter file"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].replace('void ', '').replace('chord::', '').split('(')[0]
        a = agg[(k, r['Counter_Name'])]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for (k, c), v in sorted(agg.items()):
        if 'raster' in k or 'cull' in k: print("%-50s %-12s launches %3d  per launch %.0f KiB" % (k, c, v[0], v[1] / v[0]))
PY
