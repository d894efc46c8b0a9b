#!/bin/bash
# round 5: the rank that owns the hot tile of the 8-rank hotspot frame (re-balanced map): what its block kernel waits for
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05f
L=profiles/r04_tile_loads_config5_hotspot.npy
for v in "" hot128 nochunk; do
  CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${v:+_$v}.so FRAMES=10 python tools/shard_rank.py subpixel_1g_hotspot 8 0 $L 2>&1 | grep "rank 0 of 8" | sed "s/^/[${v:-product}] /"
done
FRAMES=10 python tools/shard_rank.py subpixel_1g 8 3 2>&1 | grep "rank 3 of 8" | sed "s/^/[uniform, product] /"
( cd /tmp && FRAMES=6 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05f/pmc -o r -- python $GRAFT_REPO_ROOT/tools/shard_rank.py subpixel_1g_hotspot 8 0 $GRAFT_REPO_ROOT/$L > $GRAFT_REPO_ROOT/gpurun_out/r05f/pmc.log 2>&1 )
python - <<'PY'
import csv, collections, glob
f = glob.glob('gpurun_out/r05f/pmc/**/r_counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].replace('void ', '').replace('chord::', '').split('(')[0]
    a = agg[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k in agg:
    if 'blocks' in k or 'tile_kernel' in k: print(k, {c: round(v[1] / v[0]) for c, v in agg[k].items()})
PY
