#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_call3; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for v in "" _ahead32 _ahead128; do
  for rk in 0 3 7; do
    CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis$v.so python tools/shard_rank.py subpixel_1g_hotspot 8 $rk profiles/r04_tile_loads_config5_hotspot.npy 2>&1 | grep "rank $rk of" | sed "s/^/[ahead$v] /"
  done
done
CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_ahead32.so python tools/shard_rank.py subpixel_1g_hotspot 1 0 2>&1 | grep "rank 0 of" | sed "s/^/[ahead32] /"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rank0 -o r -- python $GRAFT_REPO_ROOT/tools/shard_rank.py subpixel_1g_hotspot 8 0 $GRAFT_REPO_ROOT/profiles/r04_tile_loads_config5_hotspot.npy > $OUT/rank0.txt 2>&1
find $OUT/prof_rank0 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}'
find $OUT -name "*kernel_trace.csv" -delete
