#!/bin/bash
# round 4: every rank's time on config 5 (64 M, 1 G, hotspot) under the tile maps
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_call1; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
RANKS=1,8 timeout 600 python tools/shard_time.py subpixel_64m > $OUT/shard_64m.txt 2>&1; tail -5 $OUT/shard_64m.txt
RANKS=1,2,4,8 timeout 900 python tools/shard_time.py subpixel_1g > $OUT/shard_1g.txt 2>&1; tail -8 $OUT/shard_1g.txt
RANKS=1,8 timeout 900 python tools/shard_time.py subpixel_1g_hotspot > $OUT/shard_1g_hotspot.txt 2>&1; tail -5 $OUT/shard_1g_hotspot.txt
