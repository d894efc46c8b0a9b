#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_call2; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
DUMP=$OUT RANKS=1,8 timeout 600 python tools/shard_time.py subpixel_64m > $OUT/shard_64m.txt 2>&1; tail -3 $OUT/shard_64m.txt
DUMP=$OUT RANKS=1,8 timeout 900 python tools/shard_time.py subpixel_1g_hotspot > $OUT/shard_1g_hotspot.txt 2>&1; tail -3 $OUT/shard_1g_hotspot.txt
DUMP=$OUT RANKS=1,8 timeout 900 python tools/shard_time.py street_x64_4k_hzb > $OUT/shard_c4.txt 2>&1; tail -3 $OUT/shard_c4.txt
python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; python -c "
import json; d=json.load(open('$OUT/bench_c3.json')); print(d['ms_per_step'], d['value'], d['gpu_ms'])"
