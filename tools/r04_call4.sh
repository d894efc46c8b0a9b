#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_call4; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -15 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"
RANKS=1,8 timeout 900 python tools/shard_time.py subpixel_1g_hotspot > $OUT/shard_1g_hotspot.txt 2>&1; tail -3 $OUT/shard_1g_hotspot.txt
RANKS=1,8 MAP=balanced timeout 900 python tools/shard_time.py street_x64_4k_hzb > $OUT/shard_c4.txt 2>&1; tail -2 $OUT/shard_c4.txt
