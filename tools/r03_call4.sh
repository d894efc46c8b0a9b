#!/bin/bash
# Round 3, GPU call 4: full suite on the sharded-list / pipelined-body / builder-cone changes; stripe height experiments.
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03d; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/pytest.txt
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $O/pytest.txt | tail -30
python tools/group_host_time.py 8 > $O/group_host_time.txt 2>&1; grep "ranks on" $O/group_host_time.txt
CHORDVIS_AB_OLD_LIB=1 CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_r02.so python tools/group_host_time.py 8 > $O/group_host_time_r02.txt 2>&1; grep "ranks on" $O/group_host_time_r02.txt | sed 's/^/r02: /'
RANKS=8 STRIPE=270 python tools/shard_time.py subpixel_1g 2>&1 | grep "^ranks" > $O/shard_time_c5_270.txt; cat $O/shard_time_c5_270.txt
for s in 32 64; do RANKS=8 STRIPE=$s python tools/shard_time.py subpixel_1g_hotspot 2>&1 | grep "^ranks" > $O/shard_time_hot_$s.txt; cat $O/shard_time_hot_$s.txt; done
RANKS=8 STRIPE=64 python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks" > $O/shard_time_c4_64.txt; cat $O/shard_time_c4_64.txt
