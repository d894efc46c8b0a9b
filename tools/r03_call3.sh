#!/bin/bash
# Round 3, GPU call 3: rank lists from the cull (no stripe filter), lock-free group exchange, bench fallbacks, hotspot variant,
# 16-byte block loads in the tile kernel.
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/pytest.txt
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $O/pytest.txt | tail -30
b() { # name, args...
  n=$1; shift
  python bench.py --cpu-baseline-frames 0 "$@" > $O/$n.json 2> $O/$n.err
  python3 - <<PY
import json
try:
    d = json.load(open("$O/$n.json")); g = d["gpu_ms"]
    print("%-16s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f tile %.1f us  blocks %d bins %d tiles %s" % ("$n", d["ms_per_step"], d["value"], g["msInstanceCulling"]*1e3, g["msRasterCluster"]*1e3, g["msRasterChunk"]*1e3, d["pixel_blocks_per_step"], d["bin_entries_per_step"], d.get("tiles_touched_view_a")))
except Exception as e:
    print("$n", "FAILED", e)
PY
}
b c3
b c4 --workload street_x64_4k_hzb
b c5 --workload subpixel_1g --steps 10 --warmup 2
b hot64 --workload subpixel_64m_hotspot --steps 20 --warmup 4 --debug-flags 65536
b hot1g --workload subpixel_1g_hotspot --steps 6 --warmup 2
python tools/group_host_time.py 8 > $O/group_host_time.txt 2>&1; grep "ranks on" $O/group_host_time.txt
CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_r02.so python tools/group_host_time.py 8 > $O/group_host_time_r02.txt 2>&1; grep "ranks on" $O/group_host_time_r02.txt | sed 's/^/r02: /'
python tools/shard_time.py subpixel_1g 2>&1 | grep "^ranks" > $O/shard_time_c5.txt; cat $O/shard_time_c5.txt
RANKS=1,8 python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks" > $O/shard_time_c4.txt; cat $O/shard_time_c4.txt
RANKS=8 PIPELINED=1 python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks" > $O/shard_time_c4_pipe.txt; cat $O/shard_time_c4_pipe.txt
RANKS=1,8 python tools/shard_time.py subpixel_1g_hotspot 2>&1 | grep "^ranks" > $O/shard_time_hot.txt; cat $O/shard_time_hot.txt
