#!/bin/bash
# round 5: hot tiles known from the last frame draw ahead from the first cluster on
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05g
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hotspot or sharded or blocks or config5" 2>&1 | tail -3
L=profiles/r04_tile_loads_config5_hotspot.npy
FRAMES=10 python tools/shard_rank.py subpixel_1g_hotspot 8 0 $L 2>&1 | grep "rank 0 of 8" | sed "s/^/[hot prior] /"
FRAMES=10 python tools/shard_rank.py subpixel_1g 8 3 2>&1 | grep "rank 3 of 8" | sed "s/^/[uniform] /"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['gpu_ms']
print('%-44s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f tile %.1f us launches %s' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterChunk']*1e3, d.get('kernel_launches')))"; }
python bench.py --steps 10 --warmup 2 --cpu-baseline-frames 0 --workload subpixel_1g_hotspot 2>/dev/null | line "hotspot 1 GPU"
python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 --workload street_x64_4k_hzb 2>/dev/null | line "config 4"
