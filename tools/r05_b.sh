#!/bin/bash
# round 5, second GPU call: suite after the tile-kernel change (early first fetch, blocks flag in the work item), single-GPU lines, 8-rank cull
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05b
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r05b/pytest.txt
grep -a "passed\|failed\|Error\|error" gpurun_out/r05b/pytest.txt | tail -8
for a in "" "--workload street_x64_4k_hzb" "--workload street_4k_masked"; do
  python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['gpu_ms']
print('%-32s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f tile %.1f us launches/frame %s' % ('$a', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterChunk']*1e3, d.get('kernel_launches')))"
done
python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver form (20 steps):', d['ms_per_step'], d['value'], d['warmup'])"
MAP=default CULL=sharded RANKS=8 timeout 900 python tools/shard_time.py subpixel_1g 2>&1 | grep "^ranks\|Error\|error" > gpurun_out/r05b/shard_time_c5.txt
sed 's/; per rank clusters.*rank-0 GPU/; rank-0 GPU/' gpurun_out/r05b/shard_time_c5.txt | cut -c1-600
