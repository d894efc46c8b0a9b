#!/bin/bash
# round 5, third GPU call: the masked pass as a kernel of its own; tile / setup kernels with the per-item opaque thread index
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05c
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r05c/pytest.txt
grep -a "passed\|failed\|Error\|error" gpurun_out/r05c/pytest.txt | tail -8
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['gpu_ms']
print('%-44s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f tile %.1f us launches %s' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterChunk']*1e3, d.get('kernel_launches')))"; }
for a in "" "--workload street_x64_4k_hzb" "--workload street_4k_masked" "--workload street_4k_masked_twin" "--workload subpixel_64m --debug-flags 65536"; do
  python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 $a 2>/dev/null | line "$a"
done
CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_m2.so python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 --workload street_4k_masked 2>/dev/null | line "masked, 2 px per trip"
python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver form (20 steps):', d['ms_per_step'], d['value'], d['warmup'])"
