#!/bin/bash
# round 5: where the alpha-tested triangles are scan-converted -- own pass (LPT order) vs inside the tile kernel (now without scratch), 1 / 2 px per trip
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05d
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['gpu_ms']
print('%-44s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order(+masked pass) %.1f tile %.1f us launches %s' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3, d.get('kernel_launches')))"; }
for v in "" mf mf2 m2; do
  lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${v:+_$v}.so
  CHORDVIS_LIB=$lib timeout 600 python -m pytest tests/test_masked.py tests/test_gpu_parity.py -m gpu -x -q -k "masked" 2>&1 | tail -2 | head -1
  CHORDVIS_LIB=$lib python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 --workload street_4k_masked 2>/dev/null | line "masked [${v:-separate pass}]"
done
python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 --workload street_4k_masked_twin 2>/dev/null | line "twin"
