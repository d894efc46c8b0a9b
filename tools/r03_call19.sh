#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hotspot or pixel_blocks or config5_subpixel_reduced or sharded" 2>&1 | tail -4
for tag in base "" base ""; do
  lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${tag:+_$tag}.so
  CHORDVIS_LIB=$lib python bench.py --workload subpixel_64m --steps 40 --warmup 10 --debug-flags 65536 --cpu-baseline-frames 0 > $O/t.json 2>/dev/null
  python3 -c "
import json
d = json.load(open('$O/t.json')); g = d['gpu_ms']
print('${tag:-product}', '%.4f ms/step setup %.3f tile %.3f' % (d['ms_per_step'], g['msRasterCluster'], g['msRasterChunk']))"
done
for w in subpixel_1g subpixel_1g_hotspot; do
python bench.py --workload $w --steps 10 --warmup 4 --cpu-baseline-frames 0 > $O/$w.json 2>/dev/null
python3 -c "
import json
d = json.load(open('$O/$w.json')); g = d['gpu_ms']
print('$w', '%.4f ms/step %.3f Gtri/s setup %.3f tile %.3f' % (d['ms_per_step'], d['value'], g['msRasterCluster'], g['msRasterChunk']))"
done
