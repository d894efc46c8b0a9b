#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03q; mkdir -p $O
for tag in base "" base ""; do
  lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${tag:+_$tag}.so
  CHORDVIS_LIB=$lib python bench.py --workload subpixel_64m --steps 40 --warmup 10 --debug-flags 65536 --cpu-baseline-frames 0 > $O/t.json 2>/dev/null
  python3 -c "
import json
d = json.load(open('$O/t.json')); g = d['gpu_ms']
print('${tag:-product}', '%.4f ms/step setup %.3f tile %.3f' % (d['ms_per_step'], g['msRasterCluster'], g['msRasterChunk']))"
done
