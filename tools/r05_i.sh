#!/bin/bash
# round 5: the tile kernel at three workgroups per CU (256-entry batches, 1 536-unit rounds: 53 KB of LDS; 80 VGPRs with 48 B of scratch)
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['gpu_ms']
print('%-52s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f tile %.1f us' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterChunk']*1e3))"; }
for v in "" t3; do
  lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${v:+_$v}.so
  for a in "" "--workload street_x64_4k_hzb" "--workload street_4k_masked" "--workload subpixel_64m --debug-flags 65536" "--workload atrium_1080p --no-hzb"; do
    CHORDVIS_LIB=$lib python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 $a 2>/dev/null | line "[${v:-product}] $a"
  done
done
CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_t3.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or config2 or config4 or masked or sharded or close_ups or clipper" 2>&1 | tail -2
