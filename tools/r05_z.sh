#!/bin/bash
# The clip / large-record binner launched with 64 + 64 workgroups when the pass had <= 256 clip triangles + large records in the last
# finished frame (CLIP_GRID_HINT): parity on the clip / close-up / frame tests, then A/B against -DCLIP_GRID_HINT=0 (--tag nohint).
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "clipper or big_triangles or first_frame or two_pass or moving_camera or kept_tile or ground_level or masked" > gpurun_out/r05z_pytest.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05z_pytest.txt | tail -3
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_ms']
print('%-44s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3))"; }
for rep in 1 2; do
for v in hint nohint; do
  if [ $v = nohint ]; then export CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_nohint.so; else unset CHORDVIS_LIB; fi
  python bench.py --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_hzb"
  python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_4k_hzb 20 steps"
  [ $rep = 1 ] && python bench.py --workload street_x64_4k_hzb --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] street_x64_4k_hzb"
  [ $rep = 1 ] && python bench.py --workload atrium_1080p --no-hzb --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[$v] atrium_1080p"
done
done
true
