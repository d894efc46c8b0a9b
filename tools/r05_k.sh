#!/bin/bash
# round 5: the clip / large-record binner launch left out of passes that had nothing for it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05k
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "large_records or clipper or close_ups" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05k/pytest.txt; tail -2 gpurun_out/r05k/pytest.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['gpu_ms']
print('%-52s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us launches %s' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3, d.get('kernel_launches')))"; }
for f in 0 1048576; do
  for a in "" "--workload street_x64_4k_hzb" "--workload street_4k_masked" "--workload atrium_1080p --no-hzb"; do
    python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 --debug-flags $f $a 2>/dev/null | line "[debug $f] $a"
  done
done
python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver form (20 steps):', d['ms_per_step'], d['value'], d['warmup'], d['kernel_launches'])"
