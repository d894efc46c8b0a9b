#!/bin/bash
# Second run of the finer cut (wave-level sums in the schedule kernel; sharded frames only by default): the rank-by-rank times
# of the 8-rank frames against CHORDVIS_TILE_SLOTS=0, the one-GPU frames as a check that nothing moved.
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q -k "shard or rank or group" > gpurun_out/r05o_pytest.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05o_pytest.txt | tail -3
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_ms']
print('%-40s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3))"; }
python bench.py --workload atrium_1080p --no-hzb --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[default] atrium_1080p"
python bench.py --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[default] street_4k_hzb"
CHORDVIS_TILE_SLOTS=512 python bench.py --workload atrium_1080p --no-hzb --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[slots 512] atrium_1080p"
CHORDVIS_TILE_SLOTS=512 python bench.py --steps 200 --cpu-baseline-frames 0 2>/dev/null | line "[slots 512] street_4k_hzb"
for s in 0 default; do
  if [ $s = default ]; then unset CHORDVIS_TILE_SLOTS; else export CHORDVIS_TILE_SLOTS=$s; fi
  MAP=both RANKS=8 python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks" | sed "s/^/[slots $s] c4 /"
  PIPELINED=1 MAP=balanced RANKS=8 python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks" | sed "s/^/[slots $s] c4 pipelined /"
done
