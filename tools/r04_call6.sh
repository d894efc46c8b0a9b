#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_call6; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_masked.py -m gpu -x -q -k "masked" 2>&1 | tail -5 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"
for wl in street_4k_hzb street_4k_masked street_4k_masked_twin; do
python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; python -c "
import json; d=json.load(open('$OUT/bench_$wl.json')); g=d['gpu_ms']; print('$wl', d['ms_per_step'], d['value'], 'tris', d['triangles_submitted_per_step'], 'cull %.1f setup %.1f clip %.1f tile %.1f' % (g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3), d['triangle_records_per_step'], d['bin_entries_per_step'])"
done
