#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_call5; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "block or subpixel or config5 or hotspot or golden" 2>&1 | tail -5 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"
for wl in subpixel_64m subpixel_1g subpixel_1g_hotspot; do
python bench.py --steps 40 --warmup 6 --cpu-baseline-frames 0 --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; python -c "
import json; d=json.load(open('$OUT/bench_$wl.json')); print('$wl', d['ms_per_step'], d['value'], d['gpu_ms']['msRasterCluster'], d['gpu_ms']['msRasterChunk'])"
done
