#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03f; mkdir -p $O
for tag in base "" cw7; do
  lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${tag:+_$tag}.so
  CHORDVIS_LIB=$lib python bench.py --steps 40 --warmup 10 --workload subpixel_64m --debug-flags 65536 --cpu-baseline-frames 0 > $O/b64_${tag:-product}.json 2> $O/b64_${tag:-product}.err
  python3 - <<PY
import json
try:
    d = json.load(open("$O/b64_${tag:-product}.json")); g = d["gpu_ms"]
    print("%-8s 64m forced blocks: %.4f ms/step  setup %.1f us  tile %.1f us" % ("${tag:-product}", d["ms_per_step"], g["msRasterCluster"]*1e3, g["msRasterChunk"]*1e3))
except Exception as e:
    print("${tag:-product}", "FAILED", e)
PY
done
python bench.py --workload subpixel_1g --steps 10 --warmup 2 --cpu-baseline-frames 0 > $O/b1g.json 2> $O/b1g.err
python3 -c "
import json
d = json.load(open('$O/b1g.json')); g = d['gpu_ms']
print('product subpixel_1g: %.3f ms/step %.2f Gtri/s setup %.3f tile %.3f cull %.3f' % (d['ms_per_step'], d['value'], g['msRasterCluster'], g['msRasterChunk'], g['msInstanceCulling']))"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $O/pytest.txt | tail -12
