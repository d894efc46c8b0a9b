#!/bin/bash
# GPU suite + optional bench lines.  usage: tools/run_gpu_suite.sh OUTDIR [bench arg sets separated by ';']
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $OUT/pytest.txt
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $OUT/pytest.txt | tail -30
IFS=';' read -ra SETS <<< "$*"
i=0
for a in "${SETS[@]}"; do
  [ -z "$a" ] && continue
  python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 $a > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$i.json")); g = d["gpu_ms"]
    print("%-50s %.4f ms/step %.3f Gtri/s  cull %.1f setup %.1f tile %.1f us" % ("$a", d["ms_per_step"], d["value"], g["msInstanceCulling"]*1e3, g["msRasterCluster"]*1e3, g["msRasterChunk"]*1e3))
except Exception as e:
    print("$a", "FAILED", e)
PY
  i=$((i+1))
done
