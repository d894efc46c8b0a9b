#!/bin/bash
# config 3, HZB OFF (single raster pass: culling does not depend on the pixels, so ablation flags do not change the work)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for f in 0 32 4096 4128 8192 16384 128 16512 1024; do
  python bench.py --steps 200 --warmup 20 --workload ${WL:-street_4k_hzb} --no-hzb --cpu-baseline-frames 0 --debug-flags $f > $OUT/b_$f.json 2> $OUT/b_$f.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/b_$f.json")); g = d["gpu_ms"]
    print("flags %-6s %.4f ms/step  cull %.1f setup %.1f clip+order %.1f tile %.1f hzb %.1f us   entries %d" % ("$f", d["ms_per_step"], g["msInstanceCulling"]*1e3, g["msRasterCluster"]*1e3, g["msRasterClip"]*1e3, g["msRasterChunk"]*1e3, g["msHzbFinal"]*1e3, d["bin_entries_per_step"]))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
