#!/bin/bash
# A/B of library variants built with `python chord_amd/build.py --tag NAME -D...`: parity tests on the product build,
# then the three bench workloads on every variant.  usage: tools/ab_variants.sh OUTDIR [tag ...]   ("" = product build)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.txt; tail -3 $OUT/pytest.txt
for tag in "" "$@"; do
  lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${tag:+_$tag}.so
  for wl in street_4k_hzb street_x64_4k_hzb subpixel_64m; do
    steps=200; [ $wl = subpixel_64m ] && steps=40
    CHORDVIS_LIB=$lib python bench.py --steps $steps --warmup 20 --workload $wl --cpu-baseline-frames 0 > $OUT/bench_${tag:-product}_$wl.json 2> $OUT/bench_${tag:-product}_$wl.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${tag:-product}_$wl.json"))
    g = d["gpu_ms"]
    print("%-10s %-18s %.4f ms/step  %.3f Gtri/s  setup %.1f us  tile %.1f us  frame(ev) %.1f us" % ("${tag:-product}", "$wl", d["ms_per_step"], d["value"], g["msRasterCluster"]*1e3, g["msRasterChunk"]*1e3, g["msFrame"]*1e3))
except Exception as e:
    print("${tag:-product}", "$wl", "FAILED", e)
PY
  done
done
