#!/bin/bash
# config 3 bench over library variants and debug flags (measurement only)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
run() { # tag flags
  lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${1:+_$1}.so
  CHORDVIS_LIB=$lib python bench.py --steps 200 --warmup 20 --workload ${WL:-street_4k_hzb} --cpu-baseline-frames 0 --debug-flags $2 > $OUT/b_${1:-product}_$2.json 2> $OUT/b_${1:-product}_$2.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/b_${1:-product}_$2.json")); g = d["gpu_ms"]
    print("%-10s flags %-5s %.4f ms/step  setup %.1f us  tile %.1f us" % ("${1:-product}", "$2", d["ms_per_step"], g["msRasterCluster"]*1e3, g["msRasterChunk"]*1e3))
except Exception as e:
    print("${1:-product}", "$2", "FAILED", e)
PY
}
for t in "" "$@"; do run "$t" 0; done
run "" 32; run "" 4096; run "" 4128; run "" 128; run "" 1
