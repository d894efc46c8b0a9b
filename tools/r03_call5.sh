#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03e; mkdir -p $O
python tools/debug_hotspot.py > $O/dbg_blocks.txt 2>&1; tail -40 $O/dbg_blocks.txt
MODE=32768 python tools/debug_hotspot.py > $O/dbg_noblocks.txt 2>&1; tail -12 $O/dbg_noblocks.txt
CHORDVIS_AB_OLD_LIB=1 CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_r02.so python tools/debug_hotspot.py > $O/dbg_r02.txt 2>&1; tail -12 $O/dbg_r02.txt
for tag in "" trim pred trimpred; do
  lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${tag:+_$tag}.so
  for wl in street_4k_hzb street_x64_4k_hzb; do
    CHORDVIS_LIB=$lib python bench.py --steps 200 --warmup 20 --workload $wl --cpu-baseline-frames 0 > $O/b_${tag:-product}_$wl.json 2> $O/b_${tag:-product}_$wl.err
    python3 - <<PY
import json
try:
    d = json.load(open("$O/b_${tag:-product}_$wl.json")); g = d["gpu_ms"]
    print("%-9s %-18s %.4f ms/step %.3f Gtri/s  setup %.1f tile %.1f us" % ("${tag:-product}", "$wl", d["ms_per_step"], d["value"], g["msRasterCluster"]*1e3, g["msRasterChunk"]*1e3))
except Exception as e:
    print("${tag:-product}", "$wl", "FAILED", e)
PY
  done
done
CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_trimpred.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not config5 and not sharded and not x64" 2>&1 | tail -4
