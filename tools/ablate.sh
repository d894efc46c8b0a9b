#!/bin/bash
# ONE ablation / A-B driver (replaces ablate.sh, ablate2.sh, ablate3.sh, ablate_blocks.sh, ab_variants.sh of rounds 1-2).
#   tools/ablate.sh OUTDIR [-w workload[,workload..]] [-f flags[,flags..]] [-t tag[,tag..]] [-s steps] [-- extra bench args]
# Runs bench.py for every (library variant, workload, debug-flag set) and prints one line each.
#   -w  workloads (default street_4k_hzb); -f  chordvis_set_debug switch sets (default 0; the switches void parity, HZB is turned
#       off for them so that culling does not depend on the pixels); -t  variant libraries built with
#       `python chord_amd/build.py --tag NAME -D...` ("" = the product build, always included first)
# The ablation switches exist only in a library built with -DRASTER_ABLATION=1 (the product library refuses them):
#   python chord_amd/build.py --tag abl -DRASTER_ABLATION=1   and then   -t abl   (the product build's line is skipped for such flags)
# Stage-by-stage ablation of the tile kernel:   tools/ablate.sh abl -t abl -f 0,4096,4128,12320,28704,28832
# Setup kernel without emission on dense geometry:               tools/ablate.sh blk -w subpixel_64m -f 65536,65538
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
WLS=street_4k_hzb; FLAGS=0; TAGS=""; STEPS=200; EXTRA=""
while [ $# -gt 0 ]; do
  case "$1" in
    -w) WLS=$2; shift 2;; -f) FLAGS=$2; shift 2;; -t) TAGS=$2; shift 2;; -s) STEPS=$2; shift 2;;
    --) shift; EXTRA="$*"; break;; *) echo "unknown argument $1"; exit 2;;
  esac
done
for tag in "" ${TAGS//,/ }; do
  lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${tag:+_$tag}.so
  for wl in ${WLS//,/ }; do
    for f in ${FLAGS//,/ }; do
      [ -z "$tag" ] && [ $(( f & 32755 )) -ne 0 ] && continue       # (the product build refuses measurement switches)
      nohzb=""; [ "$f" != "0" ] && [ "$f" != "65536" ] && [ "$f" != "32768" ] && nohzb="--no-hzb"
      n=$OUT/b_${tag:-product}_${wl}_$f
      CHORDVIS_LIB=$lib python bench.py --steps $STEPS --warmup 20 --workload $wl --cpu-baseline-frames 0 --debug-flags $f $nohzb $EXTRA > $n.json 2> $n.err
      python3 - <<PY
import json
try:
    d = json.load(open("$n.json")); g = d["gpu_ms"]
    print("%-10s %-20s flags %-6s %.4f ms/step %8.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us  entries %d" % ("${tag:-product}", "$wl", "$f", d["ms_per_step"], d["value"], g["msInstanceCulling"]*1e3, g["msRasterCluster"]*1e3, g["msRasterClip"]*1e3, g["msRasterChunk"]*1e3, d["bin_entries_per_step"]))
except Exception as e:
    print("${tag:-product}", "$wl", "$f", "FAILED", e)
PY
    done
  done
done
