#!/bin/bash
# Ablation of the setup kernel's pixel-block path on subpixel_64m (every cluster takes it).  Build the variants first, here:
#   python -c "from chord_amd import build; [build.build(defines=d, tag=t) for t, d in (('noras', ('-DBLK_NO_RASTER',)),
#       ('nostore', ('-DBLK_NO_STORE',)), ('noatom', ('-DBLK_NO_ATOMIC', '-DBLK_NO_STORE')),
#       ('nothing', ('-DBLK_NO_ATOMIC', '-DBLK_NO_STORE', '-DBLK_NO_RASTER')))]"
# (the variants render wrong images by construction; the last line, debug flag 2, is the kernel without any emission)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abl
for tag in "" noras nostore noatom nothing; do
  lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${tag:+_$tag}.so
  CHORDVIS_LIB=$lib python bench.py --steps 40 --warmup 6 --workload subpixel_64m --cpu-baseline-frames 0 --debug-flags 65536 > gpurun_out/abl/b_${tag:-product}.json 2> gpurun_out/abl/b_${tag:-product}.err
  python - <<PY
import json
d = json.load(open("gpurun_out/abl/b_${tag:-product}.json")); g = d["gpu_ms"]
print("%-10s %.4f ms/step setup %.1f us tile %.1f us" % ("${tag:-product}", d["ms_per_step"], g["msRasterCluster"]*1e3, g["msRasterChunk"]*1e3))
PY
done
CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis.so python bench.py --steps 40 --warmup 6 --workload subpixel_64m --cpu-baseline-frames 0 --debug-flags 2 > gpurun_out/abl/b_nobin.json 2>/dev/null
python - <<PY
import json
d = json.load(open("gpurun_out/abl/b_nobin.json")); g = d["gpu_ms"]
print("%-10s %.4f ms/step setup %.1f us tile %.1f us" % ("DBG_NO_BIN", d["ms_per_step"], g["msRasterCluster"]*1e3, g["msRasterChunk"]*1e3))
PY
