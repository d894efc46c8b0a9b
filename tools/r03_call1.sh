#!/bin/bash
# Round 3, GPU call 1: diagnostics only (nothing here changes the product).
#   * tile kernel: per-tile profile of config 3 (empty tiles, heaviest tile, phase sums)
#   * setup kernel on sub-pixel geometry: SQ counters + per-cluster phase clocks, launch-bounds 5 variant
#   * LDS bank conflicts of the tile kernel attributed to its stages (ablation flags, HZB off)
#   * WRITE_SIZE / FETCH_SIZE calibration on the store patterns of the raster kernels
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03a; mkdir -p $O
python tools/tile_profile.py hzb > $O/tile_profile_c3.txt 2>&1
python tools/setup_profile.py subpixel_64m > $O/setup_profile_64m.txt 2>&1
tools/pmc.sh r03a/sq_c5_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" --workload subpixel_64m > $O/sq_c5_a.txt 2>&1
tools/pmc.sh r03a/sq_c5_b "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" --workload subpixel_64m > $O/sq_c5_b.txt 2>&1
for f in 0 4096 4128 12320; do
  tools/pmc.sh r03a/bank_$f "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES" --no-hzb --debug-flags $f > $O/bank_$f.txt 2>&1
done
for tag in "" lb5; do
  lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${tag:+_$tag}.so
  CHORDVIS_LIB=$lib python bench.py --steps 40 --warmup 10 --workload subpixel_64m --cpu-baseline-frames 0 > $O/bench_64m_${tag:-product}.json 2> $O/bench_64m_${tag:-product}.err
done
cd tools/microbench
hipcc -O3 --offload-arch=gfx950 -o write_size_calib write_size_calib.hip
./write_size_calib > $O/calib_useful.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/calib_w -o r -- $GRAFT_REPO_ROOT/tools/microbench/write_size_calib > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_f -o r -- $GRAFT_REPO_ROOT/tools/microbench/write_size_calib > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob
for kind in ("calib_w", "calib_f"):
    for p in glob.glob("$O/%s/**/r_counter_collection.csv" % kind, recursive=True) + glob.glob("$O/%s/r_counter_collection.csv" % kind):
        for r in csv.DictReader(open(p)):
            print(kind, r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Counter_Value"])
PY
cat $O/calib_useful.txt
tail -5 $O/sq_c5_a.txt $O/sq_c5_b.txt
tail -3 $O/bank_*.txt
cat $O/setup_profile_64m.txt | tail -3
grep -h '"ms_per_step"' $O/bench_64m_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_per_step'], d['gpu_ms']['msRasterCluster'], d['gpu_ms']['msRasterChunk'])"
tail -25 $O/tile_profile_c3.txt
