#!/bin/bash
# Round 3, GPU call 2: the block kernel (raster_setup_blocks_kernel) -- parity suite, then A/B of occupancy variants against the
# round-2 library on forced-block subpixel_64m, the full config 5, SQ counters and WRITE_SIZE of the new kernel.
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" $O/pytest.txt | tail -12
for tag in r02 "" bw5 bw7 bw8 mvpa; do
  lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis${tag:+_$tag}.so
  CHORDVIS_LIB=$lib python bench.py --steps 40 --warmup 10 --workload subpixel_64m --debug-flags 65536 --cpu-baseline-frames 0 > $O/b64_${tag:-product}.json 2> $O/b64_${tag:-product}.err
  python3 - <<PY
import json
try:
    d = json.load(open("$O/b64_${tag:-product}.json")); g = d["gpu_ms"]
    print("%-8s 64m forced blocks: %.4f ms/step  setup %.1f us  tile %.1f us  blocks %d records %d" % ("${tag:-product}", d["ms_per_step"], g["msRasterCluster"]*1e3, g["msRasterChunk"]*1e3, d["pixel_blocks_per_step"], d["triangle_records_per_step"]))
except Exception as e:
    print("${tag:-product}", "FAILED", e)
PY
done
python bench.py --workload subpixel_1g --steps 10 --warmup 2 --cpu-baseline-frames 0 > $O/b1g_product.json 2> $O/b1g_product.err
python3 -c "
import json
d = json.load(open('$O/b1g_product.json')); g = d['gpu_ms']
print('product subpixel_1g: %.3f ms/step %.2f Gtri/s setup %.3f tile %.3f cull %.3f' % (d['ms_per_step'], d['value'], g['msRasterCluster'], g['msRasterChunk'], g['msInstanceCulling']))"
tools/pmc.sh r03b/sq_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" --workload subpixel_64m --debug-flags 65536 > $O/sq_a.txt 2>&1
tools/pmc.sh r03b/sq_b "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" --workload subpixel_64m --debug-flags 65536 > $O/sq_b.txt 2>&1
tools/pmc.sh r03b/wr "WRITE_SIZE" --workload subpixel_64m --debug-flags 65536 > $O/wr.txt 2>&1
CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_r02.so tools/pmc.sh r03b/wr_r02 "WRITE_SIZE" --workload subpixel_64m --debug-flags 65536 > $O/wr_r02.txt 2>&1
grep -h "setup" $O/sq_a.txt $O/sq_b.txt $O/wr.txt $O/wr_r02.txt
python bench.py --cpu-baseline-frames 0 > $O/b_c3.json 2> $O/b_c3.err
python bench.py --workload street_x64_4k_hzb --cpu-baseline-frames 0 > $O/b_c4.json 2> $O/b_c4.err
python3 -c "
import json
for n in ('b_c3', 'b_c4'):
    d = json.load(open('$O/' + n + '.json')); g = d['gpu_ms']
    print(n, '%.4f ms/step %.3f Gtri/s setup %.1f tile %.1f cull %.1f' % (d['ms_per_step'], d['value'], g['msRasterCluster']*1e3, g['msRasterChunk']*1e3, g['msInstanceCulling']*1e3))"
