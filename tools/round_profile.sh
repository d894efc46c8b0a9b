#!/bin/bash
# One GPU-box call that produces everything profiles/rNN_* is built from. Usage: tools/round_profile.sh <round tag, e.g. r02>
# Counter passes are separate rocprofv3 runs (no tracing alongside --pmc).
set -u
R=${1:-r02}
cd $GRAFT_REPO_ROOT
tools/profile.sh ${R}_c3 > /dev/null
tools/profile.sh ${R}_c4 --workload street_x64_4k_hzb > /dev/null
tools/profile.sh ${R}_c5 --workload subpixel_1g --steps 6 --warmup 2 > /dev/null
tools/pmc.sh ${R}_sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" > gpurun_out/${R}_sq1.txt 2>&1
tools/pmc.sh ${R}_sq2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" > gpurun_out/${R}_sq2.txt 2>&1
tools/pmc.sh ${R}_sq3 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAVE_CYCLES" > gpurun_out/${R}_sq3.txt 2>&1
tools/pmc.sh ${R}_tcc "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" > gpurun_out/${R}_tcc.txt 2>&1
tools/pmc.sh ${R}_tcc_c5 "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" --workload subpixel_1g --steps 4 --warmup 2 > gpurun_out/${R}_tcc_c5.txt 2>&1
cd tools/microbench
for b in lds_atomics launch_floor atomics; do
  [ -x $b ] || hipcc -O3 --offload-arch=gfx950 -o $b $b.hip
  timeout 120 ./$b > $GRAFT_REPO_ROOT/gpurun_out/${R}_microbench_$b.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
python bench.py --workload street_x64_4k_hzb --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c4.json 2>/dev/null
python bench.py --workload subpixel_1g --steps 10 --warmup 2 --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c5.json 2>/dev/null
python bench.py --workload subpixel_64m --cpu-baseline-frames 0 > gpurun_out/${R}_bench_64m.json 2>/dev/null
python bench.py --cull hierarchical --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c3_bvh.json 2>/dev/null
WL=street_4k_hzb bash tools/ablate3.sh ${R}_abl > gpurun_out/${R}_ablate3.txt 2>&1
python tools/shard_time.py subpixel_1g 2>&1 | grep "^ranks" > gpurun_out/${R}_shard_time_c5.txt
python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks" > gpurun_out/${R}_shard_time_c4.txt
python tools/shadow_time.py c3 2>&1 | grep "^c3" > gpurun_out/${R}_shadow_time.txt
bash tools/trace.sh ${R}_trace > gpurun_out/${R}_timeline.txt 2>&1
ls gpurun_out | head -50
