#!/bin/bash
# One GPU-box call that produces everything profiles/rNN_* is built from. Usage: tools/round_profile.sh <round tag, e.g. r04>
# Counter passes are separate rocprofv3 runs (no tracing alongside --pmc).  FULL=1 adds the microbenchmarks, the counter
# calibration, the phase clocks and the ablation table (the measuring builds --tag abl -DRASTER_ABLATION=1 / --tag prof
# -DRASTER_PROFILE=1 must exist then).
set -u
R=${1:-r06}
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${R}_pytest.txt 2>&1
grep -a "passed\|failed" gpurun_out/${R}_pytest.txt | tail -3
tools/profile.sh ${R}_c3 > /dev/null
tools/profile.sh ${R}_c4 --workload street_x64_4k_hzb > /dev/null
tools/profile.sh ${R}_c5 --workload subpixel_1g --steps 6 --warmup 2 > /dev/null
# (the counter passes of these two as well: round 4 ran them STATS_ONLY and their traffic files came out empty)
tools/profile.sh ${R}_c5hot --workload subpixel_1g_hotspot --steps 6 --warmup 2 > /dev/null
tools/profile.sh ${R}_masked --workload street_4k_masked > /dev/null
tools/pmc.sh ${R}_sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" > gpurun_out/${R}_sq1.txt 2>&1
tools/pmc.sh ${R}_sq2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" > gpurun_out/${R}_sq2.txt 2>&1
tools/pmc.sh ${R}_sq3 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY" > gpurun_out/${R}_sq3.txt 2>&1
# (round 6: what the waits are made of -- instruction classes in flight, LDS conflicts by kind, instruction fetch, the memory pipes' FIFOs)
tools/pmc.sh ${R}_sq4 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_IFETCH_LEVEL" > gpurun_out/${R}_sq4.txt 2>&1
tools/pmc.sh ${R}_sq5 "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" > gpurun_out/${R}_sq5.txt 2>&1
tools/pmc.sh ${R}_sq6 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_SMEM SQ_BUSY_CU_CYCLES SQ_WAVES" > gpurun_out/${R}_sq6.txt 2>&1
# (the block kernel's SQ counters on the 1/16-size workload with the block kernel forced: same kernel, 5 s of scene generation instead of 70)
tools/pmc.sh ${R}_sq1_c5 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" --workload subpixel_64m --debug-flags 65536 > gpurun_out/${R}_sq1_c5.txt 2>&1
tools/pmc.sh ${R}_sq2_c5 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" --workload subpixel_64m --debug-flags 65536 > gpurun_out/${R}_sq2_c5.txt 2>&1
( cd tools/microbench && { [ -x valu_issue ] || hipcc -O3 --offload-arch=gfx950 -o valu_issue valu_issue.hip; } && timeout 300 ./valu_issue > $GRAFT_REPO_ROOT/gpurun_out/${R}_microbench_valu_issue.txt 2>&1 )
if [ -n "${FULL:-}" ]; then
tools/pmc.sh ${R}_tcc "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" > gpurun_out/${R}_tcc.txt 2>&1
tools/pmc.sh ${R}_tcc_c5 "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" --workload subpixel_1g --steps 4 --warmup 2 > gpurun_out/${R}_tcc_c5.txt 2>&1
cd tools/microbench
for b in ${MICROBENCH-lds_atomics launch_floor atomics}; do
  [ -x $b ] || hipcc -O3 --offload-arch=gfx950 -o $b $b.hip
  timeout 120 ./$b > $GRAFT_REPO_ROOT/gpurun_out/${R}_microbench_$b.txt 2>&1
done
cd $GRAFT_REPO_ROOT
bash tools/ablate.sh ${R}_abl -t abl -f 0,4096,4128,8192,16384,16512,128 > gpurun_out/${R}_ablate_tile.txt 2>&1
CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_prof.so python tools/tile_profile.py hzb > gpurun_out/${R}_tile_profile.txt 2>&1
CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_prof.so python tools/setup_profile.py subpixel_64m 65536 > gpurun_out/${R}_setup_profile_64m.txt 2>&1
fi
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 > gpurun_out/${R}_bench_default_20steps.json 2>/dev/null
python bench.py --workload street_x64_4k_hzb --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c4.json 2>/dev/null
python bench.py --workload subpixel_1g --steps 10 --warmup 2 --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c5.json 2>/dev/null
python bench.py --workload subpixel_1g_hotspot --steps 10 --warmup 2 --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c5hot.json 2>/dev/null
python bench.py --workload subpixel_64m --cpu-baseline-frames 0 --debug-flags 65536 > gpurun_out/${R}_bench_64m.json 2>/dev/null
python bench.py --workload atrium_1080p --no-hzb --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c2.json 2>/dev/null
python bench.py --cull hierarchical --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c3_bvh.json 2>/dev/null
python bench.py --workload street_x64_4k_hzb --cull hierarchical --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c4_bvh.json 2>/dev/null
python bench.py --workload street_4k_masked --cpu-baseline-frames 0 > gpurun_out/${R}_bench_masked.json 2>/dev/null
python bench.py --workload street_4k_masked_twin --cpu-baseline-frames 0 > gpurun_out/${R}_bench_masked_twin.json 2>/dev/null
# where the alpha-tested triangles are scan-converted: the variant libraries (chord_amd/build.py --tag ...) when they exist
for v in msep msep2 mf2; do
  [ -f chord_amd/_build/libchordvis_$v.so ] && CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_$v.so python bench.py --workload street_4k_masked --cpu-baseline-frames 0 > gpurun_out/${R}_bench_masked_$v.json 2>/dev/null
done
# every rank of the sharded frames, one at a time: the sharded group cull (default) at 1 / 2 / 4 / 8 ranks, the replicated cull at 8 beside it
CULL=sharded python tools/shard_time.py subpixel_1g 2>&1 | grep "^ranks" > gpurun_out/${R}_shard_time_c5.txt
CULL=replicated RANKS=8 python tools/shard_time.py subpixel_1g 2>&1 | grep "^ranks" | sed 's/^/[replicated cull] /' >> gpurun_out/${R}_shard_time_c5.txt
PIPELINED=1 RANKS=8 python tools/shard_time.py subpixel_1g 2>&1 | grep "^ranks" | sed 's/^/[pipelined protocol] /' >> gpurun_out/${R}_shard_time_c5.txt
CULL=both RANKS=1,8 python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks" > gpurun_out/${R}_shard_time_c4.txt
PIPELINED=1 RANKS=8 python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks" > gpurun_out/${R}_shard_time_c4_pipelined.txt
RANKS=1,8 python tools/shard_time.py subpixel_1g_hotspot 2>&1 | grep "^ranks" > gpurun_out/${R}_shard_time_c5hot.txt
RANKS=1,8 python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks" | sed 's/^/[static view, for the 1-rank figure beside bench.py] /' >> gpurun_out/${R}_shard_time_c4.txt
( cd /tmp && FRAMES=10 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${R}_rank3 -o r -- python $GRAFT_REPO_ROOT/tools/shard_rank.py subpixel_1g 8 3 > $GRAFT_REPO_ROOT/gpurun_out/${R}_rank3.log 2>&1 ); find gpurun_out/${R}_rank3 -name "*kernel_trace*" -delete
python tools/group_host_time.py 8 2>&1 | grep "ranks on" > gpurun_out/${R}_group_host_time.txt
python tools/shadow_time.py c3 2>&1 | grep "^c3" > gpurun_out/${R}_shadow_time.txt
bash tools/trace.sh ${R}_trace > gpurun_out/${R}_timeline.txt 2>&1
bash tools/trace.sh ${R}_trace_c4 --workload street_x64_4k_hzb > gpurun_out/${R}_timeline_c4.txt 2>&1
# one rank of the 8-rank config-4 frame, launch by launch (product frames: no stamps)
( cd /tmp && export TMPDIR=/tmp && FRAMES=40 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${R}_rank3_c4 -o r -- python $GRAFT_REPO_ROOT/tools/shard_rank.py street_x64_4k_hzb 8 3 > $GRAFT_REPO_ROOT/gpurun_out/${R}_rank3_c4.log 2>&1 )
( echo "frames: tools/shard_rank.py street_x64_4k_hzb 8 3 (rank 3 of 8, default map, collectives skipped, no stamps; 40 frames)"; python tools/timeline.py gpurun_out/${R}_rank3_c4/r_kernel_trace.csv ) > gpurun_out/${R}_timeline_rank3_c4.txt 2>&1
find gpurun_out/${R}_rank3_c4 -name "r_kernel_trace.csv" -delete
find gpurun_out -name "r_kernel_trace.csv" -path "*${R}_trace*" -delete
cd $GRAFT_REPO_ROOT; ls gpurun_out | grep "^${R}" | head -80
