#!/bin/bash
# End of round 5: the suite on the final tree, then the evidence that the second session's kernels changed -- config 3's kernel stats,
# traffic, SQ counters and timeline (the group cull is the four-lane form now), and the bench lines of every workload on one box.
# (tools/collect_round.py r05 afterwards; it leaves the tables this call has no passes for as they are.)
set -u
R=r05
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${R}_pytest.txt 2>&1
grep -a "passed\|failed" gpurun_out/${R}_pytest.txt | tail -3
tools/profile.sh ${R}_c3 > /dev/null
tools/pmc.sh ${R}_sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" > gpurun_out/${R}_sq1.txt 2>&1
tools/pmc.sh ${R}_sq2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" > gpurun_out/${R}_sq2.txt 2>&1
tools/pmc.sh ${R}_sq3 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY" > gpurun_out/${R}_sq3.txt 2>&1
bash tools/trace.sh ${R}_trace > gpurun_out/${R}_timeline.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 > gpurun_out/${R}_bench_default_20steps.json 2>/dev/null
python bench.py --workload street_x64_4k_hzb --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c4.json 2>/dev/null
python bench.py --workload atrium_1080p --no-hzb --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c2.json 2>/dev/null
python bench.py --cull hierarchical --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c3_bvh.json 2>/dev/null
python bench.py --workload street_4k_masked --cpu-baseline-frames 0 > gpurun_out/${R}_bench_masked.json 2>/dev/null
python bench.py --workload street_4k_masked_twin --cpu-baseline-frames 0 > gpurun_out/${R}_bench_masked_twin.json 2>/dev/null
python bench.py --workload subpixel_64m --cpu-baseline-frames 0 --debug-flags 65536 > gpurun_out/${R}_bench_64m.json 2>/dev/null
python bench.py --workload subpixel_1g --steps 10 --warmup 2 --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c5.json 2>/dev/null
python bench.py --workload subpixel_1g_hotspot --steps 10 --warmup 2 --cpu-baseline-frames 0 > gpurun_out/${R}_bench_c5hot.json 2>/dev/null
ls gpurun_out | head -50
