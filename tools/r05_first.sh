#!/bin/bash
# round 5, first GPU call: the suite, then the sharded cull's effect on every rank's frame
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r05a/pytest.txt
grep -a "passed\|failed\|Error\|error" gpurun_out/r05a/pytest.txt | tail -8
MAP=default CULL=both RANKS=1,8 timeout 900 python tools/shard_time.py subpixel_1g 2>&1 | grep "^ranks\|Error\|error" > gpurun_out/r05a/shard_time_c5.txt
MAP=default CULL=both RANKS=1,8 timeout 600 python tools/shard_time.py street_x64_4k_hzb 2>&1 | grep "^ranks\|Error\|error" > gpurun_out/r05a/shard_time_c4.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05a/bench_default_20.json 2> gpurun_out/r05a/bench_default_20.err
cat gpurun_out/r05a/shard_time_c5.txt gpurun_out/r05a/shard_time_c4.txt | cut -c1-400
python -c "
import json; d=json.load(open('gpurun_out/r05a/bench_default_20.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['warmup'])"
