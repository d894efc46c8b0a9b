#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03fin; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
grep -a "passed\|failed\|Error\|assert" $O/pytest.txt | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
tools/profile.sh r03fin_c3 > /dev/null
bash tools/trace.sh r03fin_trace > $O/timeline.txt 2>&1; tail -14 $O/timeline.txt
python bench.py > $O/default.json 2> $O/default.err
python3 -c "
import json
d = json.load(open('$O/default.json')); g = d['gpu_ms']
print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['avg_launch_us'], d['cpu_baseline']['value'], d['cpu_baseline']['all_cores']['value'])"
