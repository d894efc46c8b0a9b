#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03n; mkdir -p $O
nproc
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5" --durations=8 2>&1 | tail -14
python bench.py > $O/default.json 2> $O/default.err
python3 -c "
import json
d = json.load(open('$O/default.json')); print(d['ms_per_step'], d['value'], json.dumps(d['cpu_baseline'], indent=1))"
