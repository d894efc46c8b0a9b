#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_call7; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q -k "sharded or group or rccl or communicator or marker or bench" > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error\|assert" $OUT/pytest.txt | tail -12
