#!/bin/bash
# The whole GPU suite on the tree with the finer cut and the new empty-frame test; the tile kernel's phase clocks and the stage
# ablation of round 5's kernels (the committed ones are round 3's).
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05p_pytest.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05p_pytest.txt | tail -3
CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_prof.so timeout 300 python tools/tile_profile.py hzb > gpurun_out/r05p_tile_profile.txt 2>&1
WL=street_x64_4k_hzb CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_prof.so timeout 300 python tools/tile_profile.py hzb > gpurun_out/r05p_tile_profile_c4.txt 2>&1
bash tools/ablate.sh r05p_abl -t abl -f 0,4096,4128,8192,16384,16512,128 > gpurun_out/r05p_ablate_tile.txt 2>&1
tail -12 gpurun_out/r05p_ablate_tile.txt
