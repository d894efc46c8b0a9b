#!/bin/bash
# phase clocks of the block kernel: the rank that owns the hot tile vs a rank of the uniform frame
cd $GRAFT_REPO_ROOT
export CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_prof.so
RANKS=8 RANK=0 LOADS=profiles/r04_tile_loads_config5_hotspot.npy python tools/setup_profile.py subpixel_1g_hotspot 2>&1 | grep "^pass 0" | sed 's/^/[hot rank] /'
RANKS=8 RANK=3 python tools/setup_profile.py subpixel_1g 2>&1 | grep "^pass 0" | sed 's/^/[uniform rank] /'
