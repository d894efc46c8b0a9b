#!/bin/bash
# ONE script for the GPU-box calls of a round (replaces round 5's 27 one-shot tools/r05_*.sh):
#   gpurun --timeout S -- 'bash tools/call.sh TAG step [step ...]'
# Every step writes under gpurun_out/TAG/ and prints a short summary.  Steps (arguments behind a colon, comma-separated):
#   suite[:EXPR]        pytest -m gpu (EXPR: a -k expression, underscores for spaces are NOT translated: quote the call)
#   counters            rocprofv3 -L, the SQ / TCC counter names this box offers
#   micro:NAME[,NAME]   tools/microbench/NAME.hip built and run
#   trace[:WORKLOAD]    tools/trace.sh (product frames, no stamps) + tools/timeline.py
#   bench:ARGS          bench.py ARGS (spaces as '+'), compact line;  bench20 / bench200: the default workload in the driver's / the long form
#   ab:TAGS:WORKLOADS[:STEPS[:REPS]]   product library and the variant libraries TAGS (chord_amd/build.py --tag; a TAG of the form NAME=VALUE is an
#                       environment switch of the product library instead) on WORKLOADS, interleaved
#   rank:WORKLOAD:RANKS:RANK[:ENV+ENV]   kernel trace of ONE rank of a sharded frame (tools/shard_rank.py, no stamps) + tools/timeline.py
#   shard:WORKLOAD[:ENV+ENV]   tools/shard_time.py WORKLOAD with ENV (e.g. RANKS=1,8+PIPELINED=1)
#   pmc:NAME:COUNTERS[:ARGS]   one counter pass of bench.py (tools/pmc.sh), COUNTERS space as '+'
#   tileprof[:WORKLOAD] per-tile phase clocks of the tile kernel (tools/tile_profile.py; needs chord_amd/build.py --tag prof -DRASTER_PROFILE=1)
#   profile:NAME[:ARGS] kernel stats + FETCH_SIZE + WRITE_SIZE passes (tools/profile.sh)
set -u
TAG=$1; shift
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_ms']; rf=d['roofline'] or {}; mp=d.get('moving_path') or {}
    print('%-46s %.4f ms/step %8.3f Gtri/s  cull %.1f setup %.1f clip+order %.1f tile %.1f us  launches %s  dom %.2f us (stamped %.2f, stamp %.2f) frac %.4f  path %s' % ('$1', d['ms_per_step'], d['value'], g['msInstanceCulling']*1e3, g['msRasterCluster']*1e3, g['msRasterClip']*1e3, g['msRasterChunk']*1e3, d.get('kernel_launches'), rf.get('avg_launch_us',0), rf.get('avg_launch_us_stamped',0), rf.get('stamp_cost_us',0), rf.get('frac',0), ('%.4f ms %.3f Gtri/s' % (mp['ms_per_step'], mp['value'])) if mp else '-'))
except Exception as e:
    print('$1', 'FAILED', e)"; }
for step in "$@"; do
  IFS=':' read -r what a1 a2 a3 a4 <<< "$step"
  case "$what" in
    suite)
      if [ -n "${a1:-}" ]; then timeout 2400 python -m pytest tests -m gpu -x -q -k "$a1" > $OUT/pytest.txt 2>&1; else timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; fi
      grep -a "passed\|failed\|error" $OUT/pytest.txt | tail -3; grep -a "^FAILED\|^E  " $OUT/pytest.txt | head -20;;
    counters)
      ( cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "\(SQ\|TCC\|TCP\|GRBM\)_[A-Z0-9_]*" | sort -u > $GRAFT_REPO_ROOT/$OUT/counters.txt ); wc -l $OUT/counters.txt;;
    micro)
      for b in ${a1//,/ }; do ( cd tools/microbench && { [ -x $b ] || hipcc -O3 --offload-arch=gfx950 -o $b $b.hip; } && timeout 300 ./$b > $GRAFT_REPO_ROOT/$OUT/microbench_$b.txt 2>&1 ); tail -40 $OUT/microbench_$b.txt; done;;
    trace)
      bash tools/trace.sh $TAG/trace_${a1:-c3} ${a1:+--workload $a1} > $OUT/timeline_${a1:-c3}.txt 2>&1; cat $OUT/timeline_${a1:-c3}.txt | tail -30
      find gpurun_out/$TAG -name "r_kernel_trace.csv" -size +20M -delete;;
    bench)   python bench.py ${a1//+/ } > $OUT/bench_$(echo "$a1" | tr -c 'a-zA-Z0-9_\n' '_').json 2> $OUT/bench.err; cat $OUT/bench_$(echo "$a1" | tr -c 'a-zA-Z0-9_\n' '_').json | line "$a1";;
    bench20) python bench.py --steps 20 --warmup 5 > $OUT/bench_default_20steps.json 2> $OUT/bench20.err; cat $OUT/bench_default_20steps.json | line "default, 20 steps (driver form)";;
    bench200) python bench.py > $OUT/bench_default.json 2> $OUT/bench200.err; cat $OUT/bench_default.json | line "default, 200 steps";;
    ab)
      for rep in $(seq 1 ${a4:-2}); do for tag in product ${a1//,/ }; do for wl in ${a2//,/ }; do
        lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis.so; ev=_CALL_NOENV=1
        case "$tag" in product) ;; *=*) ev=$tag;; *) lib=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_$tag.so;; esac     # (TAG with '=': an environment switch of the product library)
        env $ev CHORDVIS_LIB=$lib python bench.py --workload ${wl//+/ } --steps ${a3:-200} --cpu-baseline-frames 0 --no-path 2>/dev/null | tee $OUT/ab_${tag}_${wl}_$rep.json | line "[$tag] $wl rep $rep"
      done; done; done;;
    rank)
      ( cd /tmp && export TMPDIR=/tmp && env FRAMES=40 ${a4//+/ } rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/rank_$a1 -o r -- python $GRAFT_REPO_ROOT/tools/shard_rank.py $a1 $a2 $a3 > $GRAFT_REPO_ROOT/$OUT/rank_$a1.log 2>&1 )
      echo "frames: tools/shard_rank.py $a1 $a2 $a3 (rank $a3 of $a2, collectives skipped, no stamps) ${a4:-}" > $OUT/rank_timeline_$a1.txt
      python tools/timeline.py $OUT/rank_$a1/r_kernel_trace.csv >> $OUT/rank_timeline_$a1.txt 2>&1; tail -3 $OUT/rank_$a1.log; cat $OUT/rank_timeline_$a1.txt | tail -32
      find $OUT/rank_$a1 -name "r_kernel_trace.csv" -size +20M -delete;;
    tileprof) CHORDVIS_LIB=$GRAFT_REPO_ROOT/chord_amd/_build/libchordvis_prof.so WL=${a1:-street_4k_hzb} python tools/tile_profile.py hzb > $OUT/tile_profile_${a1:-street_4k_hzb}.txt 2>&1; grep -a "phase sums\|of which\|^pass\|bins \[" $OUT/tile_profile_${a1:-street_4k_hzb}.txt;;
    shard)   env ${a2//+/ } python tools/shard_time.py $a1 2>&1 | grep "^ranks" | tee -a $OUT/shard_time_$a1.txt;;
    pmc)     bash tools/pmc.sh $TAG/pmc_$a1 "${a2//+/ }" ${a3//+/ } 2>&1 | tee $OUT/pmc_$a1.txt | grep "raster_tile\|raster_setup\|hzb_cull\|group_cull" ;;
    profile) bash tools/profile.sh $TAG/prof_$a1 ${a2//+/ } > /dev/null 2>&1; ls $OUT/prof_$a1;;
    *) echo "unknown step $step";;
  esac
done
