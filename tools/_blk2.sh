cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/blk2
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/blk2/pytest.txt 2>&1
for wl in street_4k_hzb street_x64_4k_hzb subpixel_64m; do
  python bench.py --workload $wl --cpu-baseline-frames 0 > gpurun_out/blk2/b_${wl}.json 2> gpurun_out/blk2/b_${wl}.err
done
python bench.py --workload subpixel_1g --steps 10 --warmup 2 --cpu-baseline-frames 0 > gpurun_out/blk2/b_subpixel_1g.json 2> gpurun_out/blk2/b_subpixel_1g.err
python bench.py --workload street_x64_4k_hzb --cpu-baseline-frames 0 --debug-flags 65536 > gpurun_out/blk2/b_c4_force.json 2>/dev/null
tail -5 gpurun_out/blk2/pytest.txt
