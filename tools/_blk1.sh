cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/blk1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/blk1/pytest.txt 2>&1
for wl in street_4k_hzb street_x64_4k_hzb subpixel_64m subpixel_1g; do
  for f in 0 32768; do
    python bench.py --workload $wl --steps 60 --warmup 6 --cpu-baseline-frames 0 --debug-flags $f > gpurun_out/blk1/b_${wl}_$f.json 2> gpurun_out/blk1/b_${wl}_$f.err
  done
done
tail -5 gpurun_out/blk1/pytest.txt
