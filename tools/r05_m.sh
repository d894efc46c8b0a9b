#!/bin/bash
# Re-entry check of the restored tree: the GPU suite, smoke, and the bench lines of the three headline workloads.
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05m_pytest.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05m_pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/r05m_bench_default_20.json 2> gpurun_out/r05m_bench_default_20.err
python bench.py --steps 200 --warmup 20 --cpu-baseline-frames 0 > gpurun_out/r05m_bench_c3.json 2>/dev/null
python bench.py --workload street_x64_4k_hzb --steps 200 --cpu-baseline-frames 0 > gpurun_out/r05m_bench_c4.json 2>/dev/null
python bench.py --workload street_4k_masked --steps 200 --cpu-baseline-frames 0 > gpurun_out/r05m_bench_masked.json 2>/dev/null
for f in gpurun_out/r05m_bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); g=d.get('gpu_ms',{})
print('%-44s %.4f ms/step %.3f Gtri/s  frac %s' % (sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac')))
PY
done
