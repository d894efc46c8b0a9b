for f in 0 1 2; do echo "== debug $f"; python bench.py --steps 40 --warmup 4 --cpu-baseline-frames 0 --debug-flags $f --no-hzb 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_ms'])"; done
