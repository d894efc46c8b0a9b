for tiny in 16 32 64; do for rpu in 1 2 4; do f=$(( (tiny<<8) | (rpu<<16) )); python bench.py --steps 100 --warmup 6 --cpu-baseline-frames 0 --debug-flags $f 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('tiny',$tiny,'rpu',$rpu, d['value'], d['ms_per_step'], d['gpu_ms']['msRasterChunk'])"; done; done
