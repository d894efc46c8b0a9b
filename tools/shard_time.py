"""Compute-side time of EVERY rank of an N-rank sharded frame, measured one rank at a time on one GPU (collectives skipped, so
images are incomplete; valid for workloads whose culling does not depend on the exchanged HZB, i.e. config 5 -- for two-pass
workloads a rank culls against an HZB that only holds its own stripes: an upper bound of its work).  Reports the WORST rank
(what a frame waits for), the mean and the spread; the speed-up column is  one-GPU time / (worst rank + COLLECTIVES_MS).

  python tools/shard_time.py [workload]      RANKS=1,2,4,8  STRIPE=<rows>  PIPELINED=1  COLLECTIVES_MS=0.3
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from chord_amd import lib as L, records as R
from chord_amd.renderer import VisibilityRenderer
from chord_amd.sharding import pick_stripe_rows
wl = sys.argv[1] if len(sys.argv) > 1 else "subpixel_64m"
scene, cam = bench.build_workload(wl)
view, iv = L.make_views(cam)
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | (0 if wl.startswith("subpixel") else R.FLAG_HZB_CULL)
objs = L.fill_objects(scene, cam, cam)
coll = float(os.environ.get("COLLECTIVES_MS", "0.3"))
r = VisibilityRenderer(0)
if wl.startswith("subpixel_1g"):
    r.set_limits(max_triangle_records=1152 << 20, bin_pool_chunks=1200 << 10, bin_max_chunks_per_tile=2048)
r.upload_scene(scene)
single = None
for ranks in [int(x) for x in os.environ.get("RANKS", "1,2,4,8").split(",")]:
    per_rank, stamps0 = [], None
    stripe = int(os.environ.get("STRIPE", pick_stripe_rows(cam.height, ranks))) if ranks > 1 else 0
    for rk in range(ranks):
        r.set_shard(stripe if ranks > 1 else 64, ranks, rk)
        r.allocate_gbuffer(cam.width, cam.height)
        r.reset_history()
        r.update_objects(objs); r.set_view(view, iv, flags)
        r.enable_timers(2)

        def frame():
            if ranks == 1:
                r.render_frame()
            elif os.environ.get("PIPELINED") == "1":
                # the pipelined protocol's critical path: no visibility gather, no row-major copy, history HZB from the exchange
                L.lib.chordvis_swap_visibility(r._ctx)
                r.frame_phase_a(); r.frame_phase_b()
                assert L.lib.chordvis_frame_phase_c_begin(r._ctx) == 0 and L.lib.chordvis_frame_phase_c_finish(r._ctx) == 0
            else:
                r.frame_phase_a(); r.frame_phase_b(); r.frame_phase_c()
        for _ in range(3):
            frame()
        r.sync()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            frame()
        r.sync()
        per_rank.append((time.perf_counter() - t0) / n * 1e3)
        st = r.stats()
        assert st["overflow"] == 0, "work lists overflowed"
        if rk == 0:
            stamps0 = st
    worst, mean = max(per_rank), sum(per_rank) / len(per_rank)
    if ranks == 1:
        single = worst
    st = stamps0
    print("ranks %d (stripe %s): worst rank %.3f ms/frame, mean %.3f, max/mean %.3f%s; per rank %s;  rank-0 GPU stamps (ms): cull %.3f stage0 %.3f hzb0 %.3f stage1 %.3f hzbFinal %.3f | setup %.3f clip+order %.3f tile %.3f"
          % (ranks, stripe or "-", worst, mean, worst / mean,
             ("; speed-up with %.2f ms of collectives: %.2fx" % (coll, single / (worst + coll))) if single and ranks > 1 else "",
             " ".join("%.3f" % v for v in per_rank), st["msInstanceCulling"], st["msStage0"], st["msHzbStage0"], st["msStage1"], st["msHzbFinal"],
             st["msRasterCluster"], st["msRasterClip"], st["msRasterChunk"]))
r.close()
