"""Compute-side time of ONE rank of an N-rank sharded frame, measured on one GPU (collectives skipped, so images are
incomplete; valid for workloads whose culling does not depend on the exchanged HZB, i.e. config 5)."""
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
for ranks in [int(x) for x in os.environ.get("RANKS", "1,2,4,8").split(",")]:
    r = VisibilityRenderer(0)
    if wl == "subpixel_1g":
        share = max(1, ranks // 2) if ranks > 1 else 1
        r.set_limits(max_triangle_records=(1152 << 20) // share, bin_pool_chunks=(1200 << 10) // share, bin_max_chunks_per_tile=2048)
    r.upload_scene(scene)
    if ranks > 1:
        r.set_shard(int(os.environ.get("STRIPE", pick_stripe_rows(cam.height, ranks))), ranks, 0)
    r.allocate_gbuffer(cam.width, cam.height)
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
    ms = (time.perf_counter() - t0) / n * 1e3
    st = r.stats()
    print("ranks %d: rank 0 compute %.3f ms/frame, records %d, overflow %d;  GPU stamps (ms): cull %.3f stage0 %.3f hzb0 %.3f stage1 %.3f hzbFinal %.3f | setup %.3f clip+order %.3f tile %.3f"
          % (ranks, ms, st["triangleRecords"], st["overflow"], st["msInstanceCulling"], st["msStage0"], st["msHzbStage0"], st["msStage1"], st["msHzbFinal"],
             st["msRasterCluster"], st["msRasterClip"], st["msRasterChunk"]))
    r.close()
