"""Compute-side time of EVERY rank of an N-rank sharded frame, measured one rank at a time on one GPU (collectives skipped, so
images are incomplete; valid for workloads whose culling does not depend on the exchanged HZB, i.e. config 5 -- for two-pass
workloads a rank culls against an HZB that only holds its own tiles: an upper bound of its work).  Reports the WORST rank
(what a frame waits for), the mean and the spread; the speed-up column is  one-GPU time / (worst rank + COLLECTIVES_MS).

The tile map: MAP=default (compact regions of equal area), MAP=balanced (chordvis_tile_layout from the loads of a frame rendered
under the default map -- what chordvis_rebalance installs on every rank after the end-of-frame exchange), MAP=both (default).

The group cull is the SHARDED one by default (a rank tests its share of the group instances; the peers' chunks of the rank-mask
exchange are filled once, outside the clock, by chordvis_debug_fill_cull_exchange -- the view is static); CULL=replicated measures the
round-4 form (every rank tests every group).

WHICH FRAMES: a rank's ms/frame is the wall time of n frames rendered WITHOUT event stamps (chordvis_enable_timers(0)) between two
stream syncs -- product frames.  The phase breakdown printed behind it (cull / setup / tile ...) comes from a SEPARATE pass of stamped
frames right after (every frame stamped: each of its ~20 event records is a barrier packet that keeps the next kernel from being
dispatched under the one in front, a sub-millisecond frame grows by 50-90 us -- round 5 timed those frames and reported them as the
rank's time); `stamped` is that pass's own ms/frame, for the difference.

  python tools/shard_time.py [workload]      RANKS=1,2,4,8  MAP=both  PIPELINED=1  COLLECTIVES_MS=0.3  CULL=sharded|replicated  FRAMES=20
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from chord_amd import lib as L, records as R
from chord_amd.renderer import VisibilityRenderer
from chord_amd.sharding import tile_layout
wl = sys.argv[1] if len(sys.argv) > 1 else "subpixel_64m"
scene, cam = bench.build_workload(wl)
view, iv = L.make_views(cam)
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | (0 if wl.startswith("subpixel") else R.FLAG_HZB_CULL)
objs = L.fill_objects(scene, cam, cam)
coll = float(os.environ.get("COLLECTIVES_MS", "0.3"))
# (the balanced map is made from the loads of a frame under the default map: MAP=balanced measures that frame too and prints only the balanced one)
maps = ["default", "balanced"] if os.environ.get("MAP", "both") in ("both", "balanced") else [os.environ.get("MAP")]
only_balanced = os.environ.get("MAP") == "balanced"
r = VisibilityRenderer(0)
if wl.startswith("subpixel_1g"):
    r.set_limits(max_triangle_records=1152 << 20, bin_pool_chunks=1200 << 10, bin_max_chunks_per_tile=2048)
r.upload_scene(scene)
single = None


cull_modes = {"sharded": [True], "replicated": [False], "both": [False, True]}[os.environ.get("CULL", "sharded")]
sharded_cull = cull_modes[0]


def frame(ranks):
    if ranks == 1:
        r.render_frame()
        return
    if os.environ.get("PIPELINED") == "1":
        L.lib.chordvis_swap_visibility(r._ctx)
    if sharded_cull and ranks <= 8:
        r.frame_phase_cull()                             # (the peers' chunks are in the buffer already)
    if os.environ.get("PIPELINED") == "1":
        # the pipelined protocol's critical path: no visibility gather, no row-major copy
        r.frame_phase_a(); r.frame_phase_b()
        assert L.lib.chordvis_frame_phase_c_finish(r._ctx) == 0
    else:
        r.frame_phase_a(); r.frame_phase_b(); r.frame_phase_c()


for ranks, sharded_cull in [(int(x), m) for x in os.environ.get("RANKS", "1,2,4,8").split(",") for m in (cull_modes if int(x) > 1 else cull_modes[:1])]:
    loads = None
    for which in (maps if ranks > 1 else ["-"]):
        per_rank, per_rank_stamped, stamps0, owners, detail = [], [], None, None, []
        if which == "balanced":
            owners = tile_layout(cam.width, cam.height, ranks, loads, int(L.lib.chordvis_tile_slot_capacity(cam.width, cam.height, ranks)))
        acc = None
        for rk in range(ranks):
            r.set_shard(ranks, rk)
            r.allocate_gbuffer(cam.width, cam.height)
            if ranks > 1:
                r.set_tile_owners(owners)                   # (None: the default map)
            r.reset_history()
            r.update_objects(objs); r.set_view(view, iv, flags)
            if ranks > 1 and ranks <= 8 and sharded_cull:
                r.debug_fill_cull_exchange()
            n = int(os.environ.get("FRAMES", "20"))
            r.enable_timers(0)                              # the clocked frames carry no event record
            for _ in range(4):
                frame(ranks)
            r.sync()
            t0 = time.perf_counter()
            for _ in range(n):
                frame(ranks)
            r.sync()
            per_rank.append((time.perf_counter() - t0) / n * 1e3)
            r.enable_timers(2)                              # a separate pass, every frame stamped: the phase breakdown only
            frame(ranks)
            r.sync(); r.stats()
            t0 = time.perf_counter()
            for _ in range(n):
                frame(ranks)
            r.sync()
            per_rank_stamped.append((time.perf_counter() - t0) / n * 1e3)
            st = r.stats()
            assert st["overflow"] == 0, "work lists overflowed"
            if rk == 0:
                stamps0 = st
            detail.append("%.2fM/%.2f/%.2f" % (st["countStage0Visible"] / 1e6, st["msRasterCluster"], st["msRasterChunk"]))
            if ranks > 1:
                # this rank's own tiles' loads (the other slots of its exchange buffer were never gathered: zero)
                mine = r.read_tile_loads().astype(np.int64) * (r.tile_owners() == rk)
                acc = mine if acc is None else acc + mine
        if ranks > 1 and which == "default":
            loads = acc.astype(np.uint32)
            if os.environ.get("DUMP"):
                np.save(os.path.join(os.environ["DUMP"], "loads_%s_%d.npy" % (wl, ranks)), loads)
        worst, mean = max(per_rank), sum(per_rank) / len(per_rank)
        if ranks == 1:
            single = worst
        st = stamps0
        extra = ""
        if ranks > 1:
            own = r.tile_owners()
            per = np.bincount(own, weights=acc.astype(np.float64), minlength=ranks)
            extra = "; entries max/mean %.3f, tiles per rank %d..%d; per rank clusters / setup ms / tile ms: %s" % (
                per.max() / max(per.mean(), 1.0), np.bincount(own, minlength=ranks).min(), np.bincount(own, minlength=ranks).max(), " ".join(detail))
        if only_balanced and which == "default":
            continue
        print("ranks %d (map %s): worst rank %.3f ms/frame unstamped (%d frames; the stamped pass: worst %.3f), mean %.3f, max/mean %.3f%s%s; per rank %s;  rank-0 GPU stamps of the stamped pass (ms, each interval incl. its record's cost): cull %.3f (%s) stage0 %.3f hzb0 %.3f stage1 %.3f hzbFinal %.3f | setup %.3f clip+order %.3f tile %.3f; launches per frame %d"
              % (ranks, which, worst, n, max(per_rank_stamped), mean, worst / mean,
                 ("; speed-up with %.2f ms of collectives: %.2fx" % (coll, single / (worst + coll))) if single and ranks > 1 else "", extra,
                 " ".join("%.3f" % v for v in per_rank), st["msInstanceCulling"], "sharded" if (sharded_cull and 1 < ranks <= 8) else "replicated", st["msStage0"], st["msHzbStage0"], st["msStage1"], st["msHzbFinal"],
                 st["msRasterCluster"], st["msRasterClip"], st["msRasterChunk"], st["kernelLaunches"]), flush=True)
r.close()
