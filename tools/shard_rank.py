"""ONE rank of an N-rank sharded frame under a given tile map, for profilers (rocprofv3 --kernel-trace --stats -- python tools/shard_rank.py ...).

  python tools/shard_rank.py WORKLOAD RANKS RANK [LOADS.npy]     LOADS: tile loads for chordvis_tile_layout (default map without)   FRAMES=10
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from chord_amd import lib as L, records as R
from chord_amd.renderer import VisibilityRenderer
from chord_amd.sharding import tile_layout
wl, ranks, rk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
scene, cam = bench.build_workload(wl)
view, iv = L.make_views(cam)
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | (0 if wl.startswith("subpixel") else R.FLAG_HZB_CULL)
objs = L.fill_objects(scene, cam, cam)
r = VisibilityRenderer(0)
if wl.startswith("subpixel_1g"):
    r.set_limits(max_triangle_records=1152 << 20, bin_pool_chunks=1200 << 10, bin_max_chunks_per_tile=2048)
r.upload_scene(scene)
r.set_shard(ranks, rk)
r.allocate_gbuffer(cam.width, cam.height)
if len(sys.argv) > 4:
    loads = np.load(sys.argv[4])
    r.set_tile_owners(tile_layout(cam.width, cam.height, ranks, loads, int(L.lib.chordvis_tile_slot_capacity(cam.width, cam.height, ranks))))
r.update_objects(objs); r.set_view(view, iv, flags)
sharded_cull = ranks > 1 and ranks <= 8 and os.environ.get("CULL", "sharded") != "replicated"
if sharded_cull:
    r.debug_fill_cull_exchange()                         # (the peers' chunks of the rank-mask exchange: the view is static)
# (no event stamps unless STAMPS=1: under a kernel trace the frames are then product frames; the cull / setup / tile split of the printed
# line needs the stamps and reads 0 without them)
r.enable_timers(2 if os.environ.get("STAMPS") == "1" else 0)
n = int(os.environ.get("FRAMES", "10"))
for i in range(3 + n):
    if i == 3:
        r.sync(); t0 = time.perf_counter()
    if ranks == 1:
        r.render_frame()
    else:
        if sharded_cull:
            r.frame_phase_cull()
        r.frame_phase_a(); r.frame_phase_b(); r.frame_phase_c()
r.sync()
st = r.stats()
print("%s rank %d of %d: %.3f ms/frame; cull %.3f setup %.3f tile %.3f; blocks %d bins %d overflow %d" % (
    wl, rk, ranks, (time.perf_counter() - t0) / n * 1e3, st["msInstanceCulling"], st["msRasterCluster"], st["msRasterChunk"], st["pixelBlocks"], st["binEntries"], st["overflow"]))
r.close()
