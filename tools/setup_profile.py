"""Per-phase time of the setup kernel summed over waves (measurement aid; debug bit 512).  Needs the profiling build of the
library:  python chord_amd/build.py --tag prof -DRASTER_PROFILE=1;  CHORDVIS_LIB=chord_amd/_build/libchordvis_prof.so python tools/setup_profile.py
  python tools/setup_profile.py [workload] [debug flags]      RANKS=8 RANK=0 LOADS=loads.npy: one rank of a sharded frame (under the map made from LOADS)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from chord_amd import lib as L, records as R
from chord_amd.renderer import VisibilityRenderer
wl = sys.argv[1] if len(sys.argv) > 1 else "street_4k_hzb"
scene, cam = bench.build_workload(wl)
f = np.array(cam.front); f = f / np.linalg.norm(f)
cam_b = cam.moved(tuple(0.5 * f))
va0, _ = L.make_views(cam); vb0, _ = L.make_views(cam_b)
views = [L.make_views(cam, vb0), L.make_views(cam_b, va0)]
objs = [L.fill_objects(scene, cam, cam_b).copy(), L.fill_objects(scene, cam_b, cam).copy()]
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | (0 if wl.startswith("subpixel") else R.FLAG_HZB_CULL)
ranks, rk = int(os.environ.get("RANKS", "1")), int(os.environ.get("RANK", "0"))
r = VisibilityRenderer(0)
if wl.startswith("subpixel_1g"):
    r.set_limits(max_triangle_records=1152 << 20, bin_pool_chunks=1200 << 10, bin_max_chunks_per_tile=2048)
r.upload_scene(scene)
if ranks > 1:
    r.set_shard(ranks, rk)
r.allocate_gbuffer(cam.width, cam.height)
if ranks > 1 and os.environ.get("LOADS"):
    from chord_amd.sharding import tile_layout
    r.set_tile_owners(tile_layout(cam.width, cam.height, ranks, np.load(os.environ["LOADS"]), int(L.lib.chordvis_tile_slot_capacity(cam.width, cam.height, ranks))))
r.set_debug(512 | (int(sys.argv[2]) if len(sys.argv) > 2 else 0))      # e.g. 65536: pixel-block body (its block code counts as "emit")
for i in range(5):
    r.update_objects(objs[i & 1]); r.set_view(views[i & 1][0], views[i & 1][1], flags)
    if ranks > 1:
        r.frame_phase_a(); r.frame_phase_b(); r.frame_phase_c()          # (the replicated cull: the peers' rank masks are not here)
    else:
        r.render_frame()
st = r.stats()
for p in (0, 1):
    t = (C.c_uint64 * 5)(); w = C.c_uint32(0)
    assert L.lib.chordvis_debug_setup_profile(r._ctx, p, t, C.byref(w)) == 0
    us = np.array(list(t), dtype=np.float64) / 100.0
    n = st["countStage0Visible"] if p == 0 else st["countStage1Visible"]
    print("pass %d: %d clusters, %d waves; wave-us sums header/vertex/triangle/reserve/emit = %s; per cluster (us): %s"
          % (p, n, w.value, np.round(us, 0), np.round(us / max(1, n), 2)))
