"""Work-list occupancy of one frame of a workload (measurement aid)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from chord_amd import lib as L, records as R
from chord_amd.renderer import VisibilityRenderer
wl = sys.argv[1] if len(sys.argv) > 1 else "street_x64_4k_hzb"
scene, cam = bench.build_workload(wl)
view, iv = L.make_views(cam)
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
r = VisibilityRenderer(0); r.upload_scene(scene); r.allocate_gbuffer(cam.width, cam.height)
r.update_objects(L.fill_objects(scene, cam, cam)); r.set_view(view, iv, flags)
for i in range(2):
    r.render_frame()
    st = L.Stats(); rc = L.lib.chordvis_stats(r._ctx, C.byref(st)); d = st.as_dict()
    print("frame", i, "rc", rc, {k: d[k] for k in ("overflow", "countInstanceCulled", "countStage0Visible", "countStage0Rejected", "countStage1Visible", "trianglesSubmitted", "triangleRecords", "binEntries", "tilesTouched")})
    tx, ty = (cam.width + 63) // 64, (cam.height + 63) // 64
    for p in (0, 1):
        ticks = np.zeros(tx * ty * 9, np.uint64); cnt = np.zeros(tx * ty, np.uint32)
        L.lib.chordvis_debug_tile_profile(r._ctx, p, ticks.ctypes.data, cnt.ctypes.data, tx * ty * 9)
        print("   pass", p, "bin max", int(cnt.max()), "sum", int(cnt.sum()), "tiles>8192:", int((cnt > 8192).sum()), "tiles>16384:", int((cnt > 16384).sum()))
