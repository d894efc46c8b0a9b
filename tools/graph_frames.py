"""Stream launches vs hipGraph replay of the same frames (measurement aid)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from chord_amd import lib as L, records as R
from chord_amd.renderer import VisibilityRenderer
wl = sys.argv[1] if len(sys.argv) > 1 else "street_4k_hzb"
scene, cam = bench.build_workload(wl)
view, iv = L.make_views(cam)
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
r = VisibilityRenderer(0); r.upload_scene(scene); r.allocate_gbuffer(cam.width, cam.height)
r.update_objects(L.fill_objects(scene, cam, cam)); r.set_view(view, iv, flags)
a, b = C.c_float(0), C.c_float(0)
rc = L.lib.chordvis_debug_graph_frames(r._ctx, 200, C.byref(a), C.byref(b))
print("rc", rc, "stream %.4f ms/frame, graph %.4f ms/frame" % (a.value, b.value), L.lib.chordvis_last_error(r._ctx) if rc else "")
