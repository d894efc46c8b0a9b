"""Stage clocks of frame_cull_fused_kernel (profile build: python chord_amd/build.py --tag prof -DRASTER_PROFILE=1;
CHORDVIS_LIB=chord_amd/_build/libchordvis_prof.so python tools/fused_cull_profile.py [workload]).  Per workgroup of the last frame:
prefetch + zeroing + object pass | HZB tail into LDS | tests (group, meshlet, HZB) | block scan | publish + look-back | list writes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chord_amd import lib as L, records as R
from chord_amd.renderer import VisibilityRenderer
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "street_4k_hzb"
scene, cam = bench.build_workload(wl)
f = np.array(cam.front); f = f / np.linalg.norm(f)
cam_b = cam.moved(tuple(0.5 * f))
va0, _ = L.make_views(cam); vb0, _ = L.make_views(cam_b)
views = [L.make_views(cam, vb0), L.make_views(cam_b, va0)]
objs = [L.fill_objects(scene, cam, cam_b).copy(), L.fill_objects(scene, cam_b, cam).copy()]
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
r = VisibilityRenderer(0); r.upload_scene(scene); r.allocate_gbuffer(cam.width, cam.height)
for i in range(41):
    r.update_objects(objs[i & 1]); r.set_view(views[i & 1][0], views[i & 1][1], flags); r.render_frame()
r.sync()
blocks = (sum(1 for _ in range(1)) and 0) or 0
n = 1024
raw = np.zeros(n * 8, np.uint64)
assert L.lib.chordvis_debug_read(r._ctx, 6, 2048 * 8, raw.nbytes, raw.ctypes.data) == 0
t = raw.reshape(n, 8)[:, :7].astype(np.float64) / 100.0          # us
used = t[:, 0] > 0
t = t[used]
d = np.diff(t, axis=1)
names = ["prefetch+zero+objects", "HZB tail -> LDS", "tests (group, meshlet, HZB)", "block scan", "publish + look-back", "list writes"]
print("%s: %d workgroups; stage medians / max (us):" % (wl, len(t)))
for k, nm in enumerate(names):
    print("  %-30s median %5.2f  max %5.2f" % (nm, np.median(d[:, k]), d[:, k].max()))
print("  workgroup life: median %.2f max %.2f; first start to last end %.2f us" % (np.median(t[:, 6] - t[:, 0]), (t[:, 6] - t[:, 0]).max(), t[:, 6].max() - t[:, 0].min()))
r.close()
