"""Time of chordvis_render_shadow on the bench scene: BASELINE config 3's street, the reference's default
cascade configuration (8 cascades of 2048^2, 3 realtime; render_helper.h:467-483), sun from the upper left.  Prints ms per
call for the first tick (every cascade rendered) and for steady-state ticks (3 realtime + 1 cached cascade per tick)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chord_amd import lib as L, records as R, scenes
from chord_amd.renderer import VisibilityRenderer

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
scene, cam = scenes.config3_street() if wl == "c3" else scenes.config4_street_x64()
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
r = VisibilityRenderer(0)
r.upload_scene(scene); r.allocate_gbuffer(cam.width, cam.height)
view, iv = L.make_views(cam)
r.update_objects(L.fill_objects(scene, cam).copy()); r.set_view(view, iv, flags)
r.render_frame(); r.sync()
for dim in (2048, 4096):
    cfg = R.default_cascade_config(cascadeDim=dim, shadowBiasConst=-1.25, shadowBiasSlope=-1.75)
    light = (0.35, -0.8, 0.45)
    t0 = time.perf_counter(); _, _, mask = r.render_shadow(cfg, light, 0); r.sync(); t1 = time.perf_counter()
    print("%s dim %d: first tick (mask %s) %.3f ms (includes allocating the child context)" % (wl, dim, bin(mask), (t1 - t0) * 1e3))
    for hz in (True, False):
        n = 40
        r.sync(); t0 = time.perf_counter()
        for tick in range(1, n + 1):
            _, _, mask = r.render_shadow(cfg, light, tick, hzb_culling=hz)
        r.sync(); t1 = time.perf_counter()
        st = r.depth_view_stats()
        print("%s dim %d: steady tick, hzb culling %s: %.3f ms per call (4 cascades), last cascade: %d triangles submitted, %d bin entries" %
              (wl, dim, hz, (t1 - t0) / n * 1e3, st["trianglesSubmitted"], st["binEntries"]))
r.close()
