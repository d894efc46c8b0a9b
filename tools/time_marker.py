"""Times visibility_mark + prepare_shading_tile_param on the config 3 frame (measurement aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chord_amd import lib as L, records as R, scenes
from chord_amd.renderer import VisibilityRenderer
scene, cam = scenes.config3_street()
L.fill_objects(scene, cam)
view, iv = L.make_views(cam)
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
r = VisibilityRenderer(0); r.upload_scene(scene); r.allocate_gbuffer(cam.width, cam.height); r.set_view(view, iv, flags)
r.render_frame(); r.render_frame()
m = r.visibility_mark(); r.sync()
N = 200
t0 = time.perf_counter()
for _ in range(N): m = r.visibility_mark()
r.sync(); t1 = time.perf_counter()
for _ in range(N): t = r.prepare_shading_tile_param(1, m)
r.sync(); t2 = time.perf_counter()
px = cam.width * cam.height
print("visibility_mark %.1f us/call = %.0f GB/s of visibility words; prepare_shading_tile_param %.1f us/call"
      % ((t1 - t0) / N * 1e6, px * 8 / ((t1 - t0) / N) / 1e9, (t2 - t1) / N * 1e6))
tiles, args = r.read_shading_tiles(t)
print("tiles of type 1:", len(tiles), "of", m.markerDim[0] * m.markerDim[1], "args", args.tolist())
