"""Frame with everything culled (camera looking at the sky): the fixed cost of the frame and of the tile kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chord_amd import lib as L, records as R, scenes
from chord_amd.renderer import VisibilityRenderer
scene, cam = scenes.config3_street()
cam = scenes.Camera((-62.0, 12.0, 3.0), (0.0, 1.0, 0.05), cam.width, cam.height)
view, iv = L.make_views(cam)
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
r = VisibilityRenderer(0); r.upload_scene(scene); r.allocate_gbuffer(cam.width, cam.height)
r.update_objects(L.fill_objects(scene, cam, cam)); r.set_view(view, iv, flags)
for _ in range(10): r.render_frame()
r.enable_timers(2, 1)
for _ in range(50): r.render_frame()
st = r.stats()
print({k: round(st[k] * 1e3, 1) for k in ("msFrame", "msInstanceCulling", "msStage0", "msStage1", "msRasterCluster", "msRasterClip", "msRasterChunk", "msHzbStage0", "msHzbFinal")}, st["countInstanceCulled"], st["rasterLaunches"])
