"""One-off parity check of config 5 at a large fraction of its full size (the full 1.07 G triangles would take the
scalar oracle ten minutes): GPU frame vs the multi-threaded oracle replay, bit for bit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import orc
from chord_amd import scenes, records as R, lib as L
from chord_amd.renderer import VisibilityRenderer
prims = int(sys.argv[1]) if len(sys.argv) > 1 else 256
t = time.time(); scene, cam = scenes.config5_subpixel(3840, 2160, prims=prims); print("scene: %d triangles, %.1f s" % (scene.triangle_count_lod0(), time.time() - t))
L.fill_objects(scene, cam); view, iv = L.make_views(cam)
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL
r = VisibilityRenderer(0)
r.set_limits(max_triangle_records=(1152 << 20) * prims // 1024 + (8 << 20), bin_pool_chunks=(1200 << 10) * prims // 1024 + 32768, bin_max_chunks_per_tile=2048)
r.upload_scene(scene); r.allocate_gbuffer(cam.width, cam.height); r.set_view(view, iv, flags)
r.render_frame(); got = r.read_visibility(); st = r.stats()
print("gpu: submitted %d records %d entries %d overflow %d" % (st["trianglesSubmitted"], st["triangleRecords"], st["binEntries"], st["overflow"]))
tiles = 60 * 34
ticks = np.zeros(tiles * 9, np.uint64); cnt = np.zeros(tiles, np.uint32)
L.lib.chordvis_debug_tile_profile(r._ctx, 0, ticks.ctypes.data, cnt.ctypes.data, tiles * 9)
print("bins: max %d entries, tiles over 131072 (64 slices of more than 2048): %d" % (int(cnt.max()), int((cnt > 131072).sum())))
t = time.time(); want = orc.frame_mt(scene, view, iv, flags, None, threads=32); print("oracle (32 threads): %.1f s" % (time.time() - t))
bad = int((got != want["vis"]).sum())
print("visibility words that differ: %d of %d; triangles submitted equal: %s" % (bad, len(got), st["trianglesSubmitted"] == want["triangles_submitted"]))
sys.exit(1 if bad else 0)
