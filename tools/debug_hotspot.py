"""Debug aid: the sharded hotspot case of the GPU suite, per rank and mode, against the oracle."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers as H, orc
from chord_amd import lib as L, records as R, scenes
from chord_amd.renderer import VisibilityRenderer, decode_visibility
from chord_amd.sharding import pick_stripe_rows
ranks = int(os.environ.get("RANKS", "8"))
mode = int(os.environ.get("MODE", "65536"))
sigma = float(os.environ.get("SIGMA", "64"))
scene, cam, view, iv = H.setup_scene(lambda: scenes.config5_subpixel(960, 540, prims=16, patches_per_prim=256, instances=4, hotspot_sigma_px=sigma if sigma > 0 else None))
w, h, flags = cam.width, cam.height, H.ALL_FLAGS
want = orc.frame(scene, view, iv, flags)
stripe = int(os.environ.get("STRIPE", pick_stripe_rows(h, ranks)))
print("ranks", ranks, "stripe", stripe, "mode", mode, "sigma", sigma)
ref = VisibilityRenderer(0); ref.upload_scene(scene); ref.allocate_gbuffer(w, h); ref.set_view(view, iv, flags); ref.set_debug(mode); ref.render_frame()
print("single GPU vs oracle mismatches:", int((ref.read_visibility().reshape(h, w) != want["vis"].reshape(h, w)).sum()))
for rk in range(ranks):
    r = VisibilityRenderer(0)
    r.upload_scene(scene); r.set_shard(stripe, ranks, rk); r.allocate_gbuffer(w, h); r.set_view(view, iv, flags); r.set_debug(mode)
    r.frame_phase_a(); r.frame_phase_b(); r.frame_phase_c()        # own chunk only; compare the rank's OWN rows
    got = r.read_visibility().reshape(h, w)
    own = np.array([((y // stripe) % ranks) == rk for y in range(h)])
    bad = np.argwhere((got != want["vis"].reshape(h, w)) & own[:, None])
    st = r.stats()
    print("rank %d: own rows %d, mismatching own pixels %d, overflow %d, blocks %d records %d culled %d stage0 %d" % (rk, own.sum(), len(bad), st["overflow"], st["pixelBlocks"], st["triangleRecords"], st["countInstanceCulled"], st["countStage0Visible"]))
    for y, x in bad[:6]:
        g, wv = int(got[y, x]), int(want["vis"].reshape(h, w)[y, x])
        print("   (x=%d, y=%d) tile (%d,%d) local (%d,%d): got %#018x want %#018x" % (x, y, x // 64, y // 64, x % 64, y % 64, g, wv))
    if len(bad):
        ys = bad[:, 0]; xs = bad[:, 1]
        print("   rows", sorted(set(ys.tolist()))[:20], "cols range", xs.min(), xs.max())
    r.close()
