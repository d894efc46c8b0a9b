"""Shared helpers for the parity tests: scene/view setup and GPU-vs-oracle comparisons."""
import numpy as np

from chord_amd import records as R
from chord_amd import scenes

ALL_FLAGS = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL


def setup_scene(builder, *args, **kw):
    """(scene, camera, view, iv) with object records filled for the camera (static scene, no history)."""
    from chord_amd import lib as L
    scene, cam = builder(*args, **kw)
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    return scene, cam, view, iv


def views_for(cam, last_view=None):
    from chord_amd import lib as L
    return L.make_views(cam, last_view)


def sort_cmds(cmds):
    return np.sort(np.asarray(cmds, dtype=R.DRAW_CMD), order=["slot", "objectId", "meshletId"])


def assert_vis_equal(got, want, w, h, what=""):
    got = np.asarray(got, dtype=np.uint64).reshape(h, w)
    want = np.asarray(want, dtype=np.uint64).reshape(h, w)
    if np.array_equal(got, want):
        return
    bad = np.argwhere(got != want)
    y, x = bad[0]
    raise AssertionError("%s visibility mismatch at %d pixels; first (x=%d, y=%d): got %#018x want %#018x"
                         % (what, len(bad), x, y, int(got[y, x]), int(want[y, x])))
