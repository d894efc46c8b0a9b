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


def with_shading_types(scene, types=(1, 37, 100, 64, 127)):
    """A copy of `scene` in which object i has its own material of shading type types[i % len(types)]
    (the procedural scenes use type 1 throughout; the tile marker wants variety)."""
    mats = scene.materials[scene.objects["GLTFMaterialData"]].copy()
    mats["materialType"] = np.asarray(types, dtype=np.uint32)[np.arange(len(mats)) % len(types)]
    objs = scene.objects.copy()
    objs["GLTFMaterialData"] = np.arange(len(objs), dtype=np.uint32)
    out = scene.with_objects(objs, mats)
    out.name = scene.name + "+types"
    return out


def brute_force_marker(scene, vis, w, h, cmds):
    """Per 8x8 pixels the set of shading types present (empty pixel = type 0), straight from the definition."""
    low = (np.asarray(vis, dtype=np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32).reshape(h, w)
    slot = ((low >> 8) & 0xFFFFFF).astype(np.int64) - 1
    obj = np.asarray(cmds["objectId"], dtype=np.int64)
    typ = np.zeros((h, w), dtype=np.uint32)
    hit = low != 0
    typ[hit] = scene.materials["materialType"][scene.objects["GLTFMaterialData"][obj[slot[hit]]]]
    mw, mh = (w + 7) // 8, (h + 7) // 8
    marker = np.zeros((mh, mw, 4), dtype=np.uint32)
    for y in range(h):
        for x in range(w):
            t = int(typ[y, x])
            marker[y // 8, x // 8, t // 32] |= np.uint32(1 << (t % 32))
    return marker


def tiles_with_type(marker, t):
    ys, xs = np.nonzero(marker[:, :, t // 32] & np.uint32(1 << (t % 32)))
    return sorted((int(x) * 8, int(y) * 8) for x, y in zip(xs, ys))


def assert_rank_counts(rank_stats, single):
    """Counts of a sharded frame against the single-GPU frame's.  The instance cull is replicated (identical list, identical
    slots: the visibility ids agree); the occlusion culls of a rank run over the clusters that touch ITS screen tiles only, so
    per stage every rank counts at most the frame's clusters and together they count every one of them at least once
    (a cluster that touches tiles of two ranks is culled -- identically -- by both owners)."""
    assert all(st["overflow"] == 0 for st in rank_stats)
    assert all(st["countInstanceCulled"] == single["countInstanceCulled"] for st in rank_stats)
    for k in ("countStage0Visible", "countStage0Rejected", "countStage1Visible"):
        vals = [st[k] for st in rank_stats]
        assert max(vals) <= single[k], (k, vals, single[k])
        assert sum(vals) >= single[k], (k, vals, single[k])
