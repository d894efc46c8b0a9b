"""oracle/oracle.c against tests/spec_np.py -- an independent restatement of SURVEY Appendix A (A1-A4) in vectorised
numpy -- bit for bit, on the digest scenes and on ortho views.  Two restatements of the same HLSL by different routes:
what both got wrong the same way stays invisible, anything else does not."""
import numpy as np
import pytest

import helpers as H
import orc
import spec_np as S
from chord_amd import lib as L, records as R
from chord_amd import scenes

CASES = [
    ("small", lambda: scenes.small_test_scene(320, 200, seed=7), H.ALL_FLAGS),
    ("small_nocone", lambda: scenes.small_test_scene(200, 120, seed=5), R.FLAG_FRUSTUM_CULL),
    ("masked", lambda: scenes.masked_test_scene(320, 200), H.ALL_FLAGS),
    ("atrium", lambda: scenes.config2_atrium(480, 270), R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL),
    ("street", lambda: scenes.config3_street(640, 360), H.ALL_FLAGS),
]


@pytest.mark.parametrize("name,builder,flags", CASES, ids=[c[0] for c in CASES])
def test_cull_and_hzb_stages_agree_with_the_numpy_restatement(name, builder, flags):
    scene, cam0 = builder()
    f = np.array(cam0.front, dtype=np.float64); f /= np.linalg.norm(f)
    for back in (0.0, 7.0, 55.0):
        cam = cam0.moved(tuple(-back * f))
        L.fill_objects(scene, cam)
        view, iv = L.make_views(cam)
        got = S.instance_culling(scene, view, iv, flags)
        want = orc.instance_culling(scene, view, iv, flags)
        assert len(got) == len(want) and all(np.array_equal(got[k], want[k]) for k in ("objectId", "meshletId", "slot")), \
            "%s at -%g m: A1/A2 command lists differ (%d vs %d)" % (name, back, len(got), len(want))
    # A4 on the frame's depth, A3 for both phases against it
    L.fill_objects(scene, cam0)
    view, iv = L.make_views(cam0)
    w, h = cam0.width, cam0.height
    fr = orc.frame(scene, view, iv, flags)
    depth = (fr["vis"] >> np.uint64(32)).astype(np.uint32).view(np.float32)
    dims, offs, levels = S.hzb_build(depth, w, h, want_max=True)
    desc = orc.hzb_desc(w, h)
    assert [desc.mip_dims(l) for l in range(desc.mipCount)] == dims and [int(desc.mipOffset[l]) for l in range(desc.mipCount)] == offs[:-1].tolist()
    for l, (mn, mx) in enumerate(levels):
        mw, mh = dims[l]
        vh, vw = mn.shape
        assert (vw, vh) == desc.valid_dims(l)
        assert np.array_equal(fr["hzb_min"][offs[l]: offs[l] + mw * mh].reshape(mh, mw)[:vh, :vw], mn), "%s: min mip %d" % (name, l)
        assert np.array_equal(fr["hzb_max"][offs[l]: offs[l] + mw * mh].reshape(mh, mw)[:vh, :vw], mx), "%s: max mip %d" % (name, l)
    cmds = orc.instance_culling(scene, view, iv, flags)
    mins = [lv[0] for lv in levels]
    for phase in (0, 1):
        vis_o, rej_o = orc.hzb_culling(scene, view, flags | R.FLAG_HZB_CULL, phase, desc, fr["hzb_min"], cmds)
        vis_s = S.hzb_visible(scene, view, cmds, phase, mins, dims)
        assert np.array_equal(cmds["slot"][vis_s], vis_o["slot"]), "%s: A3 phase %d" % (name, phase)
        assert vis_s.sum() < len(cmds)                                         # (the test is not vacuous: something is occluded)


def test_object_cull_of_orthographic_views_agrees():
    scene, cam = scenes.masked_test_scene(320, 200)
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    cfg = R.default_cascade_config(cascadeCount=4, realtimeCascadeCount=2, cascadeDim=512, cascadeEndDistance=10.0, farCascadeEndDistance=40.0)
    views = L.cascade_setup(cfg, view, iv, (0.35, -1.0, 0.25))
    for k in range(4):
        got = S.instance_culling(scene, view, views[k:k + 1], H.ALL_FLAGS)
        want = orc.instance_culling(scene, view, views[k:k + 1], H.ALL_FLAGS)
        assert len(want) > 0 and len(got) == len(want) and np.array_equal(got["meshletId"], want["meshletId"]) and np.array_equal(got["objectId"], want["objectId"])
    # the first cascade is small: it must actually cull something the last one keeps
    assert len(orc.instance_culling(scene, view, views[0:1], H.ALL_FLAGS)) < len(orc.instance_culling(scene, view, views[3:4], H.ALL_FLAGS))


@pytest.mark.parametrize("name,builder,dim,last", [("small", lambda: scenes.small_test_scene(320, 200, seed=7), 256, False),
                                                   ("street", lambda: scenes.config3_street(640, 360), 256, False),
                                                   ("masked", lambda: scenes.masked_test_scene(320, 200), 256, True)],
                         ids=["small", "street", "masked_last_frame"])
def test_generic_hzb_cull_of_cascade_views_agrees(name, builder, dim, last):
    """hzb_culling_generic.hlsl (2x2 taps, extentScale, the culling view's own matrix and camera offset) for shadow cascades:
    the oracle's survivors against the numpy restatement, on the HZB of a cascade's depth image."""
    scene, cam = builder()
    L.fill_objects(scene, cam, cam.moved((0.3, 0.0, -0.2)))
    view, iv = L.make_views(cam)
    cfg = R.default_cascade_config(cascadeCount=3, realtimeCascadeCount=2, cascadeDim=dim, cascadeEndDistance=14.0, farCascadeEndDistance=40.0)
    views = L.cascade_setup(cfg, view, iv, (0.35, -1.0, 0.25))
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
    campos = np.frombuffer(iv["cameraWorldPos"][0].tobytes(), dtype=np.float64)[:3]
    desc = orc.hzb_desc(dim, dim)
    rejected = 0
    for k_hzb, k_list in ((2, 1), (1, 0), (1, 1), (2, 2)):               # cull cascade k_list's list against cascade k_hzb's depth
        cmds_h = orc.instance_culling(scene, view, views[k_hzb:k_hzb + 1], flags)
        depth, _ = orc.raster_depth(scene, views[k_hzb:k_hzb + 1], cmds_h, dim, dim)
        words = depth.view(np.uint32).astype(np.uint64) << np.uint64(32)
        _, hmin, _, _ = orc.hzb_build(words, dim, dim)
        dims, offs, levels = S.hzb_build(depth, dim, dim)
        cmds = orc.instance_culling(scene, view, views[k_list:k_list + 1], flags)
        kept = orc.hzb_culling_generic(scene, views[k_hzb:k_hzb + 1], campos, flags, 1.5, last, desc, hmin, cmds)
        keep = S.hzb_visible_generic(scene, views[k_hzb:k_hzb + 1], campos, cmds, 1.5, last, [lv[0] for lv in levels])
        assert np.array_equal(cmds["slot"][keep], kept["slot"]), "%s: cascade %d list against cascade %d HZB" % (name, k_list, k_hzb)
        rejected += int((~keep).sum())
    assert rejected > 0                                                   # (not vacuous: something is occluded in light space)
