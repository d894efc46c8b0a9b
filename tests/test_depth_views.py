"""Depth-only views: cascade setup on the host, the oracle's depth pass and generic HZB cull (CPU), and the
GPU passes against them (gpu)."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
import orc
from chord_amd import lib as L, records as R
from chord_amd import scenes

LIGHT = (0.35, -1.0, 0.25)


def _cascades(cam, **kw):
    view, iv = L.make_views(cam)
    cfg = R.default_cascade_config(**kw)
    return cfg, view, iv, L.cascade_setup(cfg, view, iv, LIGHT)


def _m(rec, name):
    return rec[name].reshape(4, 4).T.astype(np.float64)        # glm column-major -> M[r][c]


def test_cascade_views_are_texel_snapped_orthographic_boxes_around_their_split():
    scene, cam = scenes.small_test_scene(320, 200)
    cfg, view, iv, views = _cascades(cam, cascadeCount=5, realtimeCascadeCount=2, cascadeDim=1024, cascadeEndDistance=30.0, farCascadeEndDistance=120.0)
    inv_zfar = _m(view[0], "clipToTranslatedWorldWithZFar_NoJitter")
    near, far = float(view["zNear"][0]), float(view["zFar"][0])
    prev_end = 0.0
    radii = []
    for k, v in enumerate(views):
        vp, inv = _m(v, "translatedWorldToClip"), _m(v, "clipToTranslatedWorld")
        assert np.array_equal(vp[3], [0, 0, 0, 1])                                      # orthographic: isOrthoProjection, base.hlsli:243-246
        assert np.allclose(vp @ inv, np.eye(4), atol=2e-4)
        assert v["renderDimension"].tolist() == [1024.0, 1024.0, 1.0 / 1024, 1.0 / 1024]
        # texel alignment (cascade_setup.hlsl:312-326): the world origin lands on a whole texel
        o = vp @ np.array([0, 0, 0, 1.0]) * 512.0
        assert abs(o[0] - round(o[0])) < 2e-2 and abs(o[1] - round(o[1])) < 2e-2
        # inward planes of the box: its centre is inside all six, and clip-space corners are on them
        centre = (inv @ np.array([0, 0, 0.5, 1.0]))[:3]
        assert all(np.dot(pl[:3], centre) + pl[3] > 0 for pl in v["frustumPlanesRS"][:5])
        radii.append(1.0 / np.linalg.norm(vp[0, :3]))                                 # row 0 = s / R
    assert all(b >= a for a, b in zip(radii, radii[1:]))                                 # farther cascades are at least as wide
    # cache: with a valid cache only the scheduled far cascade is rewritten (isCascadeCacheValid, :8-22)
    marker = views.copy()
    marker["renderDimension"][:, 0] = -1.0
    for tick in range(3):
        out = L.cascade_setup(cfg, view, iv, LIGHT, tick=tick, cache_valid=True, views=marker.copy())
        rewritten = [k for k in range(5) if out["renderDimension"][k, 0] != -1.0]
        assert rewritten == [0, 1, 2 + tick % 3]


def test_sdsm_range_tightens_the_realtime_cascades():
    scene, cam = scenes.small_test_scene(320, 200)
    view, iv = L.make_views(cam)
    cfg = R.default_cascade_config(cascadeCount=4, realtimeCascadeCount=2, cascadeDim=512)
    wide = L.cascade_setup(cfg, view, iv, LIGHT)
    near = float(view["zNear"][0])
    # depths seen: view z from 4 m to 9 m  ->  device depth = zNear / z
    rng = np.array([np.float32(near / 9.0), np.float32(near / 4.0)], dtype=np.float32).view(np.uint32)
    tight = L.cascade_setup(cfg, view, iv, LIGHT, valid_range=rng)
    r_wide = 1.0 / np.linalg.norm(wide["translatedWorldToClip"][:, [0, 4, 8]], axis=1)
    r_tight = 1.0 / np.linalg.norm(tight["translatedWorldToClip"][:, [0, 4, 8]], axis=1)
    assert (r_tight[:2] < r_wide[:2]).all() and np.array_equal(r_tight[2:], r_wide[2:])


def _setup(width=320, height=200, dim=256, **kw):
    scene, cam = scenes.masked_test_scene(width, height)
    L.fill_objects(scene, cam)
    cfg, view, iv, views = _cascades(cam, cascadeCount=3, realtimeCascadeCount=2, cascadeDim=dim, cascadeEndDistance=14.0,
                                     farCascadeEndDistance=40.0, **kw)
    return scene, cam, cfg, view, iv, views


def test_oracle_depth_pass_properties():
    scene, cam, cfg, view, iv, views = _setup()
    dim = int(cfg["cascadeDim"][0])
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
    cmds = orc.instance_culling(scene, view, views[1:2], flags)
    assert len(cmds) > 0
    depth, st = orc.raster_depth(scene, views[1:2], cmds, dim, dim)
    assert st.trianglesBackface == 0                                            # cull mode NONE (mesh_raster.cpp:188-190)
    assert (depth >= 0).all() and (depth <= 1).all() and (depth > 0).any()
    assert st.fragmentsClipped > 0                                              # the masked bucket alpha-tests in depth passes too
    # the depth image equals the depth half of a two-sided cluster pass where no clamping happens
    two = scene.materials.copy()
    two["bTwoSided"] = 1
    vis, _ = orc.raster(scene.with_objects(None, two), views[1:2], cmds, dim, dim)
    ref = (vis >> np.uint64(32)).astype(np.uint32).view(np.float32)
    blend = scene.materials["alphaMode"][scene.objects["GLTFMaterialData"]] == R.ALPHA_BLEND
    assert np.array_equal(depth, ref) or (np.abs(depth - ref) > 0).sum() < depth.size    # (identical unless something is clamped)
    assert np.array_equal(depth[(ref > 0) & (ref < 1)], ref[(ref > 0) & (ref < 1)]) and not blend.all()
    # depth bias moves every covered texel by a bounded amount, towards the bias' sign
    biased, _ = orc.raster_depth(scene, views[1:2], cmds, dim, dim, bias_const=-64.0, bias_slope=-1.5)
    covered = depth > 0
    assert (biased[covered] <= depth[covered]).all() and (biased[covered] < depth[covered]).any()
    # generic HZB cull: nothing is occluded by an empty HZB; against the view's own depth the survivors are a subset
    desc = orc.hzb_desc(dim, dim)
    campos = np.frombuffer(iv["cameraWorldPos"][0].tobytes(), dtype=np.float64)[:3]
    empty = np.zeros(desc.totalTexels, np.uint16)
    assert len(orc.hzb_culling_generic(scene, views[1:2], campos, flags, 1.5, False, desc, empty, cmds)) == len(cmds)
    words = depth.view(np.uint32).astype(np.uint64) << np.uint64(32)
    _, hmin, _, _ = orc.hzb_build(words, dim, dim)
    kept = orc.hzb_culling_generic(scene, views[1:2], campos, flags, 1.5, False, desc, hmin, cmds)
    assert 0 < len(kept) <= len(cmds) and np.isin(kept["slot"], cmds["slot"]).all()
    # ... and drawing only the survivors yields the same depth image (what was dropped was hidden)
    again, _ = orc.raster_depth(scene, views[1:2], kept, dim, dim)
    assert np.array_equal(again, depth)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,bias,debug", [(256, (0.0, 0.0), 0), (512, (-48.0, -1.25), 0), (1000, (0.0, 0.0), 0),
                                            (256, (0.0, 0.0), 65536), (512, (-48.0, -1.25), 65536)],
                         ids=["256", "512_biased", "1000_odd", "256_pixel_blocks", "512_biased_pixel_blocks"])
def test_cascade_depth_passes_match_oracle(gpu, dim, bias, debug):
    """renderShadow's loop (mesh_raster.cpp:443-531) cascade by cascade, far to near: instanceCulling for the cascade's
    orthographic view, hzbCullingGeneric against the previous cascade's HZB, clear + renderMeshDepth, buildHZB -- every
    list, depth image and HZB chain bit for bit against the oracle."""
    from chord_amd.renderer import VisibilityRenderer
    scene, cam, cfg, view, iv, views = _setup(dim=dim)
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
    r = VisibilityRenderer(0)
    r.upload_scene(scene)
    r.allocate_gbuffer(cam.width, cam.height)
    r.set_view(view, iv, flags)
    r.set_debug(debug)           # 65536: small clusters leave the setup kernel as pixel blocks (depth words, clamp and bias included)
    r.allocate_depth_views(dim, len(views))
    r.set_instance_views(views)
    desc = orc.hzb_desc(dim, dim)
    campos = np.frombuffer(iv["cameraWorldPos"][0].tobytes(), dtype=np.float64)[:3]
    prev_hzb, prev_hzb_gpu = None, None
    for k in range(len(views) - 1, -1, -1):
        vk = views[k:k + 1]
        got_list = r.instance_culling_view(k)
        want_cmds = orc.instance_culling(scene, view, vk, flags)
        assert np.array_equal(r.read_cmds(got_list), want_cmds), "cascade %d: culled list" % k
        if prev_hzb is not None:
            got_list = r.hzb_culling_generic(prev_hzb_gpu, 1.5, k + 1, False, got_list)      # the previous cascade's view (mesh_raster.cpp:484-497)
            want_cmds = orc.hzb_culling_generic(scene, views[k + 1:k + 2], campos, flags, 1.5, False, desc, prev_hzb, want_cmds)
            assert np.array_equal(H.sort_cmds(r.read_cmds(got_list)), H.sort_cmds(want_cmds)), "cascade %d: generic HZB cull" % k
        target = r.render_mesh_depth(k, got_list, True, bias[0], bias[1])
        want_depth, st = orc.raster_depth(scene, vk, want_cmds, dim, dim, True, bias[0], bias[1])
        got_depth = r.read_depth(target)
        bad = np.nonzero(got_depth.view(np.uint32) != want_depth.view(np.uint32))[0]
        assert len(bad) == 0, "cascade %d: %d texels differ, first %d: %r vs %r" % (k, len(bad), bad[0], got_depth[bad[0]], want_depth[bad[0]])
        assert r.depth_view_stats()["overflow"] == 0
        if k != 0:
            prev_hzb_gpu = r.build_hzb_from_depth(target)
            mn, _, _ = r.read_hzb(prev_hzb_gpu)
            words = want_depth.view(np.uint32).astype(np.uint64) << np.uint64(32)
            _, prev_hzb, _, _ = orc.hzb_build(words, dim, dim)
            for lv in range(desc.mipCount):
                w_, h_ = desc.mip_dims(lv); vw, vh = desc.valid_dims(lv)
                a = mn[desc.mipOffset[lv]: desc.mipOffset[lv] + w_ * h_].reshape(h_, w_)[:vh, :vw]
                b = prev_hzb[desc.mipOffset[lv]: desc.mipOffset[lv] + w_ * h_].reshape(h_, w_)[:vh, :vw]
                assert np.array_equal(a, b), "cascade %d: HZB mip %d" % (k, lv)
    # the main view is untouched by the depth passes
    r.render_frame()
    want = orc.frame(scene, view, iv, flags)
    H.assert_vis_equal(r.read_visibility(), want["vis"], cam.width, cam.height, "main view after the shadow passes")
    r.close()


def _replay_shadow(scene, view, iv, cfg, tick, hist, flags, hzb_culling=True):
    """renderShadow (mesh_raster.cpp:331-546) replayed with the oracle's pieces.  hist: {"depths", "views"} or None."""
    n, realtime, dim = int(cfg["cascadeCount"][0]), int(cfg["realtimeCascadeCount"][0]), int(cfg["cascadeDim"][0])
    cache = hist is not None
    views = L.cascade_setup(cfg, view, iv, LIGHT, tick=tick, cache_valid=cache, views=None if hist is None else hist["views"].copy())
    depths = [None] * n if hist is None else list(hist["depths"])
    desc = orc.hzb_desc(dim, dim)
    campos = np.frombuffer(iv["cameraWorldPos"][0].tobytes(), dtype=np.float64)[:3]

    def cache_valid(k):
        return cache and k >= realtime and (tick % (n - realtime)) != (k - realtime)

    def hzb_of(depth):
        return orc.hzb_build(depth.view(np.uint32).astype(np.uint64) << np.uint64(32), dim, dim)[1]

    prev, prev_k, rendered = None, None, 0
    for k in range(n - 1, -1, -1):
        if cache_valid(k):
            continue
        cmds = orc.instance_culling(scene, view, views[k:k + 1], flags)
        if prev is None:
            if cache and hzb_culling:
                cmds = orc.hzb_culling_generic(scene, hist["views"][k:k + 1], campos, flags, 1.5, False, desc, hzb_of(hist["depths"][k]), cmds)
        elif hzb_culling:
            cmds = orc.hzb_culling_generic(scene, views[prev_k:prev_k + 1], campos, flags, 1.5, False, desc, prev, cmds)
        depths[k], _ = orc.raster_depth(scene, views[k:k + 1], cmds, dim, dim, True, float(cfg["shadowBiasConst"][0]), float(cfg["shadowBiasSlope"][0]))
        rendered |= 1 << k
        if k != 0:
            prev, prev_k = hzb_of(depths[k]), k
    return {"depths": depths, "views": views}, rendered


@pytest.mark.gpu
def test_render_shadow_with_cascade_cache_matches_the_replay(gpu):
    """chordvis_render_shadow over five ticks: the first renders every cascade, the following ones the realtime cascades
    plus one far cascade per tick (isCascadeCacheValid), each first culled against the HZB of its own cached depth; the
    camera moves on the last two ticks.  Every cascade's depth image and view against the oracle replay, every tick."""
    from chord_amd.renderer import VisibilityRenderer
    scene, cam0 = scenes.masked_test_scene(320, 200)
    cfg = R.default_cascade_config(cascadeCount=5, realtimeCascadeCount=2, cascadeDim=384, cascadeEndDistance=12.0, farCascadeEndDistance=60.0,
                                   shadowBiasConst=-8.0, shadowBiasSlope=-0.5)
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
    r = VisibilityRenderer(0)
    r.upload_scene(scene)
    r.allocate_gbuffer(cam0.width, cam0.height)
    hist = None
    f = np.array(cam0.front, dtype=np.float64); f /= np.linalg.norm(f)
    masks = []
    for tick in range(5):
        cam = cam0 if tick < 3 else cam0.moved(tuple(0.4 * (tick - 2) * f))
        objs = L.fill_objects(scene, cam).copy()
        view, iv = L.make_views(cam)
        r.update_objects(objs)
        r.set_view(view, iv, flags)
        depths, views, mask = r.render_shadow(cfg, LIGHT, tick)
        hist, want_mask = _replay_shadow(scene.with_objects(objs), view, iv, cfg, tick, hist, flags)
        masks.append(mask)
        assert mask == want_mask, "tick %d: rendered cascades %s vs %s" % (tick, bin(mask), bin(want_mask))
        assert np.array_equal(views.view(np.uint8), hist["views"].view(np.uint8)), "tick %d: cascade views" % tick
        for k in range(5):
            got = r.read_depth(depths[k])
            assert np.array_equal(got.view(np.uint32), hist["depths"][k].view(np.uint32)), "tick %d cascade %d" % (tick, k)
    assert masks[0] == 0b11111 and masks[1:] == [0b00011 | (1 << (2 + t % 3)) for t in range(1, 5)]
    # a new light direction invalidates the cache: everything is rendered again
    _, _, mask = r.render_shadow(cfg, (0.1, -1.0, -0.4), 5)
    assert mask == 0b11111
    r.close()


@pytest.mark.parametrize("ranged", [False, True], ids=["full_range", "sdsm_range"])
def test_cascade_setup_agrees_with_a_float64_restatement_of_the_shader(ranged):
    """chordvis_cascade_setup (fp32, host C++) against tests/spec_np.py::cascade_views_f64, written from cascade_setup.hlsl in
    float64: matrices, inverse, the six planes and orthoDepthConvertToView of every cascade agree to fp32 accuracy (a texel
    snap that rounds the other way in fp32 may move the view by exactly one texel)."""
    import spec_np as S
    for builder in (lambda: scenes.small_test_scene(320, 200), lambda: scenes.config3_street(640, 360)):
        scene, cam = builder()
        view, iv = L.make_views(cam)
        cfg = R.default_cascade_config(cascadeCount=6, realtimeCascadeCount=2, cascadeDim=1024, cascadeEndDistance=40.0, farCascadeEndDistance=400.0)
        near = float(view["zNear"][0])
        rng = np.array([np.float32(near / 25.0), np.float32(near / 3.0)], dtype=np.float32).view(np.uint32) if ranged else None
        got = L.cascade_setup(cfg, view, iv, LIGHT, valid_range=rng)
        want = S.cascade_views_f64(cfg, view, LIGHT, valid_range=rng)
        for k, (g, w) in enumerate(zip(got, want)):
            vp, inv = _m(g, "translatedWorldToClip"), _m(g, "clipToTranslatedWorld")
            d = vp - w["vp"]
            # the snap: [0][3] / [1][3] may differ by one whole texel step; everything else to fp32 accuracy of its magnitude
            for r in (0, 1):
                steps = d[r, 3] / w["texel"]
                assert abs(steps - round(steps)) < 2e-2 and abs(round(steps)) <= 1, "cascade %d row %d: snap differs by %.3f texels" % (k, r, steps)
                d[r, 3] -= round(steps) * w["texel"]
            scale = np.abs(w["vp"]).max()
            assert np.abs(d).max() < 2e-5 * max(1.0, scale), "cascade %d: translatedWorldToClip" % k
            assert np.allclose(vp @ inv, np.eye(4), atol=5e-4)
            pl = g["frustumPlanesRS"].astype(np.float64)
            assert np.abs(pl[:, :3] - w["planes"][:, :3]).max() < 1e-4, "cascade %d: plane normals" % k
            assert np.abs(pl[:, 3] - w["planes"][:, 3]).max() < 2e-3 * max(1.0, w["radius"]), "cascade %d: plane distances" % k
            assert np.allclose(g["orthoDepthConvertToView"].astype(np.float64), w["ortho"], rtol=2e-5, atol=1e-6), "cascade %d: orthoDepthConvertToView" % k


def test_cascade_setup_matches_the_float32_fixture_bit_for_bit():
    """chordvis_cascade_setup against tests/golden/cascade_setup.json: the float32 bit patterns of every matrix, plane and
    orthoDepthConvertToView of every cascade as the float-by-float numpy restatement of cascade_setup.hlsl computes them
    (spec_np.cascade_views_f32, source order, one rounding per operation) -- no tolerance, no texel-snap allowance; three
    cases: 5 cascades without SDSM range, the reference's default 8 cascades with one, the same with the cascade cache on tick
    7 (only the scheduled far cascade is rewritten).  The restatement itself must still reproduce the fixture."""
    import json
    import os
    import spec_np
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cascade_setup.json")))
    for case in fx["cases"]:
        cfg = R.default_cascade_config(**case["config"])
        view = np.zeros(1, dtype=R.CAMERA_VIEW)
        iv = np.zeros(1, dtype=R.INSTANCE_CULLING_VIEW)
        view["zNear"], view["zFar"] = case["zNear"], case["zFar"]
        view["clipToTranslatedWorldWithZFar_NoJitter"][0] = np.asarray(case["clipToTranslatedWorldWithZFar_NoJitter_bits"], np.uint32).view(np.float32)
        vr = None if case["validDepthMinMax"] is None else np.asarray(case["validDepthMinMax"], np.uint32)
        marker = np.zeros(int(cfg["cascadeCount"][0]), dtype=R.INSTANCE_CULLING_VIEW)
        marker["renderDimension"] = -7.0
        got = L.cascade_setup(cfg, view, iv, case["lightDir"], valid_range=vr, tick=case["tick"], cache_valid=case["cacheValid"], views=marker.copy())
        again = spec_np.cascade_views_f32(cfg, view, case["lightDir"], valid_range=vr, tick=case["tick"], cache_valid=case["cacheValid"])
        for k, want in enumerate(case["cascades"]):
            if want is None:                                          # the cache keeps the view of this cascade: untouched
                assert again[k] is None and np.all(got[k]["renderDimension"] == -7.0), (case["name"], k)
                continue
            u = lambda a: np.asarray(a, np.float32).reshape(-1).view(np.uint32)
            w2c = np.asarray(got[k]["translatedWorldToClip"], np.float32).reshape(4, 4).T     # M[r][c] of the column-major record
            c2w = np.asarray(got[k]["clipToTranslatedWorld"], np.float32).reshape(4, 4).T
            for label, a, b, c in (("translatedWorldToClip", w2c, want["translatedWorldToClip_rc"], again[k]["translatedWorldToClip"]),
                                   ("clipToTranslatedWorld", c2w, want["clipToTranslatedWorld_rc"], again[k]["clipToTranslatedWorld"]),
                                   ("frustumPlanesRS", got[k]["frustumPlanesRS"], want["planes"], again[k]["planes"]),
                                   ("orthoDepthConvertToView", got[k]["orthoDepthConvertToView"], want["orthoDepthConvertToView"], again[k]["ortho"])):
                assert np.array_equal(u(a), np.asarray(b, np.uint32)), "%s cascade %d: %s differs from the fixture" % (case["name"], k, label)
                assert np.array_equal(u(c), np.asarray(b, np.uint32)), "%s cascade %d: the restatement no longer reproduces the fixture (%s)" % (case["name"], k, label)
