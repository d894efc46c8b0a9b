"""Known-answer tests that pin the CPU oracle (the reference has no golden vectors for this path:
SURVEY §4 / §8c).  Everything here is derivable from the Vulkan raster rules, IEEE-754 and the
HLSL source without running the reference."""
import ctypes as C

import numpy as np
import pytest

import helpers as HP
import orc
from chord_amd import records as R
from chord_amd import scenes

W = H = 64


def px(v):
    """pixel coordinate -> 24.8 fixed point"""
    return int(round(v * 256))


def coverage(X, Y, two_sided=True, w=W, h=H, d=(0.5, 0.5, 0.5), payload=0x101):
    vis, st = orc.raster_snapped_triangle([px(x) for x in X], [px(y) for y in Y], d, two_sided, payload, w, h)
    return (vis.reshape(h, w) != 0), st


# ------------------------------------------------------------------------------- raster rules ---

@pytest.mark.parametrize("n", [1, 2, 5, 17])
def test_right_triangle_pixel_counts(n):
    """Axis-aligned right triangle with legs n px: centres strictly inside + top-left edges.
    Vertices on integer pixel corners; the hypotenuse passes through no pixel centre only when
    counted with the fill rule: n(n-1)/2 + (diagonal centres are ON the hypotenuse)."""
    # legs along the top and left edges: top edge (y=8) and left edge (x=8) are inclusive,
    # the hypotenuse x + y = 16 + n passes exactly through the centres with (i + j) = n - 1.
    cov, _ = coverage([8, 8 + n, 8], [8, 8, 8 + n])
    ii, jj = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    cx, cy = ii + 0.5 - 8, jj + 0.5 - 8
    inside = (cx > 0) & (cy > 0) & (cx + cy < n)          # no centre lies on the legs; diagonal centres: cx+cy == n
    # hypotenuse goes from (8+n, 8) to (8, 8+n): neither top nor left => centres on it are excluded
    assert np.array_equal(cov, inside)
    assert cov.sum() == n * (n - 1) // 2


def test_top_left_rule_shared_edge_covers_once():
    """Two triangles sharing an edge cover every pixel of their union exactly once."""
    rng = np.random.default_rng(1)
    for _ in range(200):
        p = rng.uniform(4, 60, size=(4, 2))
        p = np.round(p * 256) / 256
        a, b, c, d = p
        # quad a-b-c-d split along a-c; orientation of both halves made consistent (two-sided raster)
        v1, s1 = orc.raster_snapped_triangle([px(a[0]), px(b[0]), px(c[0])], [px(a[1]), px(b[1]), px(c[1])], (0.5,) * 3, 1, 1, W, H)
        v2, s2 = orc.raster_snapped_triangle([px(a[0]), px(c[0]), px(d[0])], [px(a[1]), px(c[1]), px(d[1])], (0.5,) * 3, 1, 1, W, H)
        # only test convex, consistently wound quads (b and d on opposite sides of a-c)
        def side(p0, p1, q):
            return (p1[0] - p0[0]) * (q[1] - p0[1]) - (p1[1] - p0[1]) * (q[0] - p0[0])
        if side(a, c, b) * side(a, c, d) >= 0:
            continue
        both = (v1 != 0).astype(int) + (v2 != 0).astype(int)
        assert both.max() <= 1, "a pixel on the shared edge was covered twice"
        assert s1.fragments + s2.fragments == int((both > 0).sum())


def test_pixel_centre_on_top_and_left_edges_is_inside():
    # square [10,12]x[10,12] as two triangles: covers exactly centres 10.5, 11.5 in both axes
    c1, _ = coverage([10, 12, 12], [10, 10, 12])
    c2, _ = coverage([10, 12, 10], [10, 12, 12])
    cov = c1 | c2
    assert cov.sum() == 4 and cov[10:12, 10:12].all()
    # half-pixel shifted: edges pass through centres; top/left inclusive, bottom/right exclusive
    c1, _ = coverage([10.5, 12.5, 12.5], [10.5, 10.5, 12.5])
    c2, _ = coverage([10.5, 12.5, 10.5], [10.5, 12.5, 12.5])
    cov = c1 | c2
    assert cov.sum() == 4 and cov[10:12, 10:12].all()
    assert not (c1 & c2).any()


def test_triangle_between_pixel_centres_has_no_coverage():
    cov, st = coverage([10.55, 10.95, 10.75], [20.55, 20.6, 20.95])
    assert cov.sum() == 0 and st.fragments == 0


def test_back_face_culled_when_one_sided():
    # reference front faces (CCW in NDC, y up) are clockwise in y-down screen space: area2 < 0
    front = ([10, 10, 20], [10, 20, 10])      # (10,10)->(10,20)->(20,10): area2 = (0)(0)-(10)(10) < 0
    cov_f, _ = coverage(*front, two_sided=False)
    cov_b, _ = coverage(front[0][::-1], front[1][::-1], two_sided=False)
    assert cov_f.sum() > 0 and cov_b.sum() == 0
    cov_b2, _ = coverage(front[0][::-1], front[1][::-1], two_sided=True)
    assert np.array_equal(cov_b2, cov_f)


def test_depth_is_exact_at_vertices_and_affine():
    # right triangle, depth = plane through the vertices: d(x, y) = 0.25 + x/64 + y/128 (exact binary fractions)
    X, Y = [0, 32, 0], [0, 0, 32]
    f = lambda x, y: 0.25 + x / 64.0 + y / 128.0
    d = [f(x, y) for x, y in zip(X, Y)]
    vis, _ = orc.raster_snapped_triangle([px(x) for x in X], [px(y) for y in Y], d, 1, 7, W, H)
    vis = vis.reshape(H, W)
    depth = (vis >> np.uint64(32)).astype(np.uint32).view(np.float32)
    ys, xs = np.nonzero(vis)
    want = np.array([f(x + 0.5, y + 0.5) for x, y in zip(xs, ys)], dtype=np.float32)
    # barycentrics of power-of-two triangles are exact binary fractions -> bit-exact plane values
    assert np.array_equal(depth[ys, xs], want)
    assert ((vis[ys, xs] & np.uint64(0xFFFFFFFF)) == 7).all()


def test_larger_packed_word_wins():
    """Depth test GREATER_OR_EQUAL (helper.h:7-13) == 64-bit max; ties go to the larger payload."""
    X, Y = [px(2), px(30), px(2)], [px(2), px(2), px(30)]
    vis, _ = orc.raster_snapped_triangle(X, Y, (0.25,) * 3, 1, 5, W, H)
    vis, _ = orc.raster_snapped_triangle(X, Y, (0.5,) * 3, 1, 3, W, H, vis=vis)     # nearer (reverse-Z) wins
    vis, _ = orc.raster_snapped_triangle(X, Y, (0.125,) * 3, 1, 9, W, H, vis=vis)   # farther loses
    vis, _ = orc.raster_snapped_triangle(X, Y, (0.5,) * 3, 1, 4, W, H, vis=vis)     # tie: larger payload
    nz = vis[vis != 0]
    assert len(nz) and (nz == ((np.uint64(np.float32(0.5).view(np.uint32)) << np.uint64(32)) | np.uint64(4))).all()


def test_tile_sharding_partitions_the_pixels():
    """Screen ownership by 64 x 64 tiles: the ranks' images are disjoint, each inside its own tiles, together the full one."""
    Wb, Hb = 200, 150                                         # 4 x 3 tiles
    big = lambda v: int(round(v * 256))
    X, Y = [big(3.3), big(190.1), big(60.7)], [big(2.2), big(70.9), big(141.4)]
    full, _ = orc.raster_snapped_triangle(X, Y, (0.3, 0.6, 0.9), 1, 11, Wb, Hb)
    owners = np.array([0, 1, 2, 3, 3, 2, 1, 0, 1, 1, 0, 2], dtype=np.uint8)
    own_px = owners[(np.arange(Hb)[:, None] // 64) * 4 + np.arange(Wb)[None, :] // 64]
    acc = np.zeros_like(full)
    for r in range(4):
        part, _ = orc.raster_snapped_triangle(X, Y, (0.3, 0.6, 0.9), 1, 11, Wb, Hb, shard=(owners, 4, 4, r))
        assert not ((acc != 0) & (part != 0)).any()
        assert (own_px[part.reshape(Hb, Wb) != 0] == r).all()
        acc |= part
    assert np.array_equal(acc, full) and len(np.unique(own_px[full.reshape(Hb, Wb) != 0])) == 4


# ------------------------------------------------------------------------------- id encoding ---

def test_encode_decode_round_trip():
    from chord_amd.renderer import decode_visibility          # imports the product lib: symbols must load
    for slot in (0, 1, 255, 256, 65535, (1 << 24) - 3):
        for tri in (0, 1, 127):
            word = (((slot + 1) & 0xFFFFFF) << 8) | (tri & 0xFF)
            packed = np.array([(np.uint64(np.float32(0.75).view(np.uint32)) << np.uint64(32)) | np.uint64(word)], dtype=np.uint64)
            d, s, t = decode_visibility(packed)
            assert d[0] == np.float32(0.75) and s[0] == slot and t[0] == tri
    d, s, t = decode_visibility(np.zeros(1, dtype=np.uint64))
    assert s[0] == -1                                           # 0 == empty (sky)


# --------------------------------------------------------------------------------------- f16 ---

def test_f16_conversion_matches_ieee_rne():
    rng = np.random.default_rng(0)
    vals = np.concatenate([
        rng.uniform(0, 1, 4000).astype(np.float32),
        (rng.uniform(0, 1, 2000) * 1e-5).astype(np.float32),      # half denormals
        np.array([0.0, 1.0, 65504.0, 65520.0, 1e-8, 6.1e-5, 5.96e-8, 2.98e-8, 0.333333], dtype=np.float32),
        np.arange(0, 2048, dtype=np.float32) / 2048.0 + np.float32(2.0 ** -12),   # exact ties
    ])
    want = vals.astype(np.float16).view(np.uint16)
    got = np.array([orc.lib.orc_f32_to_f16(float(v)) for v in vals], dtype=np.uint16)
    assert np.array_equal(got, want)
    allh = np.arange(0, 0x7C00, dtype=np.uint16)
    back = np.array([orc.lib.orc_f16_to_f32(int(h)) for h in allh], dtype=np.float32)
    assert np.array_equal(back, allh.view(np.float16).astype(np.float32))


# --------------------------------------------------------------------------------------- HZB ---

def _depth_image(w, h, seed):
    rng = np.random.default_rng(seed)
    d = rng.uniform(0.0005, 0.9, size=(h, w)).astype(np.float32)
    d[rng.uniform(size=(h, w)) < 0.2] = 0.0                        # sky
    return (d.view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.uint64(0x123), d


@pytest.mark.parametrize("w,h", [(64, 64), (160, 96), (200, 120), (67, 131)])
def test_hzb_equals_brute_force_min_max(w, h):
    vis, depth = _depth_image(w, h, w * 1000 + h)
    desc, mn, mx, rng = orc.hzb_build(vis.reshape(-1), w, h, want_max=True, want_range=True)
    assert desc.width == orc.hzb_desc(w, h).width
    for l in range(desc.mipCount):
        mw, mh = desc.mip_dims(l)
        vw, vh = desc.valid_dims(l)
        size = 2 << l
        lv_mn = mn[desc.mipOffset[l]: desc.mipOffset[l] + mw * mh].reshape(mh, mw)
        lv_mx = mx[desc.mipOffset[l]: desc.mipOffset[l] + mw * mh].reshape(mh, mw)
        for y in range(vh):
            for x in range(vw):
                ys = np.minimum(np.arange(y * size, (y + 1) * size), h - 1)
                xs = np.minimum(np.arange(x * size, (x + 1) * size), w - 1)
                block = depth[np.ix_(ys, xs)]
                want_mn = np.float16(block.min()).view(np.uint16)
                want_mx = np.float16(block.max()).view(np.uint16) + (1 if l >= 5 else 0)   # hzb.hlsl:67-71
                assert lv_mn[y, x] == want_mn, (l, x, y)
                assert lv_mx[y, x] == want_mx, (l, x, y)
    valid = depth[depth > 0]
    assert rng[1] == valid.max().view(np.uint32)
    assert rng[0] == valid[valid < 1.0].min().view(np.uint32)


def test_hzb_of_constant_image_is_constant():
    w, h = 128, 96
    c = np.float32(0.3125)
    vis = np.full(w * h, (np.uint64(c.view(np.uint32)) << np.uint64(32)) | np.uint64(1), dtype=np.uint64)
    desc, mn, _, _ = orc.hzb_build(vis, w, h)
    for l in range(desc.mipCount):
        mw, mh = desc.mip_dims(l)
        vw, vh = desc.valid_dims(l)
        lv = mn[desc.mipOffset[l]: desc.mipOffset[l] + mw * mh].reshape(mh, mw)
        assert (lv[:vh, :vw] == np.float16(c).view(np.uint16)).all()


def test_hzb_desc_matches_reference_sizes():
    # hzb.cpp:49-63 worked examples
    for (w, h), (ew, eh, mips) in {(3840, 2160): (2048, 2048, 12), (1920, 1080): (1024, 1024, 11), (256, 256): (128, 128, 8)}.items():
        d = orc.hzb_desc(w, h)
        assert (d.width, d.height, d.mipCount) == (ew, eh, mips)


# ------------------------------------------------------------------------------------ culling ---

def _scene(builder=scenes.small_test_scene, **kw):
    import helpers as Hh
    return Hh.setup_scene(lambda: builder(**kw))


def test_frustum_known_boxes():
    """Hand-placed boxes against each frustum plane (camera at origin looking down -z)."""
    from chord_amd import lib as L
    pb = scenes.PrimitiveBuilder()
    pb.add_surface(scenes.plane_surface((-0.5, -0.5, 0.0), (1, 0, 0), (0, 1, 0)), 1, 1)
    sb = scenes.SceneBuilder("boxes")
    prim = sb.add_primitive(pb)
    places = {
        "centre": (0, 0, -10, True), "behind": (0, 0, 10, False), "left_out": (-30, 0, -10, False),
        "right_out": (30, 0, -10, False), "up_out": (0, 30, -10, False), "down_out": (0, -30, -10, False),
        "beyond_far": (0, 0, -20500, False), "straddle_left": (-4.3, 0, -10, True), "near": (0, 0, -0.01, True),
    }
    for x, y, z, _ in places.values():
        sb.add_object(prim, scenes.translate(x, y, z))
    scene = sb.build()
    cam = scenes.Camera((0, 0, 0), (0, 0, -1), 256, 256)
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    vis = orc.object_cull(scene, iv, R.FLAG_FRUSTUM_CULL)
    for (name, (_, _, _, want)), got in zip(places.items(), vis):
        assert bool(got) == want, name
    assert orc.object_cull(scene, iv, 0).all()            # flag off: nothing culled


def _group_flags(scene, view, groups):
    return np.array([orc.lib.orc_group_visible(view.ctypes.data, scene.objects[0:1].ctypes.data, groups[i:i + 1].ctypes.data)
                     for i in range(len(groups))], dtype=bool)


def test_lod_cut_selects_exactly_one_level_when_errors_are_monotone():
    """Along every LOD0 -> LOD1 -> LOD2 chain (own error sphere == the child's parent sphere) exactly
    one level passes isMeshletGroupVisibile when the projected errors grow up the chain."""
    from chord_amd import lib as L
    scene, _ = scenes.small_test_scene(320, 200, lods=3)
    prim = scene.primitives[scene.objects[0]["GLTFPrimitiveDetail"]]
    groups = scene.groups[prim["meshletGroupOffset"]: prim["meshletGroupOffset"] + prim["meshletGroupCount"]]
    key = {(float(g["error"]),) + tuple(float(x) for x in g["clusterPosCenter"]): i for i, g in enumerate(groups)}
    chains = []
    for i, g in enumerate(groups):
        if g["error"] >= 0:
            continue
        chain, cur = [i], g
        while cur["parentError"] < 3e38:
            j = key[(float(cur["parentError"]),) + tuple(float(x) for x in cur["parentPosCenter"])]
            chain.append(j)
            cur = groups[j]
        chains.append(chain)
    assert chains and all(len(c) == 3 for c in chains)

    checked = 0
    for dist in (6.0, 25.0, 60.0, 130.0, 300.0):
        cam = scenes.Camera((0.0, dist * 0.6, dist), (0.0, -0.55, -1.0), 320, 200)
        L.fill_objects(scene, cam)
        view, _ = L.make_views(cam)
        drawn = _group_flags(scene, view, groups)
        # float64 replay of projectSphereToScreen to find the monotone chains
        l2v = (view["translatedWorldToView"][0].reshape(4, 4).T.astype(np.float64) @
               scene.objects[0]["localToTranslatedWorld"].reshape(4, 4).T.astype(np.float64))
        k = float(view["lodScale"][0])

        def px_err(center, r):
            q = l2v[:3, :3] @ np.asarray(center, np.float64) + l2v[:3, 3]
            d2 = q @ q
            return -1.0 if d2 <= r * r else k * r / np.sqrt(d2 - r * r)
        for c in chains:
            e1 = px_err(groups[c[1]]["clusterPosCenter"], float(groups[c[1]]["error"]))
            e2 = px_err(groups[c[2]]["clusterPosCenter"], float(groups[c[2]]["error"]))
            if e1 < 0 or e2 < 0 or not (e1 * 1.001 < e2) or min(abs(e1 - 1), abs(e2 - 1)) < 1e-3:
                continue
            assert drawn[c].sum() == 1, (dist, c, e1, e2, drawn[c])
            want = 0 if e1 > 1 else (1 if e2 > 1 else 2)
            assert drawn[c][want]
            checked += 1
    assert checked > 50
    # close up only LOD0 is drawn, from far away only the root level
    cam = scenes.Camera((0.0, 1.0, 2.0), (0, -0.3, -1), 320, 200)
    L.fill_objects(scene, cam)
    assert np.array_equal(_group_flags(scene, L.make_views(cam)[0], groups), groups["error"] < 0)
    cam = scenes.Camera((0.0, 1500.0, 4000.0), (0, -0.35, -1), 320, 200)
    L.fill_objects(scene, cam)
    assert np.array_equal(_group_flags(scene, L.make_views(cam)[0], groups), groups["parentError"] > 3e38)


def test_instance_culling_invariants():
    import helpers as Hh
    scene, cam, view, iv = Hh.setup_scene(lambda: scenes.small_test_scene(200, 120))
    cmds = orc.instance_culling(scene, view, iv, Hh.ALL_FLAGS)
    assert np.array_equal(cmds["slot"], np.arange(len(cmds)))       # check(drawCmd.z == threadId), hzb_mainview_culling.hlsl:52-54
    assert (np.diff(cmds["objectId"].astype(np.int64)) >= 0).all()  # canonical (object, group, meshlet) order
    none = orc.instance_culling(scene, view, iv, 0)
    assert len(none) >= len(cmds)
    # with every test off the list is exactly the LOD cut of every object
    keys = set(zip(none["objectId"].tolist(), none["meshletId"].tolist()))
    assert all((o, m) in keys for o, m in zip(cmds["objectId"].tolist(), cmds["meshletId"].tolist()))


def test_two_pass_hzb_partitions_and_preserves_the_image():
    import helpers as Hh
    scene, cam, view, iv = Hh.setup_scene(lambda: scenes.small_test_scene(320, 200, seed=3))
    flags = Hh.ALL_FLAGS
    f0 = orc.frame(scene, view, iv, flags)
    f1 = orc.frame(scene, view, iv, flags, prev_hzb_min=f0["hzb_min"])
    c = f1["counts"]
    assert c[1] + c[2] == c[0] and c[3] <= c[2]                     # phase0 U rejected == input, phase 1 subset of rejected
    assert c[2] > 0, "test scene should have occluded clusters"
    assert np.array_equal(f1["vis"], f0["vis"])                     # static view: occlusion culling must not change the image
    vis, rej = orc.hzb_culling(scene, view, flags, 0, f0["desc"], f0["hzb_min"], f0["cmds"])
    both = np.sort(np.concatenate([vis["slot"], rej["slot"]]))
    assert np.array_equal(both, np.arange(len(f0["cmds"])))
    # HZB flag off: everything passes
    vis2, rej2 = orc.hzb_culling(scene, view, R.FLAG_FRUSTUM_CULL, 0, f0["desc"], f0["hzb_min"], f0["cmds"])
    assert len(vis2) == len(f0["cmds"]) and len(rej2) == 0


def test_consumer_contract_slot_indexes_post_cull_list():
    """lighting.hlsl:318-345 / visibility_tile.hlsl:47-54: texel -> slot -> cmd with cmd.z == slot."""
    import helpers as Hh
    from chord_amd.renderer import decode_visibility
    scene, cam, view, iv = Hh.setup_scene(lambda: scenes.small_test_scene(200, 120))
    f = orc.frame(scene, view, iv, Hh.ALL_FLAGS)
    depth, slot, tri = decode_visibility(f["vis"])
    hit = slot >= 0
    assert hit.any() and slot[hit].max() < len(f["cmds"])
    assert (f["cmds"]["slot"][slot[hit]] == slot[hit]).all()
    tcount = (scene.meshlets["vertexTriangleCount"][f["cmds"]["meshletId"][slot[hit]]] >> 8) & 0xFF
    assert (tri[hit] < tcount).all()
    assert (depth[hit] > 0).all() and (depth[~hit] == 0).all()


def test_multithreaded_raster_equals_scalar():
    import helpers as Hh
    scene, cam, view, iv = Hh.setup_scene(lambda: scenes.small_test_scene(320, 200, seed=21))
    cmds = orc.instance_culling(scene, view, iv, Hh.ALL_FLAGS)
    a, sa = orc.raster(scene, iv, cmds, 320, 200)
    b, sb = orc.raster(scene, iv, cmds, 320, 200, threads=4)
    assert np.array_equal(a, b) and sa.fragments == sb.fragments


def test_near_plane_clipping_is_watertight_and_bounded():
    """A coarse floor passing under the camera straddles w = 0 with triangles that stay on screen under the
    reference's |w| projection (mesh_raster.hlsl:159-161), so they reach the clipper: the image must be
    hole-free, covered exactly once, and nothing may be written outside the image."""
    import helpers as Hh
    for pos, front in (((0.3, 0.25, 0.2), (0.1, -0.6, -1.0)), ((0.3, 0.6, 0.2), (0.1, -1.0, -0.3))):
        scene, cam, view, iv = Hh.setup_scene(lambda: scenes.floor_under_camera(pos, front, 128, 96))
        cmds = orc.instance_culling(scene, view, iv, 0)
        vis, st = orc.raster(scene, iv, cmds, 128, 96)
        assert st.trianglesClipped > 0
        cov = vis.reshape(96, 128) != 0
        assert cov.all(), "hole in the clipped floor"
        assert st.fragments == 128 * 96                       # single layer: every pixel hit exactly once


# ---- visibility tile marker (visibility_tile.hlsl) -----------------------------------------

def _marker_case(w, h, seed):
    """A hand-made visibility buffer over a 3-object scene with shading types 1 / 37 / 100: random rectangles of
    each cluster id, single pixels on the last column / row, the rest empty."""
    scene, _ = scenes.small_test_scene(w, h)
    scene = HP.with_shading_types(scene, (1, 37, 100))
    n = 6
    cmds = np.zeros(n, dtype=R.DRAW_CMD)
    cmds["objectId"] = np.arange(n) % len(scene.objects); cmds["meshletId"] = 0; cmds["slot"] = np.arange(n)
    rng = np.random.default_rng(seed)
    vis = np.zeros((h, w), dtype=np.uint64)
    for k in range(10):
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        x1, y1 = min(w, x0 + int(rng.integers(1, 20))), min(h, y0 + int(rng.integers(1, 12)))
        slot = int(rng.integers(0, n))
        vis[y0:y1, x0:x1] = (np.uint64(0x3F000000) << np.uint64(32)) | np.uint64(((slot + 1) << 8) | int(rng.integers(0, 128)))
    vis[h - 1, 3] = np.uint64(((2 + 1) << 8) | 5)          # bottom row, top word (depth) zero on purpose: only the id counts
    vis[1, w - 1] = np.uint64(((1 + 1) << 8) | 9)          # last column
    return scene, vis.reshape(-1), cmds


@pytest.mark.parametrize("w,h", [(64, 64), (104, 72), (67, 45), (129, 33)])
def test_tile_marker_equals_the_set_of_types_per_8x8(w, h):
    """Sizes that are not multiples of 8 / 32 exercise the clamp-to-edge Gather and the dropped out-of-range stores."""
    scene, vis, cmds = _marker_case(w, h, seed=w * 1000 + h)
    got = orc.visibility_mark(scene, vis, w, h, cmds)
    want = HP.brute_force_marker(scene, vis, w, h, cmds)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert (got[:, :, 0] & 1).any() and (got[:, :, 1] & (1 << 5)).any() and (got[:, :, 3] & (1 << 4)).any()   # types 0, 37, 100 occur


def test_shading_tile_lists_and_dispatch_args():
    w, h = 104, 72
    scene, vis, cmds = _marker_case(w, h, seed=5)
    marker = orc.visibility_mark(scene, vis, w, h, cmds)
    for t in (0, 1, 37, 100, 64):
        tiles, args = orc.shading_tiles(marker, t)
        want = HP.tiles_with_type(marker, t)
        assert sorted(map(tuple, tiles.tolist())) == want and len(set(map(tuple, tiles.tolist()))) == len(tiles)
        assert args.tolist() == [(len(want) + 3) // 4, 1, 1, 1]
    assert len(orc.shading_tiles(marker, 64)[0]) == 0                      # a type nobody uses


# ---- independent restatement of the raster rule: exact rational geometry in Python -------------------------

def _ref_raster_fraction(X, Y, d, two_sided, payload, w, h):
    """Coverage from first principles with exact rationals, written independently of oracle.c: a pixel centre
    (px + 1/2, py + 1/2) is covered iff it lies strictly inside the triangle, or on an edge that is a 'top' edge
    (exactly horizontal, interior below it in y-down space) or a 'left' edge (interior to its right) -- the D3D/
    Vulkan top-left rule.  Depth: barycentric weights as exact rationals rounded ONCE to float32 is NOT the
    canonical value (the pinned formula rounds E_i and 1/2A separately), so depth is recomputed with numpy float32
    ops in the pinned order; coverage is what this reference decides on its own."""
    from fractions import Fraction as F
    P = [(F(int(X[i]), 256), F(int(Y[i]), 256)) for i in range(3)]
    area2 = (P[1][0] - P[0][0]) * (P[2][1] - P[0][1]) - (P[2][0] - P[0][0]) * (P[1][1] - P[0][1])
    out = {}
    if area2 == 0 or (area2 > 0 and not two_sided):
        return out
    # orient so that the interior is on the positive side of every directed edge
    if area2 > 0:
        P = [P[0], P[2], P[1]]
        order = [0, 2, 1]
    else:
        order = [0, 1, 2]
    # after this, walking P0->P1->P2 has negative doubled area in y-down coordinates (clockwise on screen)

    def side(a, b, q):      # > 0: q on the interior side of directed edge a->b
        return -((b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0]))

    edges = [(P[1], P[2]), (P[2], P[0]), (P[0], P[1])]      # edge i opposite vertex i
    xs = [p[0] for p in P]; ys = [p[1] for p in P]
    import math
    x0 = max(0, math.floor(min(xs)) - 1); x1 = min(w - 1, math.ceil(max(xs)) + 1)
    y0 = max(0, math.floor(min(ys)) - 1); y1 = min(h - 1, math.ceil(max(ys)) + 1)
    A = abs(int(X[1] - X[0]) * int(Y[2] - Y[0]) - int(X[2] - X[0]) * int(Y[1] - Y[0]))
    invA = np.float32(1.0) / np.float32(np.float64(A))
    dd = [np.float32(v) for v in d]
    e1, e2 = dd[1] - dd[0], dd[2] - dd[0]
    for py in range(y0, y1 + 1):
        for px in range(x0, x1 + 1):
            q = (F(2 * px + 1, 2), F(2 * py + 1, 2))
            ok = True
            for (a, b) in edges:
                sd = side(a, b, q)
                if sd > 0:
                    continue
                if sd < 0:
                    ok = False; break
                # on the edge: top edge = horizontal with the interior below (larger y); left edge = interior to the right
                ex, ey = b[0] - a[0], b[1] - a[1]
                # interior direction n = gradient of side(): d(side)/dx = ey, d(side)/dy = -ex
                nx, ny = ey, -ex
                top = (ey == 0 and ny > 0)
                left = (nx > 0)
                if not (top or left):
                    ok = False; break
            if not ok:
                continue
            # pinned depth formula, float32 ops in the pinned order, weights of ORIGINAL vertices 1 and 2
            E = [None] * 3
            for k in range(3):
                a, b = edges[k]
                E[order[k]] = side(a, b, q) * 256 * 256            # exact integer (sub-pixel^2 units)
            l1 = np.float32(np.float64(int(E[1]))) * invA
            l2 = np.float32(np.float64(int(E[2]))) * invA
            z = (dd[0] + l1 * e1) + l2 * e2
            out[(px, py)] = (int(np.float32(z).view(np.uint32)) << 32) | payload
    return out


def test_raster_rule_against_exact_rational_reference():
    """300 random triangles (slivers, sub-pixel, vertices exactly on pixel centres and on shared edges, both
    windings, off-screen parts) through orc_raster_snapped_triangle and through an independent Python restatement
    that decides coverage with exact rationals and the textbook wording of the top-left rule."""
    rng = np.random.default_rng(20260927)
    w, h = 24, 20
    covered = 0
    for it in range(300):
        mode = it % 5
        if mode == 0:      # general position
            X = rng.integers(-4 * 256, (w + 4) * 256, 3); Y = rng.integers(-4 * 256, (h + 4) * 256, 3)
        elif mode == 1:    # vertices on pixel centres / half-integers: edges through centres
            X = rng.integers(-2, w + 2, 3) * 256 + 128; Y = rng.integers(-2, h + 2, 3) * 256 + 128
        elif mode == 2:    # sub-pixel
            cx, cy = rng.integers(0, w * 256), rng.integers(0, h * 256)
            X = cx + rng.integers(-200, 200, 3); Y = cy + rng.integers(-200, 200, 3)
        elif mode == 3:    # axis-aligned edges on pixel-centre lines
            x0, y0 = rng.integers(0, w - 4) * 256 + 128, rng.integers(0, h - 4) * 256 + 128
            s = int(rng.integers(1, 5)) * 256
            X = np.array([x0, x0 + s, x0]); Y = np.array([y0, y0, y0 + s])
            if rng.integers(0, 2): X, Y = X[::-1].copy(), Y[::-1].copy()
        else:              # slivers
            x0, y0 = rng.integers(0, w * 256), rng.integers(0, h * 256)
            X = np.array([x0, x0 + rng.integers(500, 3000), x0 + rng.integers(500, 3000)])
            Y = np.array([y0, y0 + rng.integers(-40, 40), y0 + rng.integers(-40, 40)])
        X = np.asarray(X, dtype=np.int32); Y = np.asarray(Y, dtype=np.int32)
        d = rng.random(3).astype(np.float32) * np.float32(0.9) + np.float32(0.05)
        two_sided = bool(rng.integers(0, 2))
        payload = int(rng.integers(1, 1 << 31))
        vis, _ = orc.raster_snapped_triangle(X, Y, d, two_sided, payload, w, h)
        want = _ref_raster_fraction(X, Y, d, two_sided, payload, w, h)
        got = {(int(i % w), int(i // w)): int(v) for i, v in enumerate(vis) if v}
        assert got == want, "triangle %d mode %d X=%s Y=%s two_sided=%s: %d vs %d pixels, diff %s" % (
            it, mode, X.tolist(), Y.tolist(), two_sided, len(got), len(want), sorted(set(got.items()) ^ set(want.items()))[:4])
        covered += len(want)
    assert covered > 3000


def _assert_same_frame(m, f, what):
    assert np.array_equal(m["vis"], f["vis"]), what
    assert np.array_equal(m["cmds"], f["cmds"]) and m["counts"].tolist() == f["counts"].tolist(), what
    assert np.array_equal(m["hzb_min"], f["hzb_min"]) and np.array_equal(m["hzb_max"], f["hzb_max"]), what
    assert np.array_equal(m["valid_range"], f["valid_range"]), what
    for k, _ in f["stats"]._fields_:
        assert getattr(m["stats"], k) == getattr(f["stats"], k), (what, k)


@pytest.mark.parametrize("name,builder,threads", [
    ("small", lambda: scenes.small_test_scene(320, 200, seed=3), 4),
    ("small_odd", lambda: scenes.small_test_scene(333, 211, seed=8), 7),            # edge tiles, more threads than some ranges hold
    ("masked", lambda: scenes.masked_test_scene(320, 200), 3),
    ("street_360p", lambda: scenes.config3_street(640, 360), 8),
    ("floor", lambda: scenes.floor_under_camera((0.3, 0.25, 0.2), (0.1, -0.6, -1.0), 256, 192), 5),   # clipped triangles
    ("one_meshlet", scenes.config1_single_meshlet, 16),                              # more threads than objects
])
def test_all_cores_frame_replay_equals_orc_frame(name, builder, threads):
    """bench.py's all-cores CPU baseline (orc_frame_mt: culls over ranges, per-thread tile-private images merged by max, HZB
    levels over row ranges) is orc_frame: image, command list, counts per stage, HZB chains, valid range, raster statistics --
    without history, and with it (two-pass)."""
    scene, cam = builder()
    from chord_amd import lib as L
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
    f0 = orc.frame(scene, view, iv, flags)
    _assert_same_frame(orc.frame_mt(scene, view, iv, flags, None, threads), f0, name + " no history")
    cam1 = cam.moved((0.3, 0.05, -0.2))
    L.fill_objects(scene, cam1, cam)
    view1, iv1 = L.make_views(cam1, view)
    f1 = orc.frame(scene, view1, iv1, flags, prev_hzb_min=f0["hzb_min"])
    _assert_same_frame(orc.frame_mt(scene, view1, iv1, flags, f0["hzb_min"], threads), f1, name + " two-pass")
    _assert_same_frame(orc.frame_mt(scene, view1, iv1, flags, f0["hzb_min"], 1), f1, name + " two-pass, one thread")
