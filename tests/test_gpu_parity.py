"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest

import helpers as H
import orc
from chord_amd import records as R
from chord_amd import scenes

pytestmark = pytest.mark.gpu


# chordvis_set_debug switches that do not change results: small clusters as pixel blocks never / on every launch
# (by default the setup kernel decides per launch from the cluster count)
NO_BLOCKS, FORCE_BLOCKS = 32768, 65536
# the block kernel's hot-tile variant (bin slots drawn ahead), with tiles hot from 64 entries (by default: chosen from the
# previous frame's longest bin, hot from 65 536)
FORCE_HOT = 262144


def _renderer(gpu, scene, view, iv, w, h, flags, debug=0, limits=None):
    from chord_amd.renderer import VisibilityRenderer
    r = VisibilityRenderer(0)
    if limits:
        r.set_limits(**limits)
    r.upload_scene(scene)
    r.allocate_gbuffer(w, h)
    r.set_view(view, iv, flags)
    if debug:
        r.set_debug(debug)
    return r


SCENES = [
    ("small", lambda: scenes.small_test_scene(160, 96), H.ALL_FLAGS),
    ("small_hd", lambda: scenes.small_test_scene(640, 360, seed=11), H.ALL_FLAGS),
    ("small_nocone", lambda: scenes.small_test_scene(200, 120, seed=5), R.FLAG_FRUSTUM_CULL),
    ("small_nocull", lambda: scenes.small_test_scene(128, 128, seed=9, lods=2), 0),
    ("config1", scenes.config1_single_meshlet, R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL),
    # the render-size limits of renderer.h:52-53: one tile, and 4096 tiles (the tile order kernel's full table)
    ("small_64", lambda: scenes.small_test_scene(64, 64, seed=4), H.ALL_FLAGS),
    ("small_4096", lambda: scenes.small_test_scene(4096, 4096, seed=6), H.ALL_FLAGS),
    ("small_odd", lambda: scenes.small_test_scene(1237, 701, seed=8), H.ALL_FLAGS),
    # alpha-tested, blended and white-fallback materials (mesh_raster.hlsl:34-38,107-112,198-204; mesh_raster.cpp:224)
    ("masked", lambda: scenes.masked_test_scene(320, 200), H.ALL_FLAGS),
    ("masked_hd", lambda: scenes.masked_test_scene(1280, 720, lods=3, seed=9), H.ALL_FLAGS),
    # triangles of a masked floor straddling the camera plane: the clipper carries their texture coordinates
    ("masked_clipped", scenes.masked_floor_under_camera, H.ALL_FLAGS),
    ("masked_clipped_2", lambda: scenes.masked_floor_under_camera((1.1, 0.4, -0.7), (-0.7, -0.35, -0.6), 333, 211), R.FLAG_FRUSTUM_CULL),
]


@pytest.mark.parametrize("name,builder,flags", SCENES, ids=[s[0] for s in SCENES])
def test_instance_culling_matches_oracle(gpu, name, builder, flags):
    scene, cam, view, iv = H.setup_scene(builder)
    want = orc.instance_culling(scene, view, iv, flags)
    r = _renderer(gpu, scene, view, iv, cam.width, cam.height, flags)
    got = r.read_cmds(r.instance_culling())
    assert len(got) == len(want)
    # deterministic slot order: identical arrays, not just identical sets
    assert np.array_equal(got, want)
    assert np.array_equal(got["slot"], np.arange(len(got), dtype=np.uint32))   # check(drawCmd.z == threadId)
    r.close()


@pytest.mark.parametrize("name,builder,flags", SCENES, ids=[s[0] for s in SCENES])
def test_first_frame_matches_oracle(gpu, name, builder, flags):
    scene, cam, view, iv = H.setup_scene(builder)
    w, h = cam.width, cam.height
    want = orc.frame(scene, view, iv, flags)
    r = _renderer(gpu, scene, view, iv, w, h, flags)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want["vis"], w, h, name)
    st = r.stats()
    assert st["countInstanceCulled"] == want["counts"][0]
    assert st["trianglesSubmitted"] == want["stats"].trianglesSubmitted
    assert st["overflow"] == 0
    # HZB kept as history: min, max, valid range
    mn, mx, rng = r.read_hzb(r.history_hzb())
    assert np.array_equal(mn, want["hzb_min"])
    assert np.array_equal(mx, want["hzb_max"])
    assert np.array_equal(rng, want["valid_range"])
    r.close()


@pytest.mark.parametrize("name,builder,flags", SCENES[:2], ids=[s[0] for s in SCENES[:2]])
def test_two_pass_hzb_frame_matches_oracle(gpu, name, builder, flags):
    """Frame 0 builds the history HZB, frame 1 (camera moved) runs phase 0 / HZB / phase 1."""
    from chord_amd import lib as L
    scene, cam = builder()
    w, h = cam.width, cam.height
    L.fill_objects(scene, cam)
    view0, iv0 = L.make_views(cam)
    want0 = orc.frame(scene, view0, iv0, flags)
    r = _renderer(gpu, scene, view0, iv0, w, h, flags)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want0["vis"], w, h, name + " frame0")

    cam1 = cam.moved((0.35, 0.05, -0.2))
    L.fill_objects(scene, cam1, cam)        # last-frame matrices stay at frame 0 (static objects, moved camera)
    # fill_objects writes localToTranslatedWorldLastFrame from camera_last
    view1, iv1 = L.make_views(cam1, view0)
    want1 = orc.frame(scene, view1, iv1, flags, prev_hzb_min=want0["hzb_min"])
    r.update_objects(scene.objects)
    r.set_view(view1, iv1, flags)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want1["vis"], w, h, name + " frame1")
    st = r.stats()
    assert [st["countInstanceCulled"], st["countStage0Visible"], st["countStage0Rejected"], st["countStage1Visible"]] == list(want1["counts"])
    assert st["trianglesSubmitted"] == want1["stats"].trianglesSubmitted
    mn, mx, rng = r.read_hzb(r.history_hzb())
    assert np.array_equal(mn, want1["hzb_min"]) and np.array_equal(mx, want1["hzb_max"]) and np.array_equal(rng, want1["valid_range"])
    r.close()


BLOCK_SCENES = [
    ("small", lambda: scenes.small_test_scene(160, 96)),
    ("small_hd", lambda: scenes.small_test_scene(640, 360, seed=11)),
    ("masked", lambda: scenes.masked_test_scene(320, 200)),
    ("street_360p", lambda: scenes.config3_street(640, 360)),
    ("clip_view", lambda: scenes.floor_under_camera()),
]


@pytest.mark.parametrize("name,builder", BLOCK_SCENES, ids=[b[0] for b in BLOCK_SCENES])
def test_pixel_blocks_of_small_clusters_are_exact(gpu, name, builder):
    """The setup kernel's second body (clusters inside a 16x16-pixel window leave as dense blocks of packed words instead of
    triangle records) forced on for scenes that would not select it by themselves: two frames (no history / two-pass HZB)
    against the oracle, and word for word against the record path.  Clusters with masked, clipped or wide triangles, or
    that are cheaper as records, keep taking records inside this body."""
    scene, cam, view, iv = H.setup_scene(builder)
    w, h = cam.width, cam.height
    want0 = orc.frame(scene, view, iv, H.ALL_FLAGS)
    want1 = orc.frame(scene, view, iv, H.ALL_FLAGS, prev_hzb_min=want0["hzb_min"])
    rb = _renderer(gpu, scene, view, iv, w, h, H.ALL_FLAGS, FORCE_BLOCKS)
    rr = _renderer(gpu, scene, view, iv, w, h, H.ALL_FLAGS, NO_BLOCKS)
    rh = _renderer(gpu, scene, view, iv, w, h, H.ALL_FLAGS, FORCE_BLOCKS | FORCE_HOT)     # bin slots drawn ahead: holes in the bins
    for frame, want in enumerate((want0, want1)):
        rb.render_frame(); rr.render_frame(); rh.render_frame()
        got = rb.read_visibility()
        H.assert_vis_equal(got, want["vis"], w, h, "%s frame %d, pixel blocks" % (name, frame))
        assert np.array_equal(got, rr.read_visibility())
        assert np.array_equal(got, rh.read_visibility()) and rh.stats()["overflow"] == 0
        mn, mx, rng = rb.read_hzb(rb.history_hzb())
        mn2, mx2, rng2 = rr.read_hzb(rr.history_hzb())
        assert np.array_equal(mn, mn2) and np.array_equal(mx, mx2) and np.array_equal(rng, rng2)
        sb, sr = rb.stats(), rr.stats()
        assert sb["overflow"] == 0 and sr["pixelBlockBytes"] == 0
        assert sb["trianglesSubmitted"] == sr["trianglesSubmitted"] and sb["triangleRecords"] <= sr["triangleRecords"]
    rb.close(); rr.close(); rh.close()


def test_hzb_culling_lists_match_oracle(gpu):
    from chord_amd import lib as L
    scene, cam, view, iv = H.setup_scene(lambda: scenes.small_test_scene(320, 200, seed=3))
    flags = H.ALL_FLAGS
    w, h = cam.width, cam.height
    f0 = orc.frame(scene, view, iv, flags)
    r = _renderer(gpu, scene, view, iv, w, h, flags)
    r.render_frame()
    hist = r.history_hzb()
    post = r.instance_culling()
    vis, rej = r.hzb_culling(hist, True, post)
    gv, gr = r.read_cmds(vis), r.read_cmds(rej)
    wv, wr = orc.hzb_culling(scene, view, flags, 0, f0["desc"], f0["hzb_min"], f0["cmds"])
    # wave-atomic compaction: equal as sets (the reference's order is scheduling dependent too)
    assert np.array_equal(H.sort_cmds(gv), H.sort_cmds(wv))
    assert np.array_equal(H.sort_cmds(gr), H.sort_cmds(wr))
    # phase0 visible U rejected == input
    assert np.array_equal(H.sort_cmds(np.concatenate([gv, gr])), H.sort_cmds(f0["cmds"]))
    r.close()


def test_golden_config1_on_gpu(gpu):
    """The committed golden vector (tests/golden/config1.npz), through the C ABI."""
    import test_golden as TG
    g, scene, view, iv, flags = TG.load_gold()
    r = _renderer(gpu, scene, view, iv, 256, 256, flags)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), g["vis"], 256, 256, "golden config1")
    assert np.array_equal(r.read_cmds(r.last_frame_cmds()).view(np.uint8), g["cmds"])
    mn, mx, rng = r.read_hzb(r.history_hzb())
    assert np.array_equal(mn, g["hzb_min"]) and np.array_equal(mx, g["hzb_max"]) and np.array_equal(rng, g["valid_range"])
    r.close()


@pytest.mark.parametrize("pos,front", [((0.3, 0.25, 0.2), (0.1, -0.6, -1.0)), ((0.3, 0.6, 0.2), (0.1, -1.0, -0.3)),
                                       ((0.0, 0.05, 0.0), (0.7, -0.1, -0.7))])
def test_near_plane_clipper_matches_oracle(gpu, pos, front):
    """Triangles straddling w = 0 / the guard band take the clip kernel, and are big on screen."""
    scene, cam, view, iv = H.setup_scene(lambda: scenes.floor_under_camera(pos, front, 256, 192))
    want = orc.frame(scene, view, iv, 0)
    assert want["stats"].trianglesClipped > 0
    r = _renderer(gpu, scene, view, iv, 256, 192, 0)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want["vis"], 256, 192, "clipper")
    assert r.stats()["overflow"] == 0
    r.close()


def test_big_triangles_and_close_ups_match_oracle(gpu):
    """Camera a few centimetres from a wall: every triangle is hundreds of pixels (chunk kernel)."""
    def build():
        scene, _ = scenes.small_test_scene(512, 384, seed=13)
        return scene, scenes.Camera((-2.0, 0.35, 1.0), (0.9, -0.5, -0.4), 512, 384)
    scene, cam, view, iv = H.setup_scene(build)
    want = orc.frame(scene, view, iv, H.ALL_FLAGS)
    assert want["stats"].fragments > 4 * want["stats"].trianglesRastered
    r = _renderer(gpu, scene, view, iv, 512, 384, H.ALL_FLAGS)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want["vis"], 512, 384, "close-up")
    r.close()


def test_config2_atrium_1080p_matches_oracle(gpu):
    """BASELINE config 2 at full size (frustum-only, no HZB): 2048 meshlets, 262 144 triangles."""
    scene, cam, view, iv = H.setup_scene(scenes.config2_atrium)
    flags = R.FLAG_FRUSTUM_CULL
    want = orc.frame(scene, view, iv, flags)
    r = _renderer(gpu, scene, view, iv, 1920, 1080, flags)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want["vis"], 1920, 1080, "config2")
    st = r.stats()
    assert st["trianglesSubmitted"] == want["stats"].trianglesSubmitted and st["overflow"] == 0
    r.close()


def test_bins_beyond_the_fixed_capacity_use_overflow_chunks(gpu):
    """BASELINE config 4 (street x 64 instances) at 640x360: tens of thousands of triangles per 64x64 tile, so
    bins run far past their fixed 16 384 entries and continue in pool chunks.  Frame 0 (no history) and
    frame 1 (two-pass HZB; chunk-table entries of frame 0 are stale) must be exact, with no overflow."""
    from chord_amd import lib as L
    W, Hh = 640, 360
    scene, cam, view, iv = H.setup_scene(scenes.config4_street_x64, W, Hh)
    want0 = orc.frame(scene, view, iv, H.ALL_FLAGS)
    want1 = orc.frame(scene, view, iv, H.ALL_FLAGS, prev_hzb_min=want0["hzb_min"])
    # (at this size nearly every cluster is small; as pixel blocks they leave the bins short -- that variant must be exact too)
    rb = _renderer(gpu, scene, view, iv, W, Hh, H.ALL_FLAGS, FORCE_BLOCKS)
    rb.render_frame()
    H.assert_vis_equal(rb.read_visibility(), want0["vis"], W, Hh, "config4 frame 0, pixel blocks")
    assert rb.stats()["pixelBlockBytes"] > 0 and rb.stats()["overflow"] == 0
    rb.render_frame()
    H.assert_vis_equal(rb.read_visibility(), want1["vis"], W, Hh, "config4 frame 1, pixel blocks")
    rb.close()
    r = _renderer(gpu, scene, view, iv, W, Hh, H.ALL_FLAGS, NO_BLOCKS)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want0["vis"], W, Hh, "config4 frame 0")
    tiles = ((W + 63) // 64) * ((Hh + 63) // 64)
    ticks = np.zeros(tiles * 9, np.uint64); cnt = np.zeros(tiles, np.uint32)
    assert L.lib.chordvis_debug_tile_profile(r._ctx, 0, ticks.ctypes.data, cnt.ctypes.data, tiles * 9) == 0
    assert int(cnt.max()) > 16384 + 2 * 1024, "the scene no longer exercises the overflow chunks: max bin %d" % int(cnt.max())
    assert r.stats()["overflow"] == 0
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want1["vis"], W, Hh, "config4 frame 1")
    st = r.stats()
    assert st["overflow"] == 0 and st["trianglesSubmitted"] == want1["stats"].trianglesSubmitted
    r.close()


def test_config4_street_x64_4k_two_frames_match_oracle(gpu):
    """BASELINE config 4 at full size (179 M scene triangles, 260 k clusters after culling, 4K): frame 0 (no history:
    33 M triangles submitted, bins of 90 k entries) and frame 1 (two-pass HZB) bit-exact vs the oracle."""
    scene, cam, view, iv = H.setup_scene(scenes.config4_street_x64)
    W, Hh = cam.width, cam.height
    want0 = orc.frame(scene, view, iv, H.ALL_FLAGS)
    want1 = orc.frame(scene, view, iv, H.ALL_FLAGS, prev_hzb_min=want0["hzb_min"])
    r = _renderer(gpu, scene, view, iv, W, Hh, H.ALL_FLAGS)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want0["vis"], W, Hh, "config4 4K frame 0")
    st = r.stats()
    assert st["overflow"] == 0 and st["trianglesSubmitted"] == want0["stats"].trianglesSubmitted
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want1["vis"], W, Hh, "config4 4K frame 1")
    st = r.stats()
    assert st["overflow"] == 0 and st["trianglesSubmitted"] == want1["stats"].trianglesSubmitted
    mn, mx, rng = r.read_hzb(r.history_hzb())
    assert np.array_equal(mn, want1["hzb_min"]) and np.array_equal(mx, want1["hzb_max"]) and np.array_equal(rng, want1["valid_range"])
    r.close()


def test_fused_cull_of_a_scene_with_more_workgroups_than_one_poll_covers(gpu):
    """Four street instances at 1080p: 49 212 group instances = 769 workgroups of the fused short-scene cull (256 threads, 64 group
    instances each) -- a workgroup polls the look-back words of the workgroups in front of it in a loop of up to four rounds, and
    the grid is larger than the 256 CUs of the device.  Frame 0 (no history: count + scatter fused) and frame 1 (the phase-0
    occlusion cull inside the same kernel): lists, counts, image and chain equal the oracle's."""
    scene, cam, view, iv = H.setup_scene(lambda: scenes.config4_street_x64(1920, 1080, grid=2))
    W, Hh = cam.width, cam.height
    want0 = orc.frame(scene, view, iv, H.ALL_FLAGS)
    want1 = orc.frame(scene, view, iv, H.ALL_FLAGS, prev_hzb_min=want0["hzb_min"])
    r = _renderer(gpu, scene, view, iv, W, Hh, H.ALL_FLAGS)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want0["vis"], W, Hh, "street x4 frame 0")
    st = r.stats()
    assert st["overflow"] == 0 and st["trianglesSubmitted"] == want0["stats"].trianglesSubmitted
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want1["vis"], W, Hh, "street x4 frame 1")
    st = r.stats()
    assert st["overflow"] == 0 and st["trianglesSubmitted"] == want1["stats"].trianglesSubmitted
    assert [st["countInstanceCulled"], st["countStage0Visible"], st["countStage0Rejected"], st["countStage1Visible"]] == list(want1["counts"])
    mn, mx, rng = r.read_hzb(r.history_hzb())
    assert np.array_equal(mn, want1["hzb_min"]) and np.array_equal(mx, want1["hzb_max"]) and np.array_equal(rng, want1["valid_range"])
    r.close()


def test_config5_subpixel_reduced_matches_oracle(gpu):
    """BASELINE config 5 at reduced size (16 k camera-facing patches of ~8x8 px = 2.1 M triangles of ~0.5 px^2 into
    960x540): nearly every triangle survives the per-triangle culls and about half of them hit a pixel centre --
    the sub-pixel stress case; with a raised per-tile chunk limit as the full-size workload uses."""
    W, Hh = 960, 540
    scene, cam, view, iv = H.setup_scene(scenes.config5_subpixel, W, Hh, prims=16, patches_per_prim=256, instances=4)
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL
    want = orc.frame(scene, view, iv, flags)
    assert want["stats"].trianglesRastered > 0.8 * want["stats"].trianglesSubmitted
    assert 0.3 * want["stats"].trianglesRastered < want["stats"].fragments < 0.8 * want["stats"].trianglesRastered
    from chord_amd.renderer import VisibilityRenderer
    want1 = orc.frame(scene, view, iv, H.ALL_FLAGS, prev_hzb_min=want["hzb_min"])
    for mode in (NO_BLOCKS, FORCE_BLOCKS):               # one record per triangle / one pixel block per cluster and tile
        r = VisibilityRenderer(0)
        r.set_limits(max_triangle_records=8 << 20, bin_pool_chunks=16384, bin_max_chunks_per_tile=2048)
        r.upload_scene(scene)
        r.allocate_gbuffer(W, Hh)
        r.set_view(view, iv, flags)
        r.set_debug(mode)
        r.render_frame()
        H.assert_vis_equal(r.read_visibility(), want["vis"], W, Hh, "config5 reduced, mode %d" % mode)
        st = r.stats()
        assert st["overflow"] == 0 and st["trianglesSubmitted"] == want["stats"].trianglesSubmitted
        if mode == NO_BLOCKS:
            assert st["triangleRecords"] == want["stats"].trianglesRastered and st["pixelBlockBytes"] == 0
        else:
            # nearly every patch is one block (<= 2 x 2 with its tile crossings): far fewer bytes than 36 per triangle
            assert st["triangleRecords"] < 0.05 * want["stats"].trianglesRastered
            assert 0 < st["pixelBlockBytes"] < 0.5 * 36 * want["stats"].trianglesRastered
        # two-pass HZB on the same scene (occlusion between the patch layers)
        r.set_view(view, iv, H.ALL_FLAGS)
        r.render_frame()
        H.assert_vis_equal(r.read_visibility(), want1["vis"], W, Hh, "config5 reduced, HZB frame, mode %d" % mode)
        r.close()


def test_config5_hotspot_reduced_matches_oracle(gpu):
    """BASELINE config 5, variant 'hotspot' (centres Gaussian, sigma = 64 px): every cluster of the frame lands in a
    dozen 64x64 tiles -- bins far beyond their fixed part, every hot tile cut into slices that meet in the tile's slab with
    device-scope atomics.  Reduced size (2.1 M triangles into 960x540), in both forms (one record per triangle / pixel
    blocks), under the work-list limits the full-size workload documents (2048 overflow chunks per tile)."""
    W, Hh = 960, 540
    scene, cam, view, iv = H.setup_scene(scenes.config5_subpixel, W, Hh, prims=16, patches_per_prim=256, instances=4, hotspot_sigma_px=64.0)
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL
    want = orc.frame(scene, view, iv, flags)
    from chord_amd.renderer import VisibilityRenderer
    for mode in (NO_BLOCKS, FORCE_BLOCKS, FORCE_BLOCKS | FORCE_HOT):
        r = VisibilityRenderer(0)
        r.set_limits(max_triangle_records=8 << 20, bin_pool_chunks=16384, bin_max_chunks_per_tile=2048)
        r.upload_scene(scene)
        r.allocate_gbuffer(W, Hh)
        r.set_view(view, iv, flags)
        r.set_debug(mode)
        for _ in range(2):                                # (twice: the slabs of the split tiles must be all zero again)
            r.render_frame()
            H.assert_vis_equal(r.read_visibility(), want["vis"], W, Hh, "config5 hotspot reduced, mode %d" % mode)
        st = r.stats()
        assert st["overflow"] == 0 and st["trianglesSubmitted"] == want["stats"].trianglesSubmitted
        # the contention case: most tiles of the screen are empty, the hot ones hold tens of thousands of entries
        assert st["tilesTouched"][0] < 0.5 * ((W + 63) // 64) * ((Hh + 63) // 64)
        assert st["binEntries"] / max(1, st["tilesTouched"][0]) > (300 if mode & FORCE_BLOCKS else 20000)   # (a block per cluster and tile / a record per triangle)
        r.close()


def test_config5_hotspot_quarter_size_4k_matches_oracle(gpu):
    """The hotspot variant at a quarter of config 5's size and full resolution (268 M triangles, 2.1 M clusters inside a few
    dozen tiles), as the bench runs it: the dense launch takes the pixel-block kernel by itself."""
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityRenderer
    scene, cam = scenes.config5_subpixel(3840, 2160, prims=256, hotspot_sigma_px=64.0)
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL
    r = VisibilityRenderer(0)
    r.set_limits(max_triangle_records=296 << 20, bin_pool_chunks=332 << 10, bin_max_chunks_per_tile=2048)
    r.upload_scene(scene)
    r.allocate_gbuffer(cam.width, cam.height)
    r.set_view(view, iv, flags)
    r.render_frame()
    got = r.read_visibility()
    st = r.stats()
    want = orc.frame_mt(scene, view, iv, flags, None, threads=16)
    assert st["overflow"] == 0 and st["trianglesSubmitted"] == want["triangles_submitted"]
    assert st["pixelBlocks"] > 1 << 20                     # the block kernel ran
    H.assert_vis_equal(got, want["vis"], cam.width, cam.height, "config5 hotspot quarter size")
    # a second frame: the host has now seen the first frame's longest bin (> 65 536 entries) and launches the hot-tile variant of
    # the block kernel by itself -- bin slots drawn ahead, reserves filled with "no entry" words -- for the same image
    blocks0 = st["pixelBlocks"]
    r.render_frame()
    got2 = r.read_visibility()
    st2 = r.stats()
    assert st2["overflow"] == 0
    assert st2["pixelBlocks"] > blocks0, "the hot-tile variant did not run (no slots drawn ahead): %d vs %d" % (st2["pixelBlocks"], blocks0)
    assert st2["pixelBlocks"] < blocks0 + (blocks0 >> 2)   # (the unused remainders: at most 7 slots per wave and hot tile; 10 % here, 5 % at full size)
    H.assert_vis_equal(got2, want["vis"], cam.width, cam.height, "config5 hotspot quarter size, hot-tile variant")
    r.close()


def test_config5_subpixel_quarter_size_4k_matches_oracle(gpu):
    """BASELINE config 5 at a quarter of its size and full resolution: 268 M sub-pixel triangles in one pass, 120 k
    entries per 64x64 tile on average (up to 167 k: 64 equal-part slices per tile, bins deep into the pool chunks),
    raised work-list limits -- against the multi-threaded oracle replay.  (tools/verify_config5.py runs the same check
    at full size, 1.07 G triangles: also exact.)"""
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityRenderer
    scene, cam = scenes.config5_subpixel(3840, 2160, prims=256)
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL
    r = VisibilityRenderer(0)
    r.set_limits(max_triangle_records=296 << 20, bin_pool_chunks=332 << 10, bin_max_chunks_per_tile=2048)
    r.upload_scene(scene)
    r.allocate_gbuffer(cam.width, cam.height)
    r.set_view(view, iv, flags)
    r.set_debug(NO_BLOCKS)                  # one record per triangle: the long bins, slices and pool chunks are the point here
    r.render_frame()
    got = r.read_visibility()
    st = r.stats()
    want = orc.frame_mt(scene, view, iv, flags, None, threads=16)
    assert st["overflow"] == 0 and st["trianglesSubmitted"] == want["triangles_submitted"] and st["pixelBlockBytes"] == 0
    H.assert_vis_equal(got, want["vis"], cam.width, cam.height, "config5 quarter size")
    r.close()


def test_config5_subpixel_full_size_matches_oracle(gpu):
    """BASELINE config 5 at FULL size, the N > 1 bench workload: 1 073 741 824 sub-pixel triangles (8.4 M clusters) into
    3840x2160 in one pass, bit for bit against the multi-threaded oracle replay.  Run as the bench runs it: the setup
    kernel selects its pixel-block body by itself (more than one cluster per 16 pixels), so nearly every cluster leaves it
    as one block per tile instead of 128 records."""
    import os
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityRenderer
    scene, cam = scenes.config5_subpixel(3840, 2160)
    assert scene.triangle_count_lod0() == 1 << 30
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL
    r = VisibilityRenderer(0)
    r.set_limits(max_triangle_records=1152 << 20, bin_pool_chunks=1200 << 10, bin_max_chunks_per_tile=2048)
    r.upload_scene(scene)
    r.allocate_gbuffer(cam.width, cam.height)
    r.set_view(view, iv, flags)
    r.render_frame()
    got = r.read_visibility()
    st = r.stats()
    want = orc.frame_mt(scene, view, iv, flags, None, threads=min(32, os.cpu_count() or 1))
    assert st["overflow"] == 0 and st["trianglesSubmitted"] == want["triangles_submitted"] == 1 << 30
    assert st["pixelBlocks"] > 8 << 20 and st["triangleRecords"] < 1 << 20        # the block body ran
    H.assert_vis_equal(got, want["vis"], cam.width, cam.height, "config5 full size")
    r.close()


def test_config3_masked_4k_two_pass_matches_oracle(gpu):
    """bench.py --workload street_4k_masked at full size: config 3 with alpha-tested, two-sided materials on every prop and
    every other building (mesh_raster.hlsl:34-38,107-112,198-204) -- frame 0 and the two-pass frame 1 bit for bit against
    the oracle (multi-threaded replay), the alpha test really removes fragments, and the image differs from its opaque twin."""
    import os
    from chord_amd import lib as L
    scene, cam, view, iv = H.setup_scene(lambda: scenes.config3_street(masked=True))
    w, h, flags = cam.width, cam.height, H.ALL_FLAGS
    threads = min(32, os.cpu_count() or 1)
    want0 = orc.frame_mt(scene, view, iv, flags, None, threads)
    r = _renderer(gpu, scene, view, iv, w, h, flags)
    r.render_frame()
    got0 = r.read_visibility()
    H.assert_vis_equal(got0, want0["vis"], w, h, "masked config 3 frame 0")
    r.render_frame()
    want1 = orc.frame_mt(scene, view, iv, flags, want0["hzb_min"], threads)
    H.assert_vis_equal(r.read_visibility(), want1["vis"], w, h, "masked config 3 frame 1")
    st = r.stats()
    assert st["overflow"] == 0 and st["trianglesSubmitted"] == want1["triangles_submitted"] and st["countStage0Rejected"] > 0
    assert want0["stats"].fragmentsClipped > want0["stats"].fragments // 10           # the alpha test is not a formality here
    twin, tcam, tview, tiv = H.setup_scene(lambda: scenes.config3_street(masked="twin"))
    rt = _renderer(gpu, twin, tview, tiv, w, h, flags)
    rt.render_frame()
    assert rt.stats()["trianglesSubmitted"] == want0["triangles_submitted"]           # equal triangle count, by construction
    assert (rt.read_visibility() != got0).mean() > 0.01
    rt.close()
    r.close()


def test_config3_street_4k_two_pass_matches_oracle_and_properties(gpu):
    """BASELINE config 3 at full size: frame 0 (no history) and frame 1 (two-pass HZB) bit-exact vs the
    oracle; occlusion culling must not change a static image; a repeated frame is idempotent."""
    from chord_amd import lib as L
    scene, cam, view, iv = H.setup_scene(scenes.config3_street)
    w, h = cam.width, cam.height
    flags = H.ALL_FLAGS
    want0 = orc.frame(scene, view, iv, flags)
    r = _renderer(gpu, scene, view, iv, w, h, flags)
    r.render_frame()
    got0 = r.read_visibility()
    H.assert_vis_equal(got0, want0["vis"], w, h, "config3 frame0")
    r.render_frame()                                     # static camera, history from frame 0
    got1 = r.read_visibility()
    st1 = r.stats()
    assert st1["countStage0Rejected"] > 0 and st1["overflow"] == 0
    assert np.array_equal(got1, got0), "two-pass occlusion culling changed a static image"
    want1 = orc.frame(scene, view, iv, flags, prev_hzb_min=want0["hzb_min"])
    assert [st1["countInstanceCulled"], st1["countStage0Visible"], st1["countStage0Rejected"], st1["countStage1Visible"]] == list(want1["counts"])
    assert st1["trianglesSubmitted"] == want1["stats"].trianglesSubmitted
    r.render_frame()
    assert np.array_equal(r.read_visibility(), got0)     # idempotent
    # depth is in (0, 1] wherever something was drawn; the ids index the post-cull list
    from chord_amd.renderer import decode_visibility
    depth, slot, tri = decode_visibility(got0)
    hit = slot >= 0
    assert (depth[hit] > 0).all() and (depth[hit] <= 1.0).all() and slot[hit].max() < st1["countInstanceCulled"]
    r.close()


# (name, scene, ranks, tile map): "default" = the library's compact regions, "checker" = every tile border a rank border (an
# explicit map: chordvis_set_tile_owners), "rebalance" = default on frame 0, then chordvis_rebalance from that frame's loads
SHARDED = [
    ("small_3ranks", lambda: scenes.small_test_scene(320, 200, seed=17), 3, "checker"),
    # many small clusters per tile: exercises the cluster-level ownership filter of the group cull
    ("street_720p_4ranks", lambda: scenes.config3_street(1280, 720), 4, "default"),
    # config 4 (the N > 1 bench workload) at reduced size: overflow chunks and split tiles in sharded frames
    ("street_x64_360p_8ranks", lambda: scenes.config4_street_x64(640, 360), 8, "rebalance"),
    # config 5 (sub-pixel patches, the other multi-GPU workload) at reduced size
    ("subpixel_540p_8ranks", lambda: scenes.config5_subpixel(960, 540, prims=16, patches_per_prim=256, instances=4), 8, "default"),
    # full size: long bins, pool chunks and split tiles in both raster passes of a sharded frame
    ("street_x64_4k_2ranks", scenes.config4_street_x64, 2, "default"),
    ("masked_3ranks", lambda: scenes.masked_test_scene(320, 200), 3, "checker"),
    # small clusters as pixel blocks on every rank (the parts of a block in another rank's tiles are not emitted)
    ("subpixel_540p_8ranks_blocks", lambda: scenes.config5_subpixel(960, 540, prims=16, patches_per_prim=256, instances=4), 8, "checker"),
    # the hotspot variant of config 5: all the work in a few tiles around the screen centre; the map of frame 1 spreads them
    ("hotspot_540p_8ranks_blocks", lambda: scenes.config5_subpixel(960, 540, prims=16, patches_per_prim=256, instances=4, hotspot_sigma_px=64.0), 8, "rebalance"),
    ("street_x64_360p_3ranks_blocks", lambda: scenes.config4_street_x64(640, 360), 3, "default"),
    ("masked_3ranks_blocks", lambda: scenes.masked_test_scene(320, 200), 3, "default"),
]


@pytest.mark.parametrize("name,builder,ranks,tile_map", SHARDED, ids=[s[0] for s in SHARDED])
def test_sharded_frames_reassemble_to_the_single_gpu_image(gpu, name, builder, ranks, tile_map):
    """The multi-GPU path on one device: every rank's context runs its phases in turn, the three
    all-gathers (mid-frame HZB texels; end-of-frame HZB texels; visibility words) are replaced by device-to-device
    copies of the rank chunks, and each rank must end up with exactly the single-GPU visibility buffer and HZB."""
    import ctypes as C
    import torch
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityRenderer
    from chord_amd.sharding import TileLayout
    scene, cam, view, iv = H.setup_scene(builder)
    w, h, flags = cam.width, cam.height, H.ALL_FLAGS
    case_index = [c[0] for c in SHARDED].index(name)
    # the hotspot scene puts ~300 k records into its hottest tile when rendered unsharded in the record form (the single-GPU
    # reference below): beyond the default 16 Ki + 240 Ki entries per tile, so it runs under the documented raised limit
    limits = dict(bin_max_chunks_per_tile=2048) if name.startswith("hotspot") else None
    ref = _renderer(gpu, scene, view, iv, w, h, flags, limits=limits)
    lay = TileLayout(w, h, ranks)
    ctxs = []
    for rk in range(ranks):
        r = VisibilityRenderer(0)
        if limits:
            r.set_limits(**limits)
        r.upload_scene(scene)
        r.set_shard(ranks, rk)
        r.allocate_gbuffer(w, h)
        assert np.array_equal(r.tile_owners(), lay.owners)
        if tile_map == "checker":
            r.set_tile_owners([(t % lay.tiles_x + t // lay.tiles_x) % ranks for t in range(lay.tiles)])
        r.set_view(view, iv, flags)
        if name.endswith("_blocks"):
            # (the hotspot case: odd ranks also draw bin slots ahead on hot tiles, the sharded form of the hot-tile variant)
            r.set_debug(FORCE_BLOCKS | (FORCE_HOT if name.startswith("hotspot") and rk % 2 else 0))
        ctxs.append(r)
    hip = L._preload_hip_runtime()
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def gather(ptrs, chunk_bytes):
        for r in ctxs:
            r.sync()
        for dst in range(ranks):
            for src in range(ranks):
                if src != dst:
                    assert hip.hipMemcpy(ptrs[dst] + src * chunk_bytes, ptrs[src] + src * chunk_bytes, chunk_bytes, 3) == 0
        # a device-to-device hipMemcpy is ordered on the null stream only; the contexts run on non-blocking streams
        assert hip.hipDeviceSynchronize() == 0

    # the small cases are also held against the ORACLE directly (not only against the single-GPU HIP frame: a defect common
    # to both HIP paths would pass the comparison between them)
    vs_oracle = name in ("small_3ranks", "masked_3ranks", "subpixel_540p_8ranks_blocks", "hotspot_540p_8ranks_blocks", "street_x64_360p_8ranks")
    prev_hzb = None
    for frame in range(3 if tile_map == "rebalance" else 2):   # frame 0: no history; frame 1: two-pass HZB (frame 2: again, under the re-balanced map's successor)
        ref.render_frame()
        want = ref.read_visibility()
        wmn, wmx, wrng = ref.read_hzb(ref.history_hzb())
        if vs_oracle:
            import orc
            o = orc.frame(scene, view, iv, flags, prev_hzb_min=prev_hzb)
            prev_hzb = o["hzb_min"]
            H.assert_vis_equal(want, o["vis"], w, h, "frame %d single-GPU vs oracle" % frame)
            want = o["vis"]                              # the ranks below are compared with the oracle's image
        # the sharded group cull (every frame but frame 0 of every other case, which starts at phase a: the replicated cull): each
        # rank tests its share of the group instances, the rank-mask words are all-gathered, phase a goes on from them
        if frame > 0 or case_index % 2 == 0:
            for r in ctxs:
                r.frame_phase_cull()
            cx = [r.cull_exchange() for r in ctxs]
            assert all(c[0] and c[1] == cx[0][1] for c in cx)
            gather([c[0] for c in cx], cx[0][1])
        for r in ctxs:
            r.frame_phase_a()
        ex = [r.hzb_exchange() for r in ctxs]
        gather([e[0] for e in ex], ex[0][2] * 2)
        for r in ctxs:
            r.frame_phase_b()
        fin = [r.hzb_final_exchange() for r in ctxs]
        gather([f[0] for f in fin], fin[0][1])
        gather([r.visibility_ptr() for r in ctxs], ctxs[0].visibility_chunk_words() * 8)
        for r in ctxs:
            r.frame_phase_c()
        for rk, r in enumerate(ctxs):
            H.assert_vis_equal(r.read_visibility(), want, w, h, "frame %d rank %d" % (frame, rk))
            mn, mx, rng = r.read_hzb(r.history_hzb())
            assert np.array_equal(mn, wmn) and np.array_equal(mx, wmx) and np.array_equal(rng, wrng)
        H.assert_rank_counts([r.stats() for r in ctxs], ref.stats())
        # the full post-cull list (what the visibility ids index): a rank of a sharded frame writes only its own share during the
        # frame and makes the whole list when a consumer asks -- identical to the single-GPU list, slots included; the tile marker
        # of a rank (which reads it) equals the single-GPU marker
        if frame == 1:
            # (the ORACLE's command array, slots included -- in every case; the single-GPU list must be the same array)
            import orc
            want_cmds = orc.instance_culling(scene, view, iv, flags)
            assert np.array_equal(ref.read_cmds(ref.last_frame_cmds()), want_cmds)
            for r in (ctxs[0], ctxs[-1]):
                assert np.array_equal(r.read_cmds(r.last_frame_cmds()), want_cmds)
            assert np.array_equal(ctxs[1].read_tile_marker(ctxs[1].visibility_mark()), ref.read_tile_marker(ref.visibility_mark()))
        # every rank holds every tile's load after the end-of-frame exchange; tiles nobody drew into report 0
        loads = [r.read_tile_loads() for r in ctxs]
        assert all(np.array_equal(loads[0], x) for x in loads[1:]) and loads[0].sum() > 0
        if tile_map == "rebalance" and frame < 2:
            before = [r.rebalance() for r in ctxs]
            maps = [r.tile_owners() for r in ctxs]
            assert all(np.array_equal(maps[0], m) for m in maps[1:]) and len(set(before)) == 1
            per = np.bincount(maps[0], weights=loads[0].astype(np.float64), minlength=ranks)
            # the new map under the loads it was made from: no rank far above the mean unless a single tile is
            assert per.max() <= max(1.3 * per.mean(), 1.02 * loads[0].max()), (per.tolist(), before)
            assert np.bincount(maps[0], minlength=ranks).max() == ctxs[0].visibility_chunk_words() // 4096
    for r in ctxs + [ref]:
        r.close()


def test_tile_marker_and_shading_tile_lists_match_oracle(gpu):
    """visibilityMark + prepareShadingTileParam on a rendered frame whose objects carry five different
    shading types; render size not a multiple of 8 or 32.  Marker bit-exact; tile lists equal as sets (the
    reference's list order is scheduling-dependent), counts and dispatch arguments exact."""
    W, Hh = 203, 117
    scene, cam = scenes.small_test_scene(W, Hh, seed=21)
    scene = H.with_shading_types(scene)
    from chord_amd import lib as L
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    want = orc.frame(scene, view, iv, H.ALL_FLAGS)
    r = _renderer(gpu, scene, view, iv, W, Hh, H.ALL_FLAGS)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want["vis"], W, Hh, "marker frame")
    marker = r.visibility_mark()
    got = r.read_tile_marker(marker)
    ref = orc.visibility_mark(scene, want["vis"], W, Hh, want["cmds"])
    assert got.shape == ref.shape and np.array_equal(got, ref)
    assert np.array_equal(ref, H.brute_force_marker(scene, want["vis"], W, Hh, want["cmds"]))
    types_seen = [t for t in (0, 1, 37, 100, 64, 127) if H.tiles_with_type(ref, t)]
    assert len(types_seen) >= 4, types_seen
    for t in (0, 1, 37, 100, 64, 127, 5):
        tiles, args = r.read_shading_tiles(r.prepare_shading_tile_param(t, marker))
        ref_tiles, ref_args = orc.shading_tiles(ref, t)
        assert sorted(map(tuple, tiles.tolist())) == sorted(map(tuple, ref_tiles.tolist()))
        assert args.tolist() == ref_args.tolist()
    r.close()


def test_tile_marker_config3_4k(gpu):
    """The marker of the full-size config 3 frame (4K) against the oracle's."""
    scene, cam = scenes.config3_street()
    scene = H.with_shading_types(scene, (1, 2, 3, 66, 99))
    from chord_amd import lib as L
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    want = orc.frame(scene, view, iv, H.ALL_FLAGS)
    r = _renderer(gpu, scene, view, iv, cam.width, cam.height, H.ALL_FLAGS)
    r.render_frame()
    marker = r.visibility_mark()
    got = r.read_tile_marker(marker)
    ref = orc.visibility_mark(scene, want["vis"], cam.width, cam.height, want["cmds"])
    assert np.array_equal(got, ref)
    tiles, args = r.read_shading_tiles(r.prepare_shading_tile_param(66, marker))
    assert sorted(map(tuple, tiles.tolist())) == H.tiles_with_type(ref, 66) and args[0] == (len(tiles) + 3) // 4
    r.close()


@pytest.mark.parametrize("name,builder", [("config3", scenes.config3_street), ("config4", scenes.config4_street_x64)])
def test_moving_camera_sequence_matches_oracle(gpu, name, builder):
    """What bench.py renders: frames alternating between two cameras 0.5 m apart, every frame culling against the HZB of
    the previous (different) view through the objects' last-frame transforms.  Four frames B, A, B, A at full size,
    each against the oracle fed with the oracle's own previous HZB."""
    from chord_amd import lib as L
    scene, cam_a = builder()
    f = np.array(cam_a.front, dtype=np.float64); f /= np.linalg.norm(f)
    cam_b = cam_a.moved(tuple(0.5 * f))
    va0, _ = L.make_views(cam_a); vb0, _ = L.make_views(cam_b)
    views = {"a": L.make_views(cam_a, vb0), "b": L.make_views(cam_b, va0)}
    objs = {"a": L.fill_objects(scene, cam_a, cam_b).copy(), "b": L.fill_objects(scene, cam_b, cam_a).copy()}

    def scene_with(o):
        return scene.with_objects(o)
    from chord_amd.renderer import VisibilityRenderer
    r = VisibilityRenderer(0)
    r.upload_scene(scene)
    r.allocate_gbuffer(cam_a.width, cam_a.height)
    prev = None
    for i, k in enumerate("baba"):
        view, iv = views[k]
        want = orc.frame(scene_with(objs[k]), view, iv, H.ALL_FLAGS, prev_hzb_min=prev)
        r.update_objects(objs[k])
        r.set_view(view, iv, H.ALL_FLAGS)
        r.render_frame()
        H.assert_vis_equal(r.read_visibility(), want["vis"], cam_a.width, cam_a.height, "%s frame %d (view %s)" % (name, i, k))
        st = r.stats()
        assert st["overflow"] == 0 and st["trianglesSubmitted"] == want["stats"].trianglesSubmitted
        assert [st["countInstanceCulled"], st["countStage0Visible"], st["countStage0Rejected"], st["countStage1Visible"]] == want["counts"].tolist() \
            or prev is None
        prev = want["hzb_min"]
    r.close()


def test_kept_tile_schedule_renders_the_same_frames(gpu):
    """chordvis_set_tile_schedule_keep: the first raster pass of a frame takes its work items, their order and the cut of long bins
    from the schedule of an earlier frame for up to `frames` frames (7 here; the library's default is 1).  Ten frames of config 3 at 1080p (bins of several thousand
    entries: cut tiles) along a camera path with a cut to the opposite direction in the middle, rendered by a context with the default
    and by one that makes a fresh schedule every frame: both equal the oracle's frames (fed with its own previous HZB), images and
    counts, frame by frame -- and the context with the default launches one kernel less in the frames between two schedules."""
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityRenderer
    scene, cam0 = scenes.config3_street(1920, 1080)
    f = np.array(cam0.front, dtype=np.float64); f /= np.linalg.norm(f)
    cams = [cam0.moved(tuple(0.5 * i * f)) for i in range(5)]
    # (the cut: from the far end of the street, looking back along it -- other tiles are the heavy ones, other bins the long ones)
    back = scenes.Camera(tuple(np.array(cam0.position) + 90.0 * f * np.array([1.0, 0.0, 1.0])), (-cam0.front[0], cam0.front[1], -cam0.front[2]), cam0.width, cam0.height)
    cams += [back.moved(tuple(-0.5 * i * f)) for i in range(5)]
    ctx = []
    for keep in (7, 0):                                           # (7: a long-kept schedule, whatever the library's default is)
        r = VisibilityRenderer(0)
        r.upload_scene(scene)
        r.allocate_gbuffer(cam0.width, cam0.height)
        if keep is not None:
            r.set_tile_schedule_keep(keep)
        ctx.append(r)
    assert ctx[0].tile_schedule_keep() == 7 and ctx[1].tile_schedule_keep() == 0
    probe = VisibilityRenderer(0); assert probe.tile_schedule_keep() == 1; probe.close()      # the library's default: the frame a schedule is made in, and the next
    prev = None
    launches = [[], []]
    for i, cam in enumerate(cams):
        last = cams[i - 1] if i else cam
        view0, _ = L.make_views(last)
        view, iv = L.make_views(cam, view0)
        objs = L.fill_objects(scene, cam, last).copy()
        want = orc.frame(scene.with_objects(objs), view, iv, H.ALL_FLAGS, prev_hzb_min=prev)
        for k, r in enumerate(ctx):
            r.update_objects(objs)
            r.set_view(view, iv, H.ALL_FLAGS)
            r.render_frame()
            st = r.stats()
            launches[k].append(st["kernelLaunches"])
            H.assert_vis_equal(r.read_visibility(), want["vis"], cam.width, cam.height, "frame %d, schedule kept for %d frames" % (i, r.tile_schedule_keep()))
            assert st["overflow"] == 0 and st["trianglesSubmitted"] == want["stats"].trianglesSubmitted
            if prev is not None:
                assert [st["countInstanceCulled"], st["countStage0Visible"], st["countStage0Rejected"], st["countStage1Visible"]] == want["counts"].tolist()
        prev = want["hzb_min"]
    # frame 0 has no history (one raster pass, both contexts alike: neither has a schedule yet); from frame 1 on the first context launches
    # no schedule kernel for the first pass -- every frame's tile kernel makes the next frame's schedule --
    assert launches[0][0] == launches[1][0], launches
    # (one launch less: the first pass's schedule; two where the second pass is a heavy one -- after the cut -- and keeps its schedule too)
    assert all(b - 2 <= a <= b - 1 for a, b in zip(launches[0][1:], launches[1][1:])), launches
    assert launches[0][1] == launches[1][1] - 1 and launches[0][2] == launches[1][2] - 1, launches
    for r in ctx:
        r.close()


def test_kept_schedule_of_a_heavy_second_pass_follows_a_camera_cut(gpu):
    """A frame's HEAVY second pass keeps its tile schedule too (launch_raster: orderAll): the schedule lists every tile, touched or not,
    so a later frame that touches other tiles finds them.  Seven frames of config 3 at 1080p in which every object 'was' 500 m further
    down the view direction in the frame before (phase 0 rejects what the history covers, the second pass draws the scene: ~3 800
    clusters, heavy), three views along the street, then a cut to the far end looking back: a context that keeps schedules for
    7 frames and one that makes every schedule afresh both equal the oracle frame by frame, and from frame 2 on the first one runs
    TWO launches fewer per frame (no schedule kernel in either pass)."""
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityRenderer
    scene, cam0 = scenes.config3_street(1920, 1080)
    f = np.array(cam0.front, dtype=np.float64); f /= np.linalg.norm(f)
    cams = [cam0.moved(tuple(0.5 * i * f)) for i in range(3)]
    back = scenes.Camera(tuple(np.array(cam0.position) + 90.0 * f * np.array([1.0, 0.0, 1.0])), (-cam0.front[0], cam0.front[1], -cam0.front[2]), cam0.width, cam0.height)
    cams += [back.moved(tuple(-0.5 * i * f)) for i in range(4)]
    ctx = []
    for keep in (7, 0):                                           # (7: a long-kept schedule, whatever the library's default is)
        r = VisibilityRenderer(0)
        r.upload_scene(scene)
        r.allocate_gbuffer(cam0.width, cam0.height)
        if keep is not None:
            r.set_tile_schedule_keep(keep)
        ctx.append(r)
    prev = None
    launches = [[], []]
    for i, cam in enumerate(cams):
        lastCam = cams[i - 1] if i else cam
        fc = np.array(cam.front, dtype=np.float64); fc /= np.linalg.norm(fc)
        last = scene.local_to_world.copy()
        last[:, 12:15] += 500.0 * fc                              # glm column-major: the translation column
        view0, _ = L.make_views(lastCam)
        view, iv = L.make_views(cam, view0)
        objs = L.fill_objects(scene, cam, lastCam, last).copy()
        want = orc.frame(scene.with_objects(objs), view, iv, H.ALL_FLAGS, prev_hzb_min=prev)
        if prev is not None and i != 3:
            assert want["counts"][3] > 1024                       # the second pass is a heavy one (TILE_DIRECT_MAX_CLUSTERS)
        if i == 3:
            assert want["counts"][3] == 0                         # (the cut: the history hides nothing of the new view -- an EMPTY second pass under the kept schedule)
        for k, r in enumerate(ctx):
            r.update_objects(objs)
            r.set_view(view, iv, H.ALL_FLAGS)
            r.render_frame()
            st = r.stats()
            launches[k].append(st["kernelLaunches"])
            H.assert_vis_equal(r.read_visibility(), want["vis"], cam.width, cam.height, "frame %d, schedules kept for %d frames" % (i, r.tile_schedule_keep()))
            assert st["overflow"] == 0 and st["trianglesSubmitted"] == want["stats"].trianglesSubmitted
            if prev is not None:
                assert [st["countInstanceCulled"], st["countStage0Visible"], st["countStage0Rejected"], st["countStage1Visible"]] == want["counts"].tolist()
            mn, mx, rng = r.read_hzb(r.history_hzb())
            assert np.array_equal(mn, want["hzb_min"]) and np.array_equal(mx, want["hzb_max"]) and np.array_equal(rng, want["valid_range"])
        prev = want["hzb_min"]
    # frame 0: one raster pass, both alike; frame 1: the second pass's first schedule is made in both; from frame 2 on neither pass of
    # the context with the default launches a schedule kernel (the other context's second pass runs direct in the frame after the empty
    # one -- its schedule kernel reported a light pass --, so the difference there is one launch, not two)
    assert launches[0][0] == launches[1][0] and launches[0][1] == launches[1][1] - 1, launches
    assert launches[0][2] == launches[1][2] - 2 and all(a <= b - 1 for a, b in zip(launches[0][2:], launches[1][2:])), launches
    # (a second pass that ran direct -- the frame after the empty one -- made no schedule: the next heavy one launches the schedule kernel once)
    assert max(launches[0][2:]) - min(launches[0][2:]) <= 1, launches
    for r in ctx:
        r.close()


def test_kept_tile_schedule_survives_a_cut_into_a_hotspot(gpu):
    """The work items of a kept schedule were cut for an EARLIER frame's bins.  A camera that turns from an empty view into config 5's
    hotspot (reduced, record form: the hottest tile's bin holds some 300 k entries) meets a schedule in which every tile is one whole
    work item: that item then spans far more overflow chunks than the 64 names the tile kernel holds in LDS at once (16 384 + 64 x
    1 024 entries), and the kernel must slide its chunk window along the bin -- round 5's kernel read everything past the window as
    'no entry' and dropped those triangles without a word.  Frames: away (a schedule of empty tiles is made), the
    hotspot twice under that kept schedule, away, the hotspot again; against the oracle, beside a context that makes a fresh schedule
    every frame -- which must launch exactly one kernel more in the kept frames."""
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityRenderer
    W, Hh = 960, 540
    scene, hot = scenes.config5_subpixel(W, Hh, prims=16, patches_per_prim=256, instances=4, hotspot_sigma_px=64.0)
    away = scenes.Camera((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), W, Hh)
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL
    order = [away, hot, hot, away, hot]
    ctx = []
    for keep in (7, 0):                                           # (7: a long-kept schedule, whatever the library's default is)
        r = VisibilityRenderer(0)
        r.set_limits(max_triangle_records=8 << 20, bin_pool_chunks=16384, bin_max_chunks_per_tile=2048)
        r.upload_scene(scene)
        r.allocate_gbuffer(W, Hh)
        if keep is not None:
            r.set_tile_schedule_keep(keep)
        ctx.append(r)
    tiles = ((W + 63) // 64) * ((Hh + 63) // 64)
    launches = [[], []]
    want_hot = None
    for i, cam in enumerate(order):
        last = order[i - 1] if i else cam
        view0, _ = L.make_views(last)
        view, iv = L.make_views(cam, view0)
        objs = L.fill_objects(scene, cam, last).copy()
        if cam is hot and want_hot is None:
            want_hot = orc.frame(scene.with_objects(objs), view, iv, flags)
        for k, r in enumerate(ctx):
            r.update_objects(objs)
            r.set_view(view, iv, flags)
            r.render_frame()
            got = r.read_visibility()
            st = r.stats()
            launches[k].append(st["kernelLaunches"])
            assert st["overflow"] == 0
            if cam is away:
                assert st["trianglesSubmitted"] == 0 and not got.any(), "frame %d: the empty view drew something" % i
                continue
            assert st["trianglesSubmitted"] == want_hot["stats"].trianglesSubmitted
            cnt = np.zeros(tiles, np.uint32); ticks = np.zeros(tiles * 9, np.uint64)
            assert L.lib.chordvis_debug_tile_profile(r._ctx, 0, ticks.ctypes.data, cnt.ctypes.data, tiles * 9) == 0
            assert int(cnt.max()) > 16384 + 64 * 1024 + 4096, "the scene no longer runs a bin past the chunk window: max bin %d" % int(cnt.max())
            H.assert_vis_equal(got, want_hot["vis"], W, Hh, "frame %d (hotspot), schedule kept for %d frames" % (i, r.tile_schedule_keep()))
    # frames 1..4 of the first context run under the schedule the frame before them made -- frames 1 and 4 under one made for an empty
    # view --: one launch less than the control
    assert launches[0][0] == launches[1][0], launches
    assert all(a == b - 1 for a, b in zip(launches[0][1:], launches[1][1:])), launches
    for r in ctx:
        r.close()


@pytest.mark.parametrize("w,h", [(400, 240), (1237, 701)])
def test_frames_with_nothing_in_view_match_oracle(gpu, w, h):
    """The empty input of this path: a camera that turns its back on the whole scene.  Frame 1 sees the scene, frame 2 nothing
    (every object fails the frustum test: empty command list, no raster work, an all-zero image and the HZB of one), frame 3 the
    scene again against that empty history (nothing may be occluded by it), frame 4 as usual -- each against the oracle fed with
    the oracle's own previous HZB, counts included."""
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityRenderer
    scene, cam_a = scenes.small_test_scene(w, h, seed=13)
    away = scenes.Camera((-30.0, 1.6, 30.0), (-0.8, 0.0, 0.6), w, h)            # (outside the scene, looking away from all of it)
    order = [cam_a, away, cam_a, cam_a]
    r = VisibilityRenderer(0)
    r.upload_scene(scene)
    r.allocate_gbuffer(w, h)
    prev, prev_view = None, None
    for i, cam in enumerate(order):
        last = order[i - 1] if i else cam
        view0, _ = L.make_views(last)
        view, iv = L.make_views(cam, view0)
        objs = L.fill_objects(scene, cam, last).copy()
        want = orc.frame(scene.with_objects(objs), view, iv, H.ALL_FLAGS, prev_hzb_min=prev)
        r.update_objects(objs)
        r.set_view(view, iv, H.ALL_FLAGS)
        r.render_frame()
        got = r.read_visibility()
        H.assert_vis_equal(got, want["vis"], w, h, "frame %d" % i)
        st = r.stats()
        assert st["overflow"] == 0 and st["trianglesSubmitted"] == want["stats"].trianglesSubmitted
        if prev is not None:
            assert [st["countInstanceCulled"], st["countStage0Visible"], st["countStage0Rejected"], st["countStage1Visible"]] == want["counts"].tolist()
        if cam is away:
            assert st["trianglesSubmitted"] == 0 and st["countInstanceCulled"] == 0 and not got.any()
        else:
            assert st["trianglesSubmitted"] > 0 and got.any()
        prev = want["hzb_min"]
    r.close()


@pytest.mark.parametrize("name,builder", [("small", lambda: scenes.small_test_scene(400, 240, seed=31)),
                                          ("config3_1080p", lambda: scenes.config3_street(1920, 1080)),
                                          ("config4_4k", scenes.config4_street_x64)])
def test_individual_passes_compose_to_the_frame(gpu, name, builder):
    """the frame recorded pass by pass through the stand-alone entry points (clear, instance
    culling, stage 0, buildHZB, stage 1, buildHZB -- renderer.cpp:315-345), with the caller carrying the history HZB,
    gives the oracle's frames.  This is the path without the fused clear / fused HZB of chordvis_render_frame (a real
    clear, the three-kernel HZB build, later raster passes merging with atomics); config 4 runs it with long bins."""
    scene, cam, view, iv = H.setup_scene(builder)
    W, Hh = cam.width, cam.height
    want0 = orc.frame(scene, view, iv, H.ALL_FLAGS)
    want1 = orc.frame(scene, view, iv, H.ALL_FLAGS, prev_hzb_min=want0["hzb_min"])
    r = _renderer(gpu, scene, view, iv, W, Hh, H.ALL_FLAGS)
    hist = None
    for frame, want in enumerate((want0, want1)):
        r.clear_gbuffer()                                                   # addClearGbufferPass       renderer.cpp:315
        post = r.instance_culling()                                         # instanceCulling           :321
        stage1, rejected = r.visibility_stage0(hist, post)                  # :326
        assert stage1 == (hist is not None)
        if stage1:
            tmp = r.build_hzb(True, False, False, slot=0)                   # buildHZB(min)             :334
            r.visibility_stage1(tmp, rejected)                              # :337
        hist = r.build_hzb(True, True, True, slot=1 + (frame & 1))          # buildHZB(min, max, range) :343
        H.assert_vis_equal(r.read_visibility(), want["vis"], W, Hh, "%s pass-by-pass frame %d" % (name, frame))
        mn, mx, rng = r.read_hzb(hist)
        d = want["desc"]
        for l in range(d.mipCount):                                         # only the sampled extent of a mip is defined
            vw, vh = d.valid_dims(l)
            mw, _ = d.mip_dims(l)
            o = d.mipOffset[l]
            for arr, ref in ((mn, want["hzb_min"]), (mx, want["hzb_max"])):
                a = arr[o:o + mw * max(1, d.height >> l)].reshape(-1, mw)[:vh, :vw]
                b = ref[o:o + mw * max(1, d.height >> l)].reshape(-1, mw)[:vh, :vw]
                assert np.array_equal(a, b), (name, frame, l)
        assert np.array_equal(rng, want["valid_range"])
        got_cmds = r.read_cmds(post)
        assert np.array_equal(H.sort_cmds(got_cmds), H.sort_cmds(want["cmds"]))
    assert r.stats()["overflow"] == 0
    r.close()


def test_context_reuse_across_sizes_and_scenes(gpu):
    """One context, re-targeted: a new render size (all per-size buffers are re-made, the HZB history is dropped), then a
    new scene -- every frame must equal the oracle's first frame / two-pass frame for that state."""
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityRenderer
    r = VisibilityRenderer(0)
    for builder, w, h in ((lambda: scenes.small_test_scene(640, 360, seed=41), 640, 360),
                          (lambda: scenes.small_test_scene(320, 200, seed=41), 320, 200),
                          (lambda: scenes.config2_atrium(800, 448), 800, 448),
                          (lambda: scenes.small_test_scene(1000, 600, seed=43), 1000, 600)):
        scene, cam, view, iv = H.setup_scene(builder)
        r.upload_scene(scene)
        r.allocate_gbuffer(w, h)
        r.set_view(view, iv, H.ALL_FLAGS)
        r.reset_history()
        want0 = orc.frame(scene, view, iv, H.ALL_FLAGS)
        want1 = orc.frame(scene, view, iv, H.ALL_FLAGS, prev_hzb_min=want0["hzb_min"])
        r.render_frame()
        H.assert_vis_equal(r.read_visibility(), want0["vis"], w, h, "%s %dx%d frame 0" % (scene.name, w, h))
        r.render_frame()
        H.assert_vis_equal(r.read_visibility(), want1["vis"], w, h, "%s %dx%d frame 1" % (scene.name, w, h))
        assert r.stats()["overflow"] == 0
    r.close()


@pytest.mark.parametrize("pos,front", [((-62.0, 0.25, 3.0), (1.0, -0.02, -0.04)), ((-30.0, 0.6, 0.5), (0.3, -0.9, 0.2)),
                                       ((-55.0, 1.2, 7.5), (0.9, 0.1, -0.4))])
def test_config3_4k_ground_level_views_match_oracle(gpu, pos, front):
    """Config 3 at 4K from cameras a hand above the ground / against a wall: thousands of triangles cross the near and
    guard planes (homogeneous clipper at scale), many span dozens of tiles (large list, 48-byte records, int64 /
    fp64 edge functions).  Two frames each (no history, then two-pass HZB)."""
    def build():
        scene, cam = scenes.config3_street()
        return scene, scenes.Camera(pos, front, cam.width, cam.height)
    scene, cam, view, iv = H.setup_scene(build)
    W, Hh = cam.width, cam.height
    want0 = orc.frame(scene, view, iv, H.ALL_FLAGS)
    want1 = orc.frame(scene, view, iv, H.ALL_FLAGS, prev_hzb_min=want0["hzb_min"])
    r = _renderer(gpu, scene, view, iv, W, Hh, H.ALL_FLAGS)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want0["vis"], W, Hh, "ground view frame 0")
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want1["vis"], W, Hh, "ground view frame 1")
    st = r.stats()
    assert st["overflow"] == 0 and st["triangleRecords"] > st["triangleRecordsCompact"]      # wide records exist
    r.close()


SECOND_PASS = [
    ("small", lambda: scenes.small_test_scene(160, 96)),
    ("small_hd", lambda: scenes.small_test_scene(640, 360, seed=11)),
    ("small_odd", lambda: scenes.small_test_scene(1237, 701, seed=8)),                # edge tiles narrower / shorter than 64 pixels
    ("street_720p", lambda: scenes.config3_street(1280, 720)),
    # cameras a hand above the ground: triangles through the near plane and across the screen, in the SECOND pass
    ("street_ground_720p", lambda: (scenes.config3_street(1280, 720)[0], scenes.Camera((-62.0, 0.25, 3.0), (1.0, -0.02, -0.04), 1280, 720))),
    ("street_wall_4k", lambda: (scenes.config3_street()[0], scenes.Camera((-55.0, 1.2, 7.5), (0.9, 0.1, -0.4), 3840, 2160))),
    ("floor", lambda: scenes.floor_under_camera((0.3, 0.25, 0.2), (0.1, -0.6, -1.0), 256, 192)),
]


@pytest.mark.parametrize("name,builder", SECOND_PASS, ids=[b[0] for b in SECOND_PASS])
def test_frame_whose_second_pass_holds_the_scene_matches_oracle(gpu, name, builder):
    """A frame whose SECOND raster pass holds nearly the whole scene (the usual second pass is a few per cent of it): every
    object 'was' 500 m further down the view direction in the previous frame, so phase 0 (previous transforms against the
    previous HZB) rejects what that HZB covers and phase 1 brings it back -- clipped and screen-filling triangles, long bins
    and the read-modify-write of every tile then happen in the pass that merges into a finished image.  Image, HZB chain and
    counts against the oracle, then three static frames."""
    from chord_amd import lib as L
    scene, cam = builder()
    W, Hh = cam.width, cam.height
    flags = H.ALL_FLAGS
    view0, iv0 = L.make_views(cam)
    L.fill_objects(scene, cam)
    want0 = orc.frame(scene, view0, iv0, flags)
    r = _renderer(gpu, scene, view0, iv0, W, Hh, flags)
    r.update_objects(scene.objects)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want0["vis"], W, Hh, name + " frame 0")

    f = np.array(cam.front, dtype=np.float64); f /= np.linalg.norm(f)
    last = scene.local_to_world.copy()
    last[:, 12:15] += 500.0 * f                                   # glm column-major: the translation column
    L.fill_objects(scene, cam, cam, last)
    view1, iv1 = L.make_views(cam, view0)
    want1 = orc.frame(scene, view1, iv1, flags, prev_hzb_min=want0["hzb_min"])
    assert want1["counts"][3] > 0 and want1["counts"][3] >= want1["counts"][1]      # the second pass is the bigger one
    r.update_objects(scene.objects)
    r.set_view(view1, iv1, flags)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want1["vis"], W, Hh, name + " frame 1 (second pass holds the scene)")
    st = r.stats()
    assert st["rasterLaunches"] == 2 and st["overflow"] == 0
    assert [st["countInstanceCulled"], st["countStage0Visible"], st["countStage0Rejected"], st["countStage1Visible"]] == list(want1["counts"])
    assert st["trianglesSubmitted"] == want1["stats"].trianglesSubmitted
    mn, mx, rng = r.read_hzb(r.history_hzb())
    assert np.array_equal(mn, want1["hzb_min"]) and np.array_equal(mx, want1["hzb_max"]) and np.array_equal(rng, want1["valid_range"])

    L.fill_objects(scene, cam, cam)
    view2, iv2 = L.make_views(cam, view1)
    r.update_objects(scene.objects)
    r.set_view(view2, iv2, flags)
    prev = want1["hzb_min"]
    for k in range(3):
        want = orc.frame(scene, view2, iv2, flags, prev_hzb_min=prev)
        r.render_frame()
        H.assert_vis_equal(r.read_visibility(), want["vis"], W, Hh, name + " static frame %d" % k)
        st = r.stats()
        assert st["countStage1Visible"] == want["counts"][3] and st["overflow"] == 0
        mn, mx, rng = r.read_hzb(r.history_hzb())
        assert np.array_equal(mn, want["hzb_min"]) and np.array_equal(mx, want["hzb_max"]) and np.array_equal(rng, want["valid_range"])
        prev = want["hzb_min"]
    r.close()


HIER = [
    ("small", lambda: scenes.small_test_scene(320, 200, seed=13)),
    ("masked", lambda: scenes.masked_test_scene(320, 200)),
    ("street_720p", lambda: scenes.config3_street(1280, 720)),
    ("street_4k", scenes.config3_street),
    ("street_x16_1080p", lambda: scenes.config4_street_x64(1920, 1080, grid=4)),
    ("subpixel_small", lambda: scenes.config5_subpixel(960, 540, prims=16, patches_per_prim=256, instances=4)),
]


@pytest.mark.parametrize("name,builder", HIER, ids=[h[0] for h in HIER])
def test_hierarchical_cull_builds_the_same_command_list_and_frames(gpu, name, builder):
    """chordvis_set_cull_mode(1): resident waves walk the primitives' GPUBVHNode trees (which the reference builds and
    never reads) and drop subtrees by their parent-error spheres; the command ARRAY (slots included) must equal the flat
    dispatch's -- i.e. the oracle's -- from several distances, and so must two frames with two-pass occlusion culling."""
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityRenderer
    scene, cam0 = builder()
    assert scene.bvh_nodes is not None and len(scene.bvh_nodes) >= len(scene.primitives)
    w, h, flags = cam0.width, cam0.height, H.ALL_FLAGS
    r = VisibilityRenderer(0)
    r.set_cull_mode(1)
    r.upload_scene(scene)
    r.allocate_gbuffer(w, h)
    f = np.array(cam0.front, dtype=np.float64)
    f /= np.linalg.norm(f)
    for back in (0.0, 12.0, 60.0, 400.0):                        # farther away: coarser LODs, more subtrees dropped
        cam = cam0.moved(tuple(-back * f))
        L.fill_objects(scene, cam)
        view, iv = L.make_views(cam)
        r.update_objects(scene.objects)
        r.set_view(view, iv, flags)
        got = r.read_cmds(r.instance_culling())
        want = orc.instance_culling(scene, view, iv, flags)
        assert len(got) == len(want) and np.array_equal(got, want), "%s at -%g m: %d vs %d commands" % (name, back, len(got), len(want))
    L.fill_objects(scene, cam0)
    view, iv = L.make_views(cam0)
    r.update_objects(scene.objects)
    r.set_view(view, iv, flags)
    r.reset_history()
    prev = None
    for frame in range(2):
        r.render_frame()
        want = orc.frame(scene, view, iv, flags, prev_hzb_min=prev)
        H.assert_vis_equal(r.read_visibility(), want["vis"], w, h, "%s frame %d" % (name, frame))
        prev = want["hzb_min"]
    r.close()


def test_hierarchical_cull_is_refused_without_trees(gpu):
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityRenderer
    scene, cam = scenes.small_test_scene(160, 96)
    bare = R.Scene(scene.objects, scene.primitives, scene.materials, scene.meshlets, scene.groups, scene.group_indices,
                   scene.meshlet_data, scene.positions)
    r = VisibilityRenderer(0)
    r.upload_scene(bare)
    with pytest.raises(L.ChordvisError):
        r.set_cull_mode(1)
    # a tree whose sphere does not bound its groups is rejected at upload
    bad = scene.bvh_nodes.copy()
    k = int(np.argmax((np.arange(len(bad)) > 0) & (bad["leafMeshletGroupCount"] > 0)))
    bad["sphere"][k, 3] *= 0.25
    broken = R.Scene(scene.objects, scene.primitives, scene.materials, scene.meshlets, scene.groups, scene.group_indices,
                     scene.meshlet_data, scene.positions, bvh_nodes=bad)
    with pytest.raises(L.ChordvisError):
        r.upload_scene(broken)
    r.close()
