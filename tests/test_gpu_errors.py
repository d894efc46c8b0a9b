"""Error behaviour of the C ABI on the device: a scene the kernels could not index safely is refused at
upload, a frame that overflowed its work lists says so, and the context stays usable after either.  The reference asserts
(check() / checkVkResult(), utils.h:57-72); a C ABI that takes caller buffers returns codes instead."""
import os

import numpy as np
import pytest

import helpers as H
import orc
from chord_amd import lib as L, records as R, scenes

pytestmark = pytest.mark.gpu


def _clone(scene, **over):
    f = dict(objects=scene.objects, primitives=scene.primitives, materials=scene.materials, meshlets=scene.meshlets,
             groups=scene.groups, group_indices=scene.group_indices, meshlet_data=scene.meshlet_data, positions=scene.positions)
    f.update({k: v for k, v in over.items() if k in f})
    return R.Scene(f["objects"], f["primitives"], f["materials"], f["meshlets"], f["groups"], f["group_indices"], f["meshlet_data"],
                   f["positions"], name=scene.name + "_broken", texcoord0=over.get("texcoord0", scene.texcoord0),
                   textures=over.get("textures", scene.texture_images), samplers=over.get("samplers", scene.samplers),
                   bvh_nodes=over.get("bvh_nodes", scene.bvh_nodes))


def _broken_scenes(scene):
    """(what is wrong, scene) pairs: each breaks one thing a kernel would index with."""
    m0 = scene.meshlets[0]
    V, T, off = int(m0["vertexTriangleCount"]) & 0xFF, (int(m0["vertexTriangleCount"]) >> 8) & 0xFF, int(m0["dataOffset"])
    out = []
    d = scene.meshlet_data.copy(); d[off] = 0x7FFFFFFF
    out.append(("vertex id past the position buffer", _clone(scene, meshlet_data=d)))
    d = scene.meshlet_data.copy(); d[off + V] = (d[off + V] & 0xFFFFFF00) | 0xFF if V < 255 else d[off + V]
    if V < 255:
        out.append(("triangle corner >= the meshlet's vertex count", _clone(scene, meshlet_data=d)))
    m = scene.meshlets.copy(); m["dataOffset"][len(m) - 1] = len(scene.meshlet_data) - 1
    out.append(("meshlet data running past the buffer", _clone(scene, meshlets=m)))
    gi = scene.group_indices.copy(); gi[0] = len(scene.meshlets) + 5
    out.append(("group naming a meshlet that does not exist", _clone(scene, group_indices=gi)))
    g = scene.groups.copy(); g["meshletCount"][0] = 9
    out.append(("more than 4 meshlets in a group", _clone(scene, groups=g)))
    o = scene.objects.copy(); o["GLTFPrimitiveDetail"][0] = len(scene.primitives)
    out.append(("object naming a primitive that does not exist", _clone(scene, objects=o)))
    return out


def test_scenes_the_kernels_could_not_index_are_refused_and_the_context_survives(gpu):
    from chord_amd.renderer import VisibilityRenderer
    scene, cam, view, iv = H.setup_scene(lambda: scenes.small_test_scene(160, 96))
    want = orc.frame(scene, view, iv, H.ALL_FLAGS)
    r = VisibilityRenderer(0)
    with pytest.raises(L.ChordvisError):
        r.render_frame()                                      # nothing uploaded yet
    for what, bad in _broken_scenes(scene):
        with pytest.raises(L.ChordvisError):
            r.upload_scene(bad)
        assert len(r.last_error()) > 0, what
    # the same context takes the intact scene afterwards and renders it exactly
    r.upload_scene(scene)
    r.allocate_gbuffer(cam.width, cam.height)
    r.set_view(view, iv, H.ALL_FLAGS)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want["vis"], cam.width, cam.height, "after refused uploads")
    assert r.stats()["overflow"] == 0
    r.close()


def test_a_masked_material_without_its_texture_samples_white(gpu):
    """A masked material whose base-colour texture is not supplied is not an error: it samples the white fallback
    (alpha 1, the reference's default texture), so only baseColorFactor.w and the cutoff decide -- on both sides."""
    from chord_amd.renderer import VisibilityRenderer
    scene, cam = scenes.masked_test_scene(320, 200)
    L.fill_objects(scene, cam)                              # (object transforms for this camera, shared with the clone)
    bare = _clone(scene, textures=[])
    view, iv = L.make_views(cam)
    want = orc.frame(bare, view, iv, H.ALL_FLAGS)
    r = VisibilityRenderer(0)
    r.upload_scene(bare)
    r.allocate_gbuffer(cam.width, cam.height)
    r.set_view(view, iv, H.ALL_FLAGS)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want["vis"], cam.width, cam.height, "masked scene without textures")
    r.close()


def test_work_list_exhaustion_is_reported_and_recoverable(gpu):
    """Too small a record budget (chordvis_set_limits) for the scene: the frame is incomplete and chordvis_stats says
    CHORDVIS_E_CAPACITY -- nothing is written past a list -- and the same context then renders a scene that fits exactly."""
    from chord_amd.renderer import VisibilityRenderer
    big, cam_b, view_b, iv_b = H.setup_scene(lambda: scenes.config4_street_x64(640, 360))
    r = VisibilityRenderer(0)
    r.set_limits(max_triangle_records=1 << 16)                # the smallest budget the ABI takes: 1024 records per list shard
    r.upload_scene(big)
    r.allocate_gbuffer(cam_b.width, cam_b.height)
    r.set_view(view_b, iv_b, H.ALL_FLAGS)
    r.set_debug(32768)                                        # (records, not pixel blocks: the record lists are what is small)
    r.render_frame()
    with pytest.raises(L.ChordvisError) as e:
        r.stats()
    assert "(-4)" in str(e.value), str(e.value)               # CHORDVIS_E_CAPACITY
    r.render_frame()                                          # a second overflowing frame does not hang or fault either
    with pytest.raises(L.ChordvisError):
        r.stats()
    small, cam, view, iv = H.setup_scene(lambda: scenes.small_test_scene(160, 96))
    want = orc.frame(small, view, iv, H.ALL_FLAGS)
    r.upload_scene(small)
    r.allocate_gbuffer(cam.width, cam.height)
    r.set_view(view, iv, H.ALL_FLAGS)
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), want["vis"], cam.width, cam.height, "after overflowed frames")
    assert r.stats()["overflow"] == 0
    r.close()


def test_measurement_switches_are_refused_by_the_product_library(gpu):
    """The phase clocks and ablation switches of the raster kernels exist only in libraries built with -DRASTER_PROFILE=1 /
    -DRASTER_ABLATION=1; the product library says so instead of silently measuring itself.  The switches that do not change
    results (pixel blocks never / always, hot-tile variant, HZB tail as a launch of its own) stay available."""
    from chord_amd.renderer import VisibilityRenderer
    if os.environ.get("CHORDVIS_LIB"):
        pytest.skip("a variant library is loaded")
    r = VisibilityRenderer(0)
    for bit in (1, 2, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384):
        with pytest.raises(L.ChordvisError) as e:
            r.set_debug(bit)
        assert "(-1)" in str(e.value) and "built with" in str(e.value), str(e.value)        # CHORDVIS_E_INVALID
    for bits in (32768, 65536, 65536 | 262144, 131072, 0):
        r.set_debug(bits)
    r.close()
