"""Masked (alpha-tested) and blended materials: the oracle's canonical texture fetch against an independent numpy
restatement, level selection on hand-made triangles, and frame-level properties (mesh_raster.hlsl:34-38,107-112,
198-204; mesh_raster.cpp:224-252).  CPU only."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
import orc
from chord_amd import records as R
from chord_amd import scenes


def _texture(img):
    chain, mips = R.mip_chain_rgba8(img)
    return chain, R.Texture(chain.ctypes.data, img.shape[1], img.shape[0], mips, 0), mips


def _wrap_np(i, n, mode):
    i = np.asarray(i, dtype=np.int64)
    if mode == R.WRAP_CLAMP_TO_EDGE:
        return np.clip(i, 0, n - 1)
    if mode == R.WRAP_MIRRORED_REPEAT:
        m = np.mod(i, 2 * n)
        return np.where(m < n, m, 2 * n - 1 - m)
    return np.mod(i, n)


def _sample_np(levels, level, linear, wrap_s, wrap_t, u, v):
    """Independent restatement in numpy float32 of oracle.c header item 9 (sampling part)."""
    a = levels[level][..., 3].astype(np.float32) * np.float32(1.0 / 255.0)
    h, w = a.shape
    u, v = np.float32(u), np.float32(v)
    if not linear:
        ix = _wrap_np(np.floor(u * np.float32(w)), w, wrap_s)
        iy = _wrap_np(np.floor(v * np.float32(h)), h, wrap_t)
        return a[iy, ix]
    x, y = u * np.float32(w) - np.float32(0.5), v * np.float32(h) - np.float32(0.5)
    x0, y0 = np.floor(x), np.floor(y)
    fx, fy = np.float32(x - x0), np.float32(y - y0)
    ix0, ix1 = _wrap_np(x0, w, wrap_s), _wrap_np(x0 + 1, w, wrap_s)
    iy0, iy1 = _wrap_np(y0, h, wrap_t), _wrap_np(y0 + 1, h, wrap_t)
    top = np.float32(a[iy0, ix0] + np.float32(np.float32(a[iy0, ix1] - a[iy0, ix0]) * fx))
    bot = np.float32(a[iy1, ix0] + np.float32(np.float32(a[iy1, ix1] - a[iy1, ix0]) * fx))
    return np.float32(top + np.float32(np.float32(bot - top) * fy))


def _levels(img):
    chain, mips = R.mip_chain_rgba8(img)
    out, off, h, w = [], 0, img.shape[0], img.shape[1]
    for l in range(mips):
        lw, lh = max(1, w >> l), max(1, h >> l)
        out.append(chain[off:off + lw * lh * 4].reshape(lh, lw, 4))
        off += lw * lh * 4
    assert off == len(chain)
    return out


@pytest.mark.parametrize("shape", [(64, 64), (21, 37), (1, 9), (16, 16)])
def test_texture_fetch_matches_numpy_restatement(shape):
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    img = rng.integers(0, 256, size=shape + (4,), dtype=np.uint8)
    chain, tex, mips = _texture(img)
    levels = _levels(img)
    assert [l.shape[:2] for l in levels][-1] == (1, 1) and mips == len(levels)
    for wrap_s in (R.WRAP_REPEAT, R.WRAP_CLAMP_TO_EDGE, R.WRAP_MIRRORED_REPEAT):
        for wrap_t in (R.WRAP_REPEAT, R.WRAP_MIRRORED_REPEAT):
            smp = np.array([(R.FILTER_LINEAR, R.FILTER_LINEAR, wrap_s, wrap_t)], dtype=R.SAMPLER)
            for level in range(mips):
                for linear in (0, 1):
                    for _ in range(40):
                        u, v = (rng.random(2) * 6.0 - 3.0).astype(np.float32)
                        if rng.random() < 0.2:                      # exactly on texel centres / edges
                            lw, lh = levels[level].shape[1], levels[level].shape[0]
                            u = np.float32(rng.integers(-2 * lw, 2 * lw) / lw)
                            v = np.float32((rng.integers(-2 * lh, 2 * lh) + 0.5) / lh)
                        got = orc.lib.orc_sample_alpha(C.byref(tex), smp.ctypes.data, level, linear, float(u), float(v))
                        want = _sample_np(levels, level, linear, wrap_s, wrap_t, u, v)
                        assert np.float32(got) == np.float32(want), (shape, wrap_s, wrap_t, level, linear, u, v, got, want)
    # no texture: the reference's white fallback
    assert orc.lib.orc_sample_alpha(None, smp.ctypes.data, 0, 1, 0.3, 0.7) == 1.0


def test_mip_chain_is_a_2x2_box_filter():
    img = np.zeros((4, 6, 4), np.uint8)
    img[..., 3] = np.arange(24).reshape(4, 6) * 10
    lv = _levels(img)
    assert [l.shape[:2] for l in lv] == [(4, 6), (2, 3), (1, 1)]
    a0 = img[..., 3].astype(np.int64)
    want1 = (a0[0::2, 0::2] + a0[1::2, 0::2] + a0[0::2, 1::2] + a0[1::2, 1::2] + 2) // 4
    assert np.array_equal(lv[1][..., 3], want1)


def test_level_is_half_the_exponent_of_the_texel_to_pixel_area_ratio():
    scene, cam = scenes.masked_test_scene(64, 64)
    mat = scene.materials[1:2].copy()                                  # the 64 x 64 checker, LINEAR_MIPMAP_LINEAR / LINEAR
    lin = C.c_int(-1)

    def level(px_side, uv_side):
        # right triangle with legs px_side pixels and uv_side (in uv units): doubled areas px_side^2 * 65536, uv_side^2
        u = np.array([0.0, uv_side, 0.0], np.float32)
        v = np.array([0.0, 0.0, uv_side], np.float32)
        a2 = int(px_side * 256) * int(px_side * 256)
        return orc.lib.orc_mask_level(C.byref(scene.desc), mat.ctypes.data, a2, u.ctypes.data, v.ctypes.data, C.byref(lin)), lin.value

    # 64 texels per uv unit: uv_side 1 over 64 px is 1 texel per pixel -> ratio 1 -> level 0 with the MIN filter
    assert level(64, 1.0) == (0, 1)
    assert level(128, 1.0) == (0, 1) and level(65, 1.0)[0] == 0        # magnified: level 0, mag filter (linear here)
    assert level(32, 1.0) == (1, 1)                                     # 2 texels per pixel
    assert level(16, 1.0) == (2, 1) and level(23, 1.0)[0] == 1          # ratio 7.7 -> exponent 2 -> level 1
    assert level(1, 1.0) == (6, 1) and level(0.25, 4.0)[0] == 6         # clamped to the last level (64 -> 7 levels)
    nearest = scene.materials[2:3].copy()                               # NEAREST / NEAREST sampler
    u = np.array([0.0, 1.0, 0.0], np.float32); v = np.array([0.0, 0.0, 1.0], np.float32)
    assert orc.lib.orc_mask_level(C.byref(scene.desc), nearest.ctypes.data, 4 * 65536, u.ctypes.data, v.ctypes.data, C.byref(lin)) >= 1 and lin.value == 0
    # degenerate texture coordinates: ratio 0 -> magnification path
    z = np.zeros(3, np.float32)
    assert orc.lib.orc_mask_level(C.byref(scene.desc), mat.ctypes.data, 65536, z.ctypes.data, z.ctypes.data, C.byref(lin)) == 0


def _frame(scene, view, iv, mats=None, **kw):
    sc = scene if mats is None else scene.with_objects(None, mats)
    return orc.frame(sc, view, iv, H.ALL_FLAGS, **kw)


def test_masked_frame_properties():
    scene, cam, view, iv = H.setup_scene(lambda: scenes.masked_test_scene(256, 160))
    base = _frame(scene, view, iv)
    st = base["stats"]
    assert st.fragmentsClipped > 0 and st.fragments > 0
    masked = scene.materials["alphaMode"] == R.ALPHA_MASK

    # (1) a fully opaque alpha channel: the masked buckets draw exactly what the opaque ones would
    white = [t.copy() for t in scene.texture_images]
    for t in white:
        t[..., 3] = 255
    mats = scene.materials.copy()
    mats["baseColorFactor"][:, 3] = 1.0
    full = R.Scene(scene.objects, scene.primitives, mats, scene.meshlets, scene.groups, scene.group_indices, scene.meshlet_data,
                   scene.positions, texcoord0=scene.texcoord0, textures=white, samplers=scene.samplers)
    opaque_mats = mats.copy()
    opaque_mats["alphaMode"][masked] = R.ALPHA_OPAQUE
    assert np.array_equal(_frame(full, view, iv)["vis"], _frame(full, view, iv, opaque_mats)["vis"])

    # (2) alpha 0 everywhere == blended == nothing drawn for those objects
    black = [t.copy() for t in scene.texture_images]
    for t in black:
        t[..., 3] = 0
    none = R.Scene(scene.objects, scene.primitives, scene.materials, scene.meshlets, scene.groups, scene.group_indices,
                   scene.meshlet_data, scene.positions, texcoord0=scene.texcoord0, textures=black, samplers=scene.samplers)
    textured = masked & (scene.materials["baseColorId"] < len(scene.texture_images))
    blend_mats = scene.materials.copy()
    blend_mats["alphaMode"][textured] = R.ALPHA_BLEND
    assert np.array_equal(_frame(none, view, iv)["vis"], _frame(scene, view, iv, blend_mats)["vis"])

    # (3) a masked surface only ever REMOVES fragments: where its id survives the word equals the opaque frame's, and
    #     every pixel of the masked frame is at most as near as the opaque frame's
    all_opaque = scene.materials.copy()
    all_opaque["alphaMode"][masked] = R.ALPHA_OPAQUE
    op = _frame(scene, view, iv, all_opaque)["vis"]
    assert (base["vis"] <= op).all() and (base["vis"] != op).any()

    # (4) two-pass occlusion culling leaves a static masked image unchanged
    again = _frame(scene, view, iv, prev_hzb_min=base["hzb_min"])
    assert np.array_equal(again["vis"], base["vis"])

    # (5) blended objects are culled and listed like any other (they are only absent from the raster buckets)
    cmds = orc.instance_culling(scene, view, iv, H.ALL_FLAGS)
    blend_objs = np.nonzero(scene.materials["alphaMode"][scene.objects["GLTFMaterialData"]] == R.ALPHA_BLEND)[0]
    assert len(blend_objs) and np.isin(cmds["objectId"], blend_objs).any()
    slots = ((base["vis"] >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.int64) - 1
    drawn_objs = np.unique(cmds["objectId"][slots[slots >= 0]])
    assert not np.isin(drawn_objs, blend_objs).any()


def test_wrap_remainder_by_multiply_high_is_exact():
    """The tile kernel wraps a texel index of a non-power-of-two level with constants resolved at upload (chordvis_abi.cpp
    wrap_consts; kernels_raster.hip period_mod): magic = floor(2^32 / period), bias = the first multiple of the period >= 2^30;
    r = (i + bias) - mulhi(i + bias, magic) * period, minus the period once if r >= period.  Restated here in numpy over every
    period a level can have (sizes 1..16384, doubled for MIRRORED_REPEAT) at the index range texel_floor produces (|i| <= 1e9,
    and the +1 neighbour of the bilinear fetch), against the mathematical remainder."""
    rng = np.random.default_rng(11)
    periods = np.unique(np.concatenate([np.arange(3, 300), rng.integers(3, 32769, size=4000), [32767, 32766, 24576, 16383, 12289, 3 * 5461]]))
    periods = periods[(periods & (periods - 1)) != 0].astype(np.uint64)                 # powers of two wrap with a mask
    edge = np.array([-1000000001, -1000000000, -999999999, -32769, -3, -2, -1, 0, 1, 2, 3, 32767, 999999999, 1000000000, 1000000001], dtype=np.int64)
    for period in periods:
        magic = np.uint64(0x100000000) // period
        bias = ((np.uint64(0x40000000) + period - np.uint64(1)) // period) * period
        assert magic < 2**32 and bias < 2**31 and bias % period == 0
        i = np.concatenate([edge, rng.integers(-1000000001, 1000000002, size=256), np.arange(-int(period) - 2, 2 * int(period) + 3)]).astype(np.int64)
        iu = (i + np.int64(bias)).astype(np.uint64)
        assert (iu < 2**31).all()
        q = (iu * magic) >> np.uint64(32)                                                 # v_mul_hi_u32
        r = (iu - q * period) & np.uint64(0xFFFFFFFF)                                     # 32-bit wrap-around arithmetic
        r = np.where(r >= period, r - period, r)
        assert np.array_equal(r.astype(np.int64), np.mod(i, np.int64(period))), int(period)
