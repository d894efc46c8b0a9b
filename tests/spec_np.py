"""An independent restatement of SURVEY.md Appendix A, stages A1-A4 (object cull, group LOD cut + meshlet cull, HZB
occlusion test, HZB build), written from that text in vectorised numpy float32 -- not from oracle/oracle.c, whose
structure (scalar loops, per-command) it deliberately does not share.  tests/test_spec_np.py requires both to agree
bit for bit; a slip in either restatement of the HLSL shows up as a difference.  TEST INFRASTRUCTURE.

Arithmetic: numpy float32 array operations round every + - * / sqrt separately (no contraction), which is the canonical
arithmetic of SURVEY 8c.  Matrices are stored glm column-major: M[r][c] = m[c * 4 + r]."""
import numpy as np

f32 = np.float32
# kExtentApplyFactor (base.hlsli:184-194)
EXTENT = np.array([[1, 1, 1], [-1, -1, -1], [1, 1, -1], [1, -1, 1], [-1, 1, 1], [1, -1, -1], [-1, -1, 1], [-1, 1, -1]], dtype=f32)


def mat(m16):
    """(..., 16) glm column-major -> (..., 4, 4) indexed [r][c]"""
    return np.swapaxes(np.asarray(m16, dtype=f32).reshape(m16.shape[:-1] + (4, 4)), -1, -2)


def mul_mm(A, B):
    """mul(A, B)[r][c] = ((A[r][0]*B[0][c] + A[r][1]*B[1][c]) + A[r][2]*B[2][c]) + A[r][3]*B[3][c]"""
    out = np.empty(np.broadcast_shapes(A.shape, B.shape), dtype=f32)
    for r in range(4):
        for c in range(4):
            out[..., r, c] = ((A[..., r, 0] * B[..., 0, c] + A[..., r, 1] * B[..., 1, c]) + A[..., r, 2] * B[..., 2, c]) + A[..., r, 3] * B[..., 3, c]
    return out


def mul_mv(M, x, y, z, w=f32(1.0)):
    """mul(M, v) rows: ((M[r][0]*v0 + M[r][1]*v1) + M[r][2]*v2) + M[r][3]*v3  -> (..., 4)"""
    return np.stack([((M[..., r, 0] * x + M[..., r, 1] * y) + M[..., r, 2] * z) + M[..., r, 3] * w for r in range(4)], axis=-1)


def dot3(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def corners(pmin, pmax):
    """c = (posMin + posMax) * 0.5, e = posMax - c, the eight corners c + e * s_k  -> (..., 8, 3)"""
    c = (pmin + pmax) * f32(0.5)
    e = pmax - c
    return c[..., None, :] + e[..., None, :] * EXTENT


def project_uvz(p, M):
    """projectPosToUVz: clip / w, then xy * (0.5, -0.5) + 0.5"""
    h = mul_mv(M, p[..., 0], p[..., 1], p[..., 2])
    x, y, z = h[..., 0] / h[..., 3], h[..., 1] / h[..., 3], h[..., 2] / h[..., 3]
    return np.stack([x * f32(0.5) + f32(0.5), y * f32(-0.5) + f32(0.5), z], axis=-1)


def box_culled(planes, pmin, pmax, M, MVP):
    """A1's test of one box per row: True = culled.  Orthographic (MVP[3][3] == 1): culled iff the projected rectangle misses
    the unit square; else iff all eight corners are behind one of the six planes (dot(n, p) > -d means in front)."""
    P = corners(pmin, pmax)                                                   # (N, 8, 3)
    ortho = MVP[..., 3, 3] == f32(1.0)
    with np.errstate(all="ignore"):
        uvz = project_uvz(P, MVP[..., None, :, :])
        mn, mx = uvz.min(axis=-2), uvz.max(axis=-2)
        mn = np.minimum(mn, f32(10.0)); mx = np.maximum(mx, f32(-10.0))       # the shader's running min / max start at +-10
        ortho_culled = (mn[..., 0] >= 1) | (mn[..., 1] >= 1) | (mx[..., 0] <= 0) | (mx[..., 1] <= 0)
    h = mul_mv(M[..., None, :, :], P[..., 0], P[..., 1], P[..., 2])[..., :3]  # corners in relative world space
    d = (planes[None, :, None, 0] * h[:, None, :, 0] + planes[None, :, None, 1] * h[:, None, :, 1]) + planes[None, :, None, 2] * h[:, None, :, 2]
    behind_all = (~(d > -planes[None, :, None, 3])).all(axis=-1)              # (N, 6)
    persp_culled = behind_all.any(axis=-1)
    return np.where(ortho, ortho_culled, persp_culled)


def projected_error(K, l2v, scale, center, radius):
    """px(center, r) of A2: -1 when the eye is inside the (scaled) sphere, else K * R / sqrt(d2 - R^2)"""
    q = mul_mv(l2v, center[..., 0], center[..., 1], center[..., 2])
    with np.errstate(all="ignore"):                                           # (un-parented groups carry FLT_MAX radii)
        R = scale * radius
        d2 = dot3(q, q)
        r2 = R * R
        px = K * R / np.sqrt(d2 - r2)
    return np.where(d2 <= r2, f32(-1.0), px)


def instance_culling(scene, view, iv, flags):
    """A1 + A2: the command list (objectId, meshletId, slot) in (object, group, meshlet-in-group) order."""
    O = scene.objects
    prim = scene.primitives[O["GLTFPrimitiveDetail"]]
    two_sided = scene.materials["bTwoSided"][O["GLTFMaterialData"]] != 0
    M = mat(O["localToTranslatedWorld"])
    VP = mat(iv["translatedWorldToClip"])[0]
    MVP = mul_mm(VP, M)
    planes = iv["frustumPlanesRS"][0].astype(f32)
    obj_vis = np.ones(len(O), bool)
    if flags & 1:
        obj_vis = ~box_culled(planes, prim["posMin"].astype(f32), prim["posMax"].astype(f32), M, MVP)
    # flatten (object, group)
    counts = np.where(obj_vis, prim["meshletGroupCount"], 0).astype(np.int64)
    owner = np.repeat(np.arange(len(O)), counts)
    if len(owner) == 0:
        return np.zeros(0, dtype=[("objectId", np.uint32), ("meshletId", np.uint32), ("slot", np.uint32)])
    gi = np.arange(len(owner)) - np.repeat(np.cumsum(counts) - counts, counts)
    g = scene.groups[prim["meshletGroupOffset"][owner].astype(np.int64) + gi]
    V = mat(view["translatedWorldToView"])[0]
    l2v = mul_mm(V, M)[owner]
    scale = O["scaleExtractFromMatrix"][owner, 3].astype(f32)
    K = f32(view["lodScale"][0])
    pe = projected_error(K, l2v, scale, g["parentPosCenter"].astype(f32), g["parentError"].astype(f32))
    er = projected_error(K, l2v, scale, g["clusterPosCenter"].astype(f32), g["error"].astype(f32))
    parent_ok = (g["parentError"] > f32(3e38)) | ~((pe > 0) & (pe <= 1))
    self_ok = (g["error"] < f32(-0.5)) | ((er >= 0) & (er <= 1))
    keep = parent_ok & self_ok
    owner, g = owner[keep], g[keep]
    # meshlets of the kept groups (<= 4 each), in group order
    mc = g["meshletCount"].astype(np.int64)
    gowner = np.repeat(owner, mc)
    k = np.arange(len(gowner)) - np.repeat(np.cumsum(mc) - mc, mc)
    p2 = scene.primitives[O["GLTFPrimitiveDetail"][gowner]]
    idx = p2["meshletGroupIndicesOffset"].astype(np.int64) + np.repeat(g["meshletOffset"].astype(np.int64), mc) + k
    mid = p2["meshletOffset"].astype(np.int64) + scene.group_indices[idx]
    m = scene.meshlets[mid]
    vis = np.ones(len(mid), bool)
    if flags & 4:                                                             # cone (one-sided materials only)
        W2L = mat(O["translatedWorldToLocal"])[gowner]
        cam = mul_mv(W2L, f32(0), f32(0), f32(0))[..., :3]
        v = m["coneApex"].astype(f32) - cam
        with np.errstate(all="ignore"):
            n = v / np.sqrt(dot3(v, v))[..., None]
            cone_culled = dot3(n, m["coneAxis"].astype(f32)) >= m["coneCutOff"].astype(f32)
        vis &= ~(cone_culled & ~two_sided[gowner])
    if flags & 1:
        vis &= ~box_culled(planes, m["posMin"].astype(f32), m["posMax"].astype(f32), M[gowner], MVP[gowner])
    out = np.zeros(int(vis.sum()), dtype=[("objectId", np.uint32), ("meshletId", np.uint32), ("slot", np.uint32)])
    out["objectId"], out["meshletId"] = gowner[vis], mid[vis]
    out["slot"] = np.arange(len(out))
    return out


def f16_bits(x):
    return np.asarray(x, dtype=f32).astype(np.float16).view(np.uint16)       # IEEE round-to-nearest-even


def hzb_layout(W, H):
    """A4 / hzb.cpp:49-63: mip-0 extent, level count, offsets"""
    def npot(v):
        return 1 << (int(v) - 1).bit_length()
    w0, h0 = npot(W) // 2, npot(H) // 2
    if w0 == W:
        w0 //= 2
    if h0 == H:
        h0 //= 2
    w0, h0 = max(w0, 1), max(h0, 1)
    levels = int(np.floor(np.log2(max(w0, h0)))) + 1
    dims = [(max(1, w0 >> l), max(1, h0 >> l)) for l in range(levels)]
    offs = np.concatenate([[0], np.cumsum([w * h for w, h in dims])])
    return dims, offs


def hzb_build(depth, W, H, want_max=False):
    """A4: texel (x, y) of mip l = min (max) over the 2^(l+1) square of edge-clamped source depth, as binary16; max chain
    +1 ulp from mip 5 up.  Only texels inside the image are defined; returned per level as (min, max) float16-bit arrays
    cut to the valid extent ((W-1)>>1>>l)+1."""
    depth = np.asarray(depth, dtype=f32).reshape(H, W)
    dims, offs = hzb_layout(W, H)
    out = []
    for l, (mw, mh) in enumerate(dims):
        s = 2 ** (l + 1)
        vw, vh = min(mw, (((W - 1) >> 1) >> l) + 1), min(mh, (((H - 1) >> 1) >> l) + 1)
        ys = np.minimum(np.arange(vh)[:, None] * s + np.arange(s)[None, :], H - 1)       # (vh, s) clamped rows
        xs = np.minimum(np.arange(vw)[:, None] * s + np.arange(s)[None, :], W - 1)
        blk = depth[ys[:, None, :, None], xs[None, :, None, :]]                           # (vh, vw, s, s)
        mn = f16_bits(blk.min(axis=(2, 3)))
        mx = None
        if want_max:
            # the +1 ulp is applied when mip 5 is STORED and every coarser level reduces the stored halves
            if l < 5:
                mx = f16_bits(blk.max(axis=(2, 3)))
            elif l == 5:
                mx = (f16_bits(blk.max(axis=(2, 3))).astype(np.uint32) + 1).astype(np.uint16)
            else:
                prev = out[-1][1].view(np.float16).astype(f32)
                ph, pw = prev.shape
                yy = np.minimum(np.arange(vh)[:, None] * 2 + np.arange(2)[None, :], ph - 1)
                xx = np.minimum(np.arange(vw)[:, None] * 2 + np.arange(2)[None, :], pw - 1)
                mx = f16_bits(prev[yy[:, None, :, None], xx[None, :, None, :]].max(axis=(2, 3)))
        out.append((mn, mx))
    return dims, offs, out


def hzb_visible(scene, view, cmds, phase, hzb_levels, dims):
    """A3 for every command: True = visible.  hzb_levels[l] = float16-bit array of the valid extent of mip l (min chain)."""
    O = scene.objects[cmds["objectId"]]
    prim = scene.primitives[O["GLTFPrimitiveDetail"]]
    m = scene.meshlets[cmds["meshletId"]]
    if phase == 0:
        MVP = mul_mm(mat(view["translatedWorldToClipLastFrame"])[0], mat(O["localToTranslatedWorldLastFrame"]))
    else:
        MVP = mul_mm(mat(view["translatedWorldToClip"])[0], mat(O["localToTranslatedWorld"]))
    P = corners(m["posMin"].astype(f32), m["posMax"].astype(f32))
    with np.errstate(all="ignore"):
        uvz = project_uvz(P, MVP[:, None])
        mn = np.minimum(uvz.min(axis=1), f32(10.0)); mx = np.maximum(uvz.max(axis=1), f32(-10.0))
    in_range = (mx[:, 2] < 1) & (mn[:, 2] > 0)
    off_screen = in_range & ((mn[:, 0] >= 1) | (mn[:, 1] >= 1) | (mx[:, 0] <= 0) | (mx[:, 1] <= 0))
    W, H = f32(view["renderDimension"][0, 0]), f32(view["renderDimension"][0, 1])
    visible = ~off_screen
    test = visible & in_range
    sat = lambda a: np.minimum(np.maximum(a, f32(0)), f32(1))
    with np.errstate(all="ignore"):
        rx = (sat(mn[:, 0]) * W + f32(0.5)).astype(np.int32); ry = (sat(mn[:, 1]) * H + f32(0.5)).astype(np.int32)
        rz = (sat(mx[:, 0]) * W + f32(-0.5)).astype(np.int32); rw = (sat(mx[:, 1]) * H + f32(-0.5)).astype(np.int32)
    rx, ry = np.maximum(rx, 0), np.maximum(ry, 0)
    rz = np.minimum(W - f32(1), rz.astype(f32)).astype(np.int32); rw = np.minimum(H - f32(1), rw.astype(f32)).astype(np.int32)
    empty = (rz < rx) | (rw < ry)
    for i in np.nonzero(test)[0]:
        if empty[i]:
            visible[i] = False
            continue
        x0, y0, x1, y1 = rx[i] >> 1, ry[i] >> 1, rz[i] >> 1, rw[i] >> 1
        fbh = lambda v: int(v).bit_length() - 1                                # firstbithigh, -1 for 0
        lv = max(0, max(fbh(x1 - x0), fbh(y1 - y0)) - 1)
        if ((x1 >> lv) - (x0 >> lv) >= 4) or ((y1 >> lv) - (y0 >> lv) >= 4):
            lv += 1
        cx, cy, cz, cw = x0 >> lv, y0 >> lv, x1 >> lv, y1 >> lv
        tex = hzb_levels[lv].view(np.float16)
        zmin = f32(10.0)
        for xx in range(4):
            for yy in range(4):
                zmin = min(zmin, f32(tex[min(cw, cy + yy), min(cz, cx + xx)]))
        if zmin > mx[i, 2]:
            visible[i] = False
    return visible


def hzb_visible_generic(scene, cull_view, main_camera_pos, cmds, extent_scale, use_last_frame, hzb_levels):
    """hzb_culling_generic.hlsl:37-172 for every command: True = kept.  cull_view = the InstanceCullingViewInfo whose HZB is
    tested (its matrix, render dimension and camera position); main_camera_pos = PerframeCameraView.cameraWorldPos (float64
    triple) the object matrices are relative to.  hzb_levels[l] = float16-bit min texels of mip l (valid extent)."""
    O = scene.objects[cmds["objectId"]]
    m = scene.meshlets[cmds["meshletId"]]
    l2w = mat(O["localToTranslatedWorldLastFrame" if use_last_frame else "localToTranslatedWorld"]).copy()
    view_pos = np.frombuffer(np.ascontiguousarray(cull_view["cameraWorldPos"][0]).tobytes(), dtype=np.float64)[:3]
    rel = (np.asarray(main_camera_pos, dtype=np.float64) - view_pos).astype(f32)       # float3(double3 - double3)
    for r in range(3):
        l2w[:, r, 3] = l2w[:, r, 3] + rel[r]
    MVP = mul_mm(mat(cull_view["translatedWorldToClip"])[0], l2w)
    P = corners(m["posMin"].astype(f32), m["posMax"].astype(f32))
    with np.errstate(all="ignore"):
        uvz = project_uvz(P, MVP[:, None])                                               # (n, 8, 3)
        mn = np.minimum(uvz.min(axis=1), f32(10.0)); mx = np.maximum(uvz.max(axis=1), f32(-10.0))
        can = ((uvz < 1) & (uvz > 0)).all(axis=(1, 2))                                   # every corner strictly inside the unit cube
    W, H = f32(cull_view["renderDimension"][0, 0]), f32(cull_view["renderDimension"][0, 1])
    es = f32(extent_scale)
    with np.errstate(all="ignore"):
        x0 = (mn[:, 0] * W + es * f32(-0.5)).astype(np.int32); y0 = (mn[:, 1] * H + es * f32(-0.5)).astype(np.int32)
        x1 = (mx[:, 0] * W + es * f32(0.5)).astype(np.int32); y1 = (mx[:, 1] * H + es * f32(0.5)).astype(np.int32)
    x0, y0 = np.maximum(x0, 0), np.maximum(y0, 0)
    x1 = np.minimum(W - f32(1), x1.astype(f32)).astype(np.int32); y1 = np.minimum(H - f32(1), y1.astype(f32)).astype(np.int32)
    keep = np.ones(len(cmds), dtype=bool)
    fbh = lambda v: int(v).bit_length() - 1                                              # firstbithigh, -1 for 0
    for i in np.nonzero(can)[0]:
        if x1[i] < x0[i] or y1[i] < y0[i]:
            keep[i] = False
            continue
        a, b, c, d = x0[i] >> 1, y0[i] >> 1, x1[i] >> 1, y1[i] >> 1
        lv = max(0, max(fbh(c - a), fbh(d - b)))
        if ((c >> lv) - (a >> lv) >= 2) or ((d >> lv) - (b >> lv) >= 2):
            lv += 1
        cx, cy, cz, cw = a >> lv, b >> lv, c >> lv, d >> lv
        tex = hzb_levels[lv].view(np.float16)
        zmin = f32(10.0)
        for xx in range(2):
            for yy in range(2):
                zmin = min(zmin, f32(tex[min(cw, cy + yy), min(cz, cx + xx)]))
        if zmin > mx[i, 2]:
            keep[i] = False
    return keep


def cascade_views_f64(cfg, view, light_dir, valid_range=None):
    """cascadeComputeCS (cascade_setup.hlsl:79-372) in float64 numpy, written from the shader text: per cascade the matrices
    M[r][c] `translatedWorldToClip`, `clipToTranslatedWorld`, the six frustum planes and orthoDepthConvertToView.  Not a
    bit-level statement (the reference runs it in fp32 on the GPU; chordvis_cascade_setup in fp32 on the host): the test
    compares within fp32 tolerances and treats a texel snap that rounds the other way as one texel of difference."""
    count, realtime, dim = int(cfg["cascadeCount"][0]), int(cfg["realtimeCascadeCount"][0]), int(cfg["cascadeDim"][0])
    start, end, far_end = float(cfg["cascadeStartDistance"][0]), float(cfg["cascadeEndDistance"][0]), float(cfg["farCascadeEndDistance"][0])
    lam, far_lam, rs = float(cfg["splitLambda"][0]), float(cfg["farCascadeSplitLambda"][0]), float(cfg["radiusScaleFixed"][0])
    near, far = float(view["zNear"][0]), float(view["zFar"][0])
    clip_range = far - near
    inv_zfar = np.asarray(view["clipToTranslatedWorldWithZFar_NoJitter"][0], dtype=np.float64).reshape(4, 4).T
    L = np.asarray(light_dir, dtype=np.float64); L = L / np.linalg.norm(L)

    def split(far_plane, near_plane, k, n, lam_):
        rng, ratio, p = far_plane - near_plane, far_plane / near_plane, (k + 1) / n
        d = lam_ * (near_plane * abs(ratio) ** p - (near_plane + rng * p)) + (near_plane + rng * p)
        return (d - near) / clip_range

    ndc = np.array([[-1, 1, 1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 0], [1, 1, 0], [1, -1, 0], [-1, -1, 0]], dtype=np.float64)

    def unproject(M, p):
        q = M @ np.append(p, 1.0)
        return q[:3] / q[3]
    base = np.array([unproject(inv_zfar, p) for p in ndc])
    per = []
    for k in range(count):
        if k < realtime:
            mn, mx, lam_, n, kk = near + start, near + end, lam, realtime, k
            if valid_range is not None:
                lo, hi = np.asarray(valid_range, dtype=np.uint32).view(np.float32).astype(np.float64)
                if hi > 0:
                    mn = max(mn, near / hi)
                if lo > 0:
                    mx = max(mx, mn * 1.1); mx = min(mx, mn + (end - start)); mx = min(mx, near / lo)
            s0 = mn - near
        else:
            mx, lam_, mn, s0, n, kk = near + far_end, far_lam, near + end, end, count - realtime, k - realtime
        d1 = split(mx, mn, kk, n, lam_)
        d0 = s0 / clip_range if kk == 0 else split(mx, mn, kk - 1, n, lam_)
        d1_0 = split(near, near + far_end, 0, count, far_lam)              # (the shader's argument order: far = nearZ, near = nearZ + farEnd)
        cor = np.empty((8, 3)); cor0 = np.empty((8, 3))
        for i in range(4):
            ray = base[i + 4] - base[i]
            cor[i], cor[i + 4] = base[i] + ray * d0, base[i] + ray * d1
            cor0[i], cor0[i + 4] = base[i] + ray * 0.0, base[i] + ray * d1_0
        c, c0 = cor.mean(axis=0), cor0.mean(axis=0)
        r = max(np.linalg.norm(cor - c, axis=1)); r0 = max(np.linalg.norm(cor0 - c0, axis=1))
        per.append(dict(c=c, r=r, r0=r0, R=np.ceil(r * 16.0) / 16.0, mn=mn))
    zmax = max(p["R"] for p in per) * 2.0
    out = []
    for k, p in enumerate(per):
        R_ = p["R"]
        if k >= realtime:
            radius_scale, zbias = p["r0"] / p["r"], 1.0
        else:
            radius_scale = 10.0 * rs / p["r"]; radius_scale = radius_scale / (radius_scale + 1.0)
            zbias = 0.25 + (p["mn"] - near) / (end - start)
        radius_scale = min(radius_scale, 1.0)
        eye = p["c"] - L * zmax * 0.5
        f = p["c"] - eye; f /= np.linalg.norm(f)
        s = np.cross(f, [0.0, 1.0, 0.0]); s /= np.linalg.norm(s)
        u = np.cross(s, f)
        V = np.eye(4); V[0, :3], V[1, :3], V[2, :3] = s, u, -f
        V[0, 3], V[1, 3], V[2, 3] = -np.dot(s, eye), -np.dot(u, eye), np.dot(f, eye)
        P = np.zeros((4, 4))                                               # ortho_RH_ZeroOne(-R, R, -R, R, zNear = zmax, zFar = 0): reverse Z
        P[0, 0], P[1, 1], P[2, 2], P[3, 3] = 2.0 / (2 * R_), 2.0 / (2 * R_), -1.0 / (0.0 - zmax), 1.0
        P[2, 3] = -zmax / (0.0 - zmax)
        o = (P @ V @ np.array([0, 0, 0, 1.0])) * (dim / 2.0)
        snap = (np.round(o[:2]) - o[:2]) * (2.0 / dim)
        P[0, 3] += snap[0]; P[1, 3] += snap[1]
        VP = P @ V
        inv = np.linalg.inv(VP)
        q = np.array([unproject(inv, x) for x in ndc])

        def plane(a, b, o_):
            n_ = np.cross(a - o_, b - o_); n_ /= np.linalg.norm(n_)
            return n_
        nl, nd, nr = plane(q[4], q[3], q[7]), plane(q[6], q[3], q[2]), plane(q[6], q[1], q[5])
        nt, nf, nb = plane(q[5], q[0], q[4]), plane(q[1], q[3], q[0]), plane(q[5], q[7], q[6])
        planes = np.array([np.append(nl, -np.dot(nl, q[7])), np.append(nd, -np.dot(nd, q[2])), np.append(nr, -np.dot(nr, q[5])),
                           np.append(nt, -np.dot(nt, q[4])), np.append(nf, -np.dot(nf, q[0])), np.append(nb, -np.dot(nf, q[6]))])
        out.append(dict(vp=VP, inv=inv, planes=planes, ortho=np.array([P[2, 2], P[2, 3], zbias, radius_scale]), texel=2.0 / dim, radius=R_))
    return out


# ---- cascadeComputeCS in float32, operation by operation -----------------------------------------------------------------------
def cascade_views_f32(cfg, view, light_dir, valid_range=None, tick=0, cache_valid=False):
    """cascadeComputeCS (cascade_setup.hlsl:79-372, with lookAt_RH / ortho_RH_ZeroOne / matrixInverse of base.hlsli:637-730) in
    numpy float32 SCALARS, one rounding per source operation in source order: the bit-level statement of what
    chordvis_cascade_setup must produce (tests/golden/cascade_setup.json is generated from this function).  Canonical choices
    where the shader leaves the arithmetic to the driver: no FMA contraction; `pow` is evaluated in binary64 and rounded once;
    `round` is round-half-to-even.  Returns per cascade (or None where the cache keeps the old view) a dict of
    float32 arrays: translatedWorldToClip / clipToTranslatedWorld as M[r][c], planes[6][4], ortho[4]."""
    f = np.float32
    count, realtime, dim = int(cfg["cascadeCount"][0]), int(cfg["realtimeCascadeCount"][0]), int(cfg["cascadeDim"][0])
    start, end, far_end = f(cfg["cascadeStartDistance"][0]), f(cfg["cascadeEndDistance"][0]), f(cfg["farCascadeEndDistance"][0])
    lam, far_lam, rs_fixed = f(cfg["splitLambda"][0]), f(cfg["farCascadeSplitLambda"][0]), f(cfg["radiusScaleFixed"][0])
    near, far = f(view["zNear"][0]), f(view["zFar"][0])
    clip_range = f(far - near)
    inv_zfar = np.asarray(view["clipToTranslatedWorldWithZFar_NoJitter"][0], dtype=np.float32).reshape(4, 4).T   # M[r][c]

    def mul_point(M, x, y, z, w):
        return [f(f(f(f(M[r][0] * x) + f(M[r][1] * y)) + f(M[r][2] * z)) + f(M[r][3] * w)) for r in range(4)]

    def sub(a, b): return [f(a[i] - b[i]) for i in range(3)]
    def add(a, b): return [f(a[i] + b[i]) for i in range(3)]
    def mul(a, s): return [f(a[i] * s) for i in range(3)]
    def dot(a, b): return f(f(f(a[0] * b[0]) + f(a[1] * b[1])) + f(a[2] * b[2]))
    def cross(a, b): return [f(f(a[1] * b[2]) - f(b[1] * a[2])), f(f(a[2] * b[0]) - f(b[2] * a[0])), f(f(a[0] * b[1]) - f(b[0] * a[1]))]
    def normalize(a):
        l = f(np.sqrt(dot(a, a)))
        return [f(a[0] / l), f(a[1] / l), f(a[2] / l)]

    def matmul(A, B):
        return [[f(f(f(f(A[r][0] * B[0][c]) + f(A[r][1] * B[1][c])) + f(A[r][2] * B[2][c])) + f(A[r][3] * B[3][c])) for c in range(4)] for r in range(4)]

    def inverse(M):
        # the cofactor expansion of matrixInverse (base.hlsli:684-730) on the column-major element list m[c * 4 + r]
        m = [M[i % 4][i // 4] for i in range(16)]
        def t3(a, b, c): return f(f(m[a] * m[b]) * m[c])
        def row(s):
            acc = None
            for sign, a, b, c in s:
                v = t3(a, b, c)
                if acc is None:
                    acc = v if sign > 0 else f(-v)           # ((-m[a]) * m[b]) * m[c]: the negation is exact
                else:
                    acc = f(acc + v) if sign > 0 else f(acc - v)
            return acc
        P, N = 1, -1
        inv = [None] * 16
        inv[0] = row([(P, 5, 10, 15), (N, 5, 11, 14), (N, 9, 6, 15), (P, 9, 7, 14), (P, 13, 6, 11), (N, 13, 7, 10)])
        inv[4] = row([(N, 4, 10, 15), (P, 4, 11, 14), (P, 8, 6, 15), (N, 8, 7, 14), (N, 12, 6, 11), (P, 12, 7, 10)])
        inv[8] = row([(P, 4, 9, 15), (N, 4, 11, 13), (N, 8, 5, 15), (P, 8, 7, 13), (P, 12, 5, 11), (N, 12, 7, 9)])
        inv[12] = row([(N, 4, 9, 14), (P, 4, 10, 13), (P, 8, 5, 14), (N, 8, 6, 13), (N, 12, 5, 10), (P, 12, 6, 9)])
        inv[1] = row([(N, 1, 10, 15), (P, 1, 11, 14), (P, 9, 2, 15), (N, 9, 3, 14), (N, 13, 2, 11), (P, 13, 3, 10)])
        inv[5] = row([(P, 0, 10, 15), (N, 0, 11, 14), (N, 8, 2, 15), (P, 8, 3, 14), (P, 12, 2, 11), (N, 12, 3, 10)])
        inv[9] = row([(N, 0, 9, 15), (P, 0, 11, 13), (P, 8, 1, 15), (N, 8, 3, 13), (N, 12, 1, 11), (P, 12, 3, 9)])
        inv[13] = row([(P, 0, 9, 14), (N, 0, 10, 13), (N, 8, 1, 14), (P, 8, 2, 13), (P, 12, 1, 10), (N, 12, 2, 9)])
        inv[2] = row([(P, 1, 6, 15), (N, 1, 7, 14), (N, 5, 2, 15), (P, 5, 3, 14), (P, 13, 2, 7), (N, 13, 3, 6)])
        inv[6] = row([(N, 0, 6, 15), (P, 0, 7, 14), (P, 4, 2, 15), (N, 4, 3, 14), (N, 12, 2, 7), (P, 12, 3, 6)])
        inv[10] = row([(P, 0, 5, 15), (N, 0, 7, 13), (N, 4, 1, 15), (P, 4, 3, 13), (P, 12, 1, 7), (N, 12, 3, 5)])
        inv[14] = row([(N, 0, 5, 14), (P, 0, 6, 13), (P, 4, 1, 14), (N, 4, 2, 13), (N, 12, 1, 6), (P, 12, 2, 5)])
        inv[3] = row([(N, 1, 6, 11), (P, 1, 7, 10), (P, 5, 2, 11), (N, 5, 3, 10), (N, 9, 2, 7), (P, 9, 3, 6)])
        inv[7] = row([(P, 0, 6, 11), (N, 0, 7, 10), (N, 4, 2, 11), (P, 4, 3, 10), (P, 8, 2, 7), (N, 8, 3, 6)])
        inv[11] = row([(N, 0, 5, 11), (P, 0, 7, 9), (P, 4, 1, 11), (N, 4, 3, 9), (N, 8, 1, 7), (P, 8, 3, 5)])
        inv[15] = row([(P, 0, 5, 10), (N, 0, 6, 9), (N, 4, 1, 10), (P, 4, 2, 9), (P, 8, 1, 6), (N, 8, 2, 5)])
        det = f(f(f(f(m[0] * inv[0]) + f(m[1] * inv[4])) + f(m[2] * inv[8])) + f(m[3] * inv[12]))
        inv_det = f(f(1.0) / det)
        out = [f(inv[i] * inv_det) for i in range(16)]
        return [[out[c * 4 + r] for c in range(4)] for r in range(4)]

    def log_split(far_plane, near_plane, cid, n, lam_):
        rng_, ratio = f(far_plane - near_plane), f(far_plane / near_plane)
        p = f(f(cid + 1) / f(n))
        log_scale = f(near_plane * f(np.float64(abs(ratio)) ** np.float64(p)))
        uniform_scale = f(near_plane + f(rng_ * p))
        d = f(f(lam_ * f(log_scale - uniform_scale)) + uniform_scale)
        return f(f(d - near) / clip_range)

    ndc = [(-1, 1, 1), (1, 1, 1), (1, -1, 1), (-1, -1, 1), (-1, 1, 0), (1, 1, 0), (1, -1, 0), (-1, -1, 0)]
    cs = []
    for cid in range(count):
        if cid < realtime:
            min_z, max_z, split_lam, n_split, id_split = f(near + start), f(near + end), lam, realtime, cid
            if valid_range is not None:
                vmin, vmax = np.asarray(valid_range, np.uint32).view(np.float32)
                if vmax > 0:
                    min_z = max(min_z, f(near / vmax))
                if vmin > 0:
                    stable = f(end - start)
                    max_z = max(max_z, f(min_z * f(1.1)))
                    max_z = min(max_z, f(min_z + stable))
                    max_z = min(max_z, f(near / vmin))
            split_start = f(min_z - near)
        else:
            max_z, split_lam, min_z, split_start = f(near + far_end), far_lam, f(near + end), end
            n_split, id_split = count - realtime, cid - realtime
        split = log_split(max_z, min_z, id_split, n_split, split_lam)
        prev_split = f(split_start / clip_range) if id_split == 0 else log_split(max_z, min_z, id_split - 1, n_split, split_lam)
        split0, prev_split0 = log_split(near, f(near + far_end), 0, count, far_lam), f(0.0)            # :163 (argument order as written there)
        corner = []
        for x, y, z in ndc:
            h = mul_point(inv_zfar, f(x), f(y), f(z), f(1.0))
            corner.append([f(h[0] / h[3]), f(h[1] / h[3]), f(h[2] / h[3])])
        corner0 = [None] * 8
        for i in range(4):
            ray = sub(corner[i + 4], corner[i])
            corner0[i + 4] = add(corner[i], mul(ray, split0))
            corner0[i] = add(corner[i], mul(ray, prev_split0))
        for i in range(4):
            ray = sub(corner[i + 4], corner[i])
            near_ray, far_ray = mul(ray, prev_split), mul(ray, split)
            corner[i + 4] = add(corner[i], far_ray)
            corner[i] = add(corner[i], near_ray)
        center, center0 = [f(0), f(0), f(0)], [f(0), f(0), f(0)]
        for i in range(8):
            center, center0 = add(center, corner[i]), add(center0, corner0[i])
        center = [f(c / f(8.0)) for c in center]
        center0 = [f(c / f(8.0)) for c in center0]
        radius, radius0 = f(0), f(0)
        for i in range(8):
            d, d0 = sub(corner[i], center), sub(corner0[i], center0)
            radius = max(radius, f(np.sqrt(dot(d, d))))
            radius0 = max(radius0, f(np.sqrt(dot(d0, d0))))
        cs.append(dict(radius=radius, snapped=f(f(np.ceil(f(radius * f(16.0)))) / f(16.0)), radius0=radius0, min_z=min_z, center=center))
    max_snapped = max(c["snapped"] for c in cs)                                                        # WaveActiveMax, :259
    up = [f(0), f(1), f(0)]
    L = normalize([f(light_dir[0]), f(light_dir[1]), f(light_dir[2])])
    out = []
    for cid in range(count):
        c = cs[cid]
        max_e, min_e = c["snapped"], f(-c["snapped"])
        extent_z = f(max_snapped * f(2.0))
        if cid >= realtime:
            radius_scale, z_bias = f(c["radius0"] / c["radius"]), f(1.0)
        else:
            radius_scale = f(f(f(10.0) * rs_fixed) / c["radius"])
            radius_scale = f(radius_scale / f(radius_scale + f(1.0)))
            z_bias = f(f(0.25) + f(f(c["min_z"] - near) / f(end - start)))
        radius_scale = min(radius_scale, f(1.0))
        cam = sub(c["center"], mul(mul(L, extent_z), f(0.5)))
        fw = normalize(sub(c["center"], cam))
        s = normalize(cross(fw, up))
        u = cross(s, fw)
        V = [[s[0], s[1], s[2], f(-dot(s, cam))], [u[0], u[1], u[2], f(-dot(u, cam))], [f(-fw[0]), f(-fw[1]), f(-fw[2]), dot(fw, cam)], [f(0), f(0), f(0), f(1)]]
        left, right, bottom, top, zn, zf = min_e, max_e, min_e, max_e, extent_z, f(0.0)
        P = [[f(0)] * 4 for _ in range(4)]
        P[0][0] = f(f(2.0) / f(right - left)); P[1][1] = f(f(2.0) / f(top - bottom)); P[2][2] = f(f(-1.0) / f(zf - zn))
        P[0][3] = f(f(-f(right + left)) / f(right - left)); P[1][3] = f(f(-f(top + bottom)) / f(top - bottom)); P[2][3] = f(f(-zn) / f(zf - zn)); P[3][3] = f(1.0)
        size = f(dim)
        origin = mul_point(matmul(P, V), f(0), f(0), f(0), f(1))
        origin = [f(o * f(size / f(2.0))) for o in origin]
        P[0][3] = f(P[0][3] + f(f(f(np.rint(origin[0])) - origin[0]) * f(f(2.0) / size)))
        P[1][3] = f(P[1][3] + f(f(f(np.rint(origin[1])) - origin[1]) * f(f(2.0) / size)))
        VP = matmul(P, V)
        IVP = inverse(VP)
        pts = []
        for x, y, z in ndc:
            h = mul_point(IVP, f(x), f(y), f(z), f(1.0))
            pts.append([f(h[0] / h[3]), f(h[1] / h[3]), f(h[2] / h[3])])
        planes = []
        def plane(a, b, o):
            n = normalize(cross(sub(a, o), sub(b, o)))
            planes.append([n[0], n[1], n[2], f(-dot(n, o))])
            return n
        plane(pts[4], pts[3], pts[7]); plane(pts[6], pts[3], pts[2]); plane(pts[6], pts[1], pts[5]); plane(pts[5], pts[0], pts[4])
        front_n = plane(pts[1], pts[3], pts[0])
        plane(pts[5], pts[7], pts[6])
        planes[5][3] = f(-dot(front_n, pts[6]))                                                        # :365 uses frontN for the back plane's distance
        keep = cache_valid and cid >= realtime and (tick % (count - realtime)) != (cid - realtime)
        out.append(None if keep else dict(translatedWorldToClip=np.array(VP, np.float32), clipToTranslatedWorld=np.array(IVP, np.float32),
                                          planes=np.array(planes, np.float32), ortho=np.array([P[2][2], P[2][3], z_bias, radius_scale], np.float32)))
    return out
