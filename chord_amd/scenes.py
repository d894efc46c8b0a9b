"""Deterministic procedural meshlet scenes for BASELINE.json's configs.

The reference ships no Sponza/Bistro assets (install/resource/mesh holds only
low_sphere.glb) and its importer needs METIS (nanite_builder.cpp:692-716), so
the inputs are synthesized here in the *output format* of the reference's
importer: meshlets of <=255 vertices / <=128 triangles with AABB + normal cone
(nanite_builder.cpp:1024-1044, meshopt_clusterizer.cpp:760-860), meshlet groups
that share one (error sphere, parent error sphere) (nanite_builder.cpp:313-395),
and the packed meshletData stream [V vertex ids][T tris i0|i1<<8|i2<<16]
(asset_gltf_helper.cpp:522-548).

Every surface is a grid of 9x9-vertex patches (81 vertices / 128 triangles per
meshlet).  LOD levels follow a patch quadtree: the 4 LOD0 patches of a 2x2
block are replaced by 2 LOD1 meshlets (half the triangles), the 8 LOD1 meshlets
of a 4x4 block by 4 LOD2 meshlets.
"""
import ctypes as C
import math

import numpy as np

from . import records as T

FLT_MAX = np.float32(3.4028234663852886e38)


def pcg_hash(v):
    """Counter-based PCG (Jarzynski & Olano 2020) on uint32 arrays."""
    v = np.asarray(v, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    state = (v * np.uint64(747796405) + np.uint64(2891336453)) & np.uint64(0xFFFFFFFF)
    sh = ((state >> np.uint64(28)) + np.uint64(4))
    word = (((state >> sh) ^ state) * np.uint64(277803737)) & np.uint64(0xFFFFFFFF)
    return (((word >> np.uint64(22)) ^ word) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def pcg32(seed, i):
    """i-th 32-bit value of stream `seed`."""
    return pcg_hash((np.uint64(seed) * np.uint64(0x9E3779B9) + np.asarray(i, dtype=np.uint64)) & np.uint64(0xFFFFFFFF))


def rand01(seed, i):
    return (pcg32(seed, i).astype(np.float64) / 4294967295.0)


# ---------------------------------------------------------------------------------------------
# 9x9 patch topology: vertex (i, j) -> j*9+i ; cell -> (v00, v10, v11), (v00, v11, v01): CCW
# when the parameter plane is seen with u to the right and v up (normal = dS/du x dS/dv).

def _patch_tri_table():
    tris = []
    for j in range(8):
        for i in range(8):
            v00 = j * 9 + i
            v10 = v00 + 1
            v01 = v00 + 9
            v11 = v01 + 1
            tris.append(v00 | (v10 << 8) | (v11 << 16))
            tris.append(v00 | (v11 << 8) | (v01 << 16))
    return np.array(tris, dtype=np.uint32)


PATCH_TRIS = _patch_tri_table()
PATCH_V, PATCH_T = 81, 128
_TI = np.stack([PATCH_TRIS & 0xFF, (PATCH_TRIS >> 8) & 0xFF, (PATCH_TRIS >> 16) & 0xFF], axis=1).astype(np.int64)


def _meshlet_bounds(pos):
    """pos: (M, 81, 3) float32 -> AABB + meshopt-style normal cone per meshlet."""
    pos64 = pos.astype(np.float64)
    pmin = pos.min(axis=1)
    pmax = pos.max(axis=1)
    a, b, c = pos64[:, _TI[:, 0]], pos64[:, _TI[:, 1]], pos64[:, _TI[:, 2]]
    n = np.cross(b - a, c - a)                                # (M, 128, 3)
    ln = np.linalg.norm(n, axis=2, keepdims=True)
    valid = ln[..., 0] > 0
    n = np.where(ln > 0, n / np.maximum(ln, 1e-300), 0.0)
    axis = n.sum(axis=1)
    la = np.linalg.norm(axis, axis=1, keepdims=True)
    axis = np.where(la > 0, axis / np.maximum(la, 1e-300), 0.0)
    dp = (n * axis[:, None, :]).sum(axis=2)
    dp = np.where(valid, dp, 1.0)
    mindp = dp.min(axis=1)
    center = 0.5 * (pmin.astype(np.float64) + pmax.astype(np.float64))
    # apex = center - axis * maxt, t = dot(center - corner, n) / dot(axis, n)
    dc = ((center[:, None, :] - a) * n).sum(axis=2)
    dn = np.where(dp > 1e-6, dp, 1.0)
    t = np.where(valid, dc / dn, 0.0)
    maxt = np.maximum(t.max(axis=1), 0.0)
    apex = center - axis * maxt[:, None]
    cutoff = np.sqrt(np.maximum(0.0, 1.0 - mindp * mindp))
    degenerate = mindp <= 0.1                                  # meshopt: cone wider than ~168 deg => never culls
    axis = np.where(degenerate[:, None], 0.0, axis)
    apex = np.where(degenerate[:, None], 0.0, apex)
    cutoff = np.where(degenerate, 1.0, cutoff)
    return pmin, pmax, axis.astype(np.float32), cutoff.astype(np.float32), apex.astype(np.float32)


class PrimitiveBuilder:
    """Accumulates parametric surfaces into one primitive (one GLTFPrimitiveBuffer)."""

    def __init__(self):
        self.positions = []        # list of (n,3) float32
        self.texcoords = []        # list of (n,2) float32: the surface parameters (u, v) times uv_scale
        self.uv_scale = (1.0, 1.0)
        self.nverts = 0
        self.meshlets = []         # list of structured arrays (dataOffset relative to primitive)
        self.meshlet_data = []     # list of uint32 arrays
        self.ndata = 0
        self.nmeshlets = 0
        self.groups = []
        self.group_indices = []
        self.nindices = 0

    def _add_meshlets(self, pos, lod, uv=None):
        """pos (M,81,3) float32, uv (M,81,2). Returns meshlet ids (relative to this primitive)."""
        M = pos.shape[0]
        self.texcoords.append(np.zeros((M * PATCH_V, 2), np.float32) if uv is None else uv.reshape(-1, 2).astype(np.float32))
        pmin, pmax, axis, cutoff, apex = _meshlet_bounds(pos)
        ml = np.zeros(M, dtype=T.MESHLET)
        ml["posMin"], ml["posMax"] = pmin, pmax
        ml["coneAxis"], ml["coneCutOff"], ml["coneApex"] = axis, cutoff, apex
        ml["lod"] = lod
        ml["vertexTriangleCount"] = PATCH_V | (PATCH_T << 8)
        stride = PATCH_V + PATCH_T
        ml["dataOffset"] = self.ndata + np.arange(M, dtype=np.uint32) * stride
        data = np.empty((M, stride), dtype=np.uint32)
        data[:, :PATCH_V] = self.nverts + np.arange(M, dtype=np.uint32)[:, None] * PATCH_V + np.arange(PATCH_V, dtype=np.uint32)[None, :]
        data[:, PATCH_V:] = PATCH_TRIS[None, :]
        ids = self.nmeshlets + np.arange(M, dtype=np.uint32)
        self.positions.append(pos.reshape(-1, 3))
        self.nverts += M * PATCH_V
        self.meshlets.append(ml)
        self.meshlet_data.append(data.reshape(-1))
        self.ndata += M * stride
        self.nmeshlets += M
        return ids

    def _add_groups(self, member_ids, center, error, parent_center, parent_error):
        """member_ids (G, k) meshlet ids; one group per row."""
        G, k = member_ids.shape
        g = np.zeros(G, dtype=T.MESHLET_GROUP)
        g["clusterPosCenter"] = center
        g["error"] = error
        g["parentPosCenter"] = parent_center
        g["parentError"] = parent_error
        g["meshletOffset"] = self.nindices + np.arange(G, dtype=np.uint32) * k
        g["meshletCount"] = k
        self.groups.append(g)
        self.group_indices.append(member_ids.reshape(-1).astype(np.uint32))
        self.nindices += G * k

    def add_surface(self, S, P, Q, lods=1, error_scale=0.06):
        """S(u, v) -> (..., 3) float64 over [0,1]^2; P x Q LOD0 patches; lods in {1, 2, 3}."""
        if lods > 1:
            assert P % 4 == 0 and Q % 4 == 0, "LOD quadtree needs patch grids in multiples of 4"
        k = np.arange(9) / 8.0

        def sample(u0, v0, du, dv):
            # u0,v0: (M,) patch origins; du,dv patch extents -> (M,81,3) float32 (row j = v, col i = u)
            U = u0[:, None, None] + du * k[None, None, :]
            V = v0[:, None, None] + dv * k[None, :, None]
            U, V = np.broadcast_arrays(U, V)
            self._last_uv = np.stack([U * self.uv_scale[0], V * self.uv_scale[1]], axis=-1).astype(np.float32).reshape(len(u0), 81, 2)
            return S(U, V).astype(np.float32).reshape(len(u0), 81, 3)

        pi, pj = np.meshgrid(np.arange(P), np.arange(Q), indexing="xy")     # (Q,P)
        pos0 = sample((pi / P).reshape(-1), (pj / Q).reshape(-1), 1.0 / P, 1.0 / Q)
        ids0 = self._add_meshlets(pos0, 0, self._last_uv).reshape(Q, P)
        edge = float(np.mean(np.linalg.norm(pos0[:, 8].astype(np.float64) - pos0[:, 0].astype(np.float64), axis=1)))
        e1, e2 = error_scale * edge, 2.0 * error_scale * edge

        def block_centers(step):
            # AABB centre of each step x step block of LOD0 patches -> (Q/step, P/step, 3)
            bq, bp = Q // step, P // step
            pm = pos0.reshape(Q, P, 81, 3)
            pm = pm.reshape(bq, step, bp, step, 81, 3)
            mn = pm.min(axis=(1, 3, 4))
            mx = pm.max(axis=(1, 3, 4))
            return (0.5 * (mn.astype(np.float64) + mx.astype(np.float64))).astype(np.float32)

        if lods == 1:
            # un-parented LOD0 groups: runs of up to 4 meshlets in id order (rootSphereGroupMap, nanite_builder.cpp:373-390)
            flat = ids0.reshape(-1)
            n4 = (len(flat) // 4) * 4
            if n4:
                self._add_groups(flat[:n4].reshape(-1, 4), 0.0, -1.0, 0.0, FLT_MAX)
            if len(flat) - n4:
                self._add_groups(flat[n4:].reshape(1, -1), 0.0, -1.0, 0.0, FLT_MAX)
            return

        c1 = block_centers(2)                                                 # (Q/2, P/2, 3)
        blk0 = ids0.reshape(Q // 2, 2, P // 2, 2).transpose(0, 2, 1, 3).reshape(-1, 4)
        self._add_groups(blk0, 0.0, -1.0, c1.reshape(-1, 3), e1)

        # LOD1: two meshlets per 2x2 block, each spanning half the block in u
        bi, bj = np.meshgrid(np.arange(P // 2), np.arange(Q // 2), indexing="xy")
        u0 = np.stack([bi * 2 / P, bi * 2 / P + 1.0 / P], axis=-1).reshape(-1)
        v0 = np.repeat((bj * 2 / Q).reshape(-1), 2)
        pos1 = sample(u0, v0, 1.0 / P, 2.0 / Q)
        ids1 = self._add_meshlets(pos1, 1, self._last_uv).reshape(Q // 2, P // 2, 2)
        if lods == 2:
            self._add_groups(ids1.reshape(-1, 2), c1.reshape(-1, 3), e1, 0.0, FLT_MAX)
            return
        c2 = block_centers(4)                                                 # (Q/4, P/4, 3)
        c2_for_b1 = np.repeat(np.repeat(c2, 2, axis=0), 2, axis=1)
        self._add_groups(ids1.reshape(-1, 2), c1.reshape(-1, 3), e1, c2_for_b1.reshape(-1, 3), e2)

        # LOD2: four meshlets per 4x4 block, each spanning a 2x2-patch quadrant
        qi, qj = np.meshgrid(np.arange(P // 2), np.arange(Q // 2), indexing="xy")
        pos2 = sample((qi * 2 / P).reshape(-1), (qj * 2 / Q).reshape(-1), 2.0 / P, 2.0 / Q)
        ids2 = self._add_meshlets(pos2, 2, self._last_uv).reshape(Q // 2, P // 2)
        blk2 = ids2.reshape(Q // 4, 2, P // 4, 2).transpose(0, 2, 1, 3).reshape(-1, 4)
        self._add_groups(blk2, c2.reshape(-1, 3), e2, 0.0, FLT_MAX)

    def finish(self):
        pos = np.concatenate(self.positions) if self.positions else np.zeros((0, 3), np.float32)
        groups, nodes = build_bvh(np.concatenate(self.groups))
        self.groups = [groups]
        return dict(
            bvh_nodes=nodes,
            positions=pos,
            texcoords=np.concatenate(self.texcoords) if self.texcoords else np.zeros((0, 2), np.float32),
            meshlets=np.concatenate(self.meshlets),
            meshlet_data=np.concatenate(self.meshlet_data),
            groups=np.concatenate(self.groups),
            group_indices=np.concatenate(self.group_indices),
        )


def build_bvh(groups):
    """The 8-wide tree over the parented cluster groups of one primitive, shaped like the reference's buildBVHTree /
    buildBVH / flattenBVH (nanite_builder.cpp:77-416): un-parented groups are the root's leaves; the others are split
    2 x 2 x 2 by sorting on the longest axis of the union of their parent-error boxes until fewer than 8 remain; nodes are
    flattened breadth first and the groups re-ordered as the nodes list them.  Returns (groups in tree order, nodes)."""
    groups = np.asarray(groups, dtype=T.MESHLET_GROUP)
    parented = np.nonzero(groups["parentError"] < 3.0e38)[0]
    roots = np.nonzero(groups["parentError"] >= 3.0e38)[0]
    pc = groups["parentPosCenter"].astype(np.float32)
    pe = groups["parentError"].astype(np.float32)

    def bounds(ids):
        if len(ids) == 0:
            return np.zeros(3, np.float32), np.zeros(3, np.float32)
        return (pc[ids] - pe[ids, None]).min(axis=0), (pc[ids] + pe[ids, None]).max(axis=0)

    def longest(mn, mx):
        d = mx - mn
        a = 0
        if d[1] >= d[0] and d[1] >= d[2]:
            a = 1
        if d[2] >= d[0] and d[2] >= d[1]:
            a = 2
        return a

    def halves(ids, mn, mx):
        order = ids[np.argsort(pc[ids, longest(mn, mx)], kind="stable")]
        n = len(order)
        return [order[i * n // 2:(i + 1) * n // 2] for i in range(2)]

    class Node:
        pass
    root = Node()
    root.mn, root.mx = bounds(parented)
    root.leaves, root.children, root.depth, root.todo = list(roots), [None] * 8, 0, parented
    queue, order = [root], []
    while queue:
        nd = queue.pop(0)
        order.append(nd)
        ids = nd.todo
        if len(ids) == 0:
            continue
        if len(ids) < 8 or nd.depth == 13:                       # kNaniteBVHLevelNodeCount / kNaniteMaxBVHLevelCount - 1
            nd.leaves = nd.leaves + list(ids)
            continue
        for i, h0 in enumerate(halves(ids, nd.mn, nd.mx)):
            for j, h1 in enumerate(halves(h0, *bounds(h0))):
                for k, h2 in enumerate(halves(h1, *bounds(h1))):
                    ch = Node()
                    ch.mn, ch.mx = bounds(h2)
                    ch.leaves, ch.children, ch.depth, ch.todo = [], [None] * 8, nd.depth + 1, h2
                    nd.children[(i * 2 + j) * 2 + k] = ch
                    queue.append(ch)
    # breadth-first flatten (the queue above already visits in that order)
    for i, nd in enumerate(order):
        nd.index = i
    nodes = np.zeros(len(order), dtype=T.BVH_NODE)
    new_order = []
    for nd in order:
        n = nodes[nd.index]
        n["sphere"][:3] = 0.5 * (nd.mx + nd.mn)
        n["sphere"][3] = 0.5 * np.float32(np.linalg.norm((nd.mx - nd.mn).astype(np.float32)))
        n["children"] = [c.index if c is not None else 0xFFFFFFFF for c in nd.children]
        n["leafMeshletGroupOffset"], n["leafMeshletGroupCount"] = len(new_order), len(nd.leaves)
        new_order += nd.leaves
    for nd in reversed(order):
        nodes[nd.index]["bvhNodeCount"] = 1 + sum(int(nodes[c.index]["bvhNodeCount"]) for c in nd.children if c is not None)
    assert sorted(new_order) == list(range(len(groups))) and nodes[0]["bvhNodeCount"] == len(nodes)
    return groups[np.array(new_order, dtype=np.int64)], nodes


class SceneBuilder:
    def __init__(self, name):
        self.name = name
        self.prims = []
        self.materials = [self._material(0)]
        self.obj_prim = []
        self.obj_mat = []
        self.obj_l2w = []
        self.textures = []
        self.samplers = []

    @staticmethod
    def _material(two_sided, alpha_mode=0, texture=0xFFFFFFFF, sampler=0, cutoff=0.5, alpha_factor=1.0):
        m = np.zeros(1, dtype=T.MATERIAL)
        m["bTwoSided"] = two_sided
        m["baseColorFactor"] = 1.0
        m["baseColorFactor"][0, 3] = alpha_factor
        m["alphaMode"], m["alphaCutOff"] = alpha_mode, cutoff
        m["baseColorId"], m["baseColorSampler"] = texture, sampler
        m["materialType"] = 1
        return m

    def add_material(self, two_sided, alpha_mode=0, texture=0xFFFFFFFF, sampler=0, cutoff=0.5, alpha_factor=1.0):
        self.materials.append(self._material(two_sided, alpha_mode, texture, sampler, cutoff, alpha_factor))
        return len(self.materials) - 1

    def add_texture(self, rgba8):
        self.textures.append(np.ascontiguousarray(rgba8, dtype=np.uint8))
        return len(self.textures) - 1

    def add_sampler(self, min_filter=T.FILTER_LINEAR_MIPMAP_LINEAR, mag_filter=T.FILTER_LINEAR, wrap_s=T.WRAP_REPEAT, wrap_t=T.WRAP_REPEAT):
        self.samplers.append((min_filter, mag_filter, wrap_s, wrap_t))
        return len(self.samplers) - 1

    def add_primitive(self, builder):
        self.prims.append(builder.finish())
        return len(self.prims) - 1

    def add_object(self, prim, l2w=None, material=0):
        self.obj_prim.append(prim)
        self.obj_mat.append(material)
        self.obj_l2w.append(np.eye(4) if l2w is None else np.asarray(l2w, dtype=np.float64))
        return len(self.obj_prim) - 1

    def build(self):
        prims = np.zeros(len(self.prims), dtype=T.PRIMITIVE)
        pos, ml, md, gr, gi, uv, bv = [], [], [], [], [], [], []
        nv = nm = nd = ng = ni = nb = 0
        for i, p in enumerate(self.prims):
            prims[i]["posMin"] = p["positions"].min(axis=0)
            prims[i]["posMax"] = p["positions"].max(axis=0)
            prims[i]["posAverage"] = p["positions"].mean(axis=0)
            prims[i]["vertexOffset"], prims[i]["vertexCount"] = nv, len(p["positions"])
            prims[i]["meshletOffset"] = nm
            prims[i]["meshletGroupOffset"], prims[i]["meshletGroupCount"] = ng, len(p["groups"])
            prims[i]["meshletGroupIndicesOffset"] = ni
            prims[i]["bvhNodeOffset"] = nb
            bv.append(p["bvh_nodes"]); nb += len(p["bvh_nodes"])
            m = p["meshlets"].copy()
            m["dataOffset"] += nd
            uv.append(p["texcoords"]); pos.append(p["positions"]); ml.append(m); md.append(p["meshlet_data"]); gr.append(p["groups"]); gi.append(p["group_indices"])
            nv += len(p["positions"]); nm += len(m); nd += len(p["meshlet_data"]); ng += len(p["groups"]); ni += len(p["group_indices"])
        objects = np.zeros(len(self.obj_prim), dtype=T.OBJECT)
        objects["GLTFPrimitiveDetail"] = np.array(self.obj_prim, dtype=np.uint32)
        objects["GLTFMaterialData"] = np.array(self.obj_mat, dtype=np.uint32)
        scene = T.Scene(objects, prims, np.concatenate(self.materials), np.concatenate(ml), np.concatenate(gr),
                        np.concatenate(gi), np.concatenate(md), np.concatenate(pos), name=self.name,
                        texcoord0=np.concatenate(uv) if (self.textures and uv) else None, textures=self.textures,
                        samplers=np.array(self.samplers, dtype=T.SAMPLER) if self.samplers else None,
                        bvh_nodes=np.concatenate(bv) if bv else None)
        # glm column-major doubles
        scene.local_to_world = np.ascontiguousarray(np.stack([m.T.reshape(16) for m in self.obj_l2w]), dtype=np.float64)
        return scene


def translate(x, y, z):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def rotate_y(a):
    c, s = math.cos(a), math.sin(a)
    m = np.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m


def scale(sx, sy=None, sz=None):
    sy = sx if sy is None else sy
    sz = sx if sz is None else sz
    return np.diag([sx, sy, sz, 1.0])


# --------------------------------------------------------------------------------- surfaces ---

def _bumps(seed, amp, freq):
    """Smooth deterministic displacement field d(a, b) (sum of 4 sinusoids)."""
    ph = rand01(seed, np.arange(12))
    ths = ph[0:4] * 2 * math.pi
    fs = freq * (0.6 + 1.4 * ph[4:8])
    ps = ph[8:12] * 2 * math.pi

    def d(a, b):
        r = 0.0
        for k in range(4):
            r = r + np.sin(fs[k] * (a * math.cos(ths[k]) + b * math.sin(ths[k])) + ps[k])
        return amp * 0.25 * r
    return d


def plane_surface(origin, du, dv, seed=0, amp=0.0, freq=1.0):
    """S(u,v) = origin + u*du + v*dv + n*d(u,v); normal n = normalize(du x dv)."""
    origin, du, dv = (np.asarray(x, dtype=np.float64) for x in (origin, du, dv))
    n = np.cross(du, dv)
    n = n / np.linalg.norm(n)
    lu, lv = np.linalg.norm(du), np.linalg.norm(dv)
    d = _bumps(seed, amp, freq)

    def S(U, V):
        h = d(U * lu, V * lv) if amp else 0.0
        return origin + U[..., None] * du + V[..., None] * dv + (np.zeros_like(U) + h)[..., None] * n
    return S


def cylinder_surface(base, radius, height, seed=0, amp=0.0):
    """Outward-facing cylinder wall around +y; u = angle, v = height."""
    base = np.asarray(base, dtype=np.float64)
    d = _bumps(seed, amp, 3.0)

    def S(U, V):
        ang = -U * 2 * math.pi                    # dS/du x dS/dv points outward
        r = radius + (d(U * 2 * math.pi * radius, V * height) if amp else 0.0)
        return base + np.stack([r * np.cos(ang), V * height, r * np.sin(ang)], axis=-1)
    return S


# ------------------------------------------------------------------------------------ camera ---

class Camera:
    def __init__(self, position, front, width, height, fovy=math.radians(45.0), z_near=0.001, z_far=20000.0,
                 world_up=(0.0, 1.0, 0.0), jitter=(0.0, 0.0)):
        self.position = tuple(float(x) for x in position)
        self.front = tuple(float(x) for x in front)
        self.world_up = world_up
        self.width, self.height = int(width), int(height)
        self.fovy, self.z_near, self.z_far = float(fovy), float(z_near), float(z_far)
        self.jitter = jitter

    def moved(self, delta):
        c = Camera(tuple(p + d for p, d in zip(self.position, delta)), self.front, self.width, self.height,
                   self.fovy, self.z_near, self.z_far, self.world_up, self.jitter)
        return c


# ------------------------------------------------------------------------------------ configs ---

def config1_single_meshlet():
    """BASELINE config 1: one 128-triangle meshlet, 256x256, fixed camera."""
    i = np.arange(81)
    z = ((pcg32(1, i) & 0xFFFF).astype(np.float64) / 65535.0 - 0.5) * 0.2

    def S(U, V):
        # exact lattice: look the height up by vertex index
        ii = np.rint(U * 8).astype(np.int64)
        jj = np.rint(V * 8).astype(np.int64)
        return np.stack([U * 2 - 1, V * 2 - 1, z[jj * 9 + ii]], axis=-1)

    pb = PrimitiveBuilder()
    pb.add_surface(S, 1, 1, lods=1)
    sb = SceneBuilder("config1_single_meshlet")
    sb.add_object(sb.add_primitive(pb))
    cam = Camera((0.0, 0.0, 3.0), (0.0, 0.0, -1.0), 256, 256)
    return sb.build(), cam


def config2_atrium(width=1920, height=1080):
    """BASELINE config 2 (Sponza-class): 2048 patches = 262 144 triangles, 22 objects, single LOD."""
    sb = SceneBuilder("config2_atrium")
    L, Wd, Hh = 32.0, 16.0, 8.0
    seed = 2000

    def obj(S, P, Q):
        nonlocal seed
        pb = PrimitiveBuilder()
        pb.add_surface(S, P, Q, lods=1)
        sb.add_object(sb.add_primitive(pb))
        seed += 1

    # floor (normal +y), ceiling (normal -y)
    obj(plane_surface((-L / 2, 0, Wd / 2), (L, 0, 0), (0, 0, -Wd), seed, 0.03, 2.0), 32, 16)
    obj(plane_surface((-L / 2, Hh, -Wd / 2), (L, 0, 0), (0, 0, Wd), seed, 0.05, 1.0), 32, 8)
    # long walls (normals facing inward)
    obj(plane_surface((-L / 2, 0, -Wd / 2), (L, 0, 0), (0, Hh, 0), seed, 0.04, 1.5), 32, 8)
    obj(plane_surface((L / 2, 0, Wd / 2), (-L, 0, 0), (0, Hh, 0), seed, 0.04, 1.5), 32, 8)
    # end walls
    obj(plane_surface((-L / 2, 0, Wd / 2), (0, 0, -Wd), (0, Hh, 0), seed, 0.04, 1.5), 16, 8)
    obj(plane_surface((L / 2, 0, -Wd / 2), (0, 0, Wd), (0, Hh, 0), seed, 0.04, 1.5), 16, 8)
    # two colonnades of 8 columns
    for side in (-1, 1):
        for k in range(8):
            x = -L / 2 + 2.0 + k * 4.0
            obj(cylinder_surface((x, 0.0, side * 4.0), 0.45, Hh, seed, 0.02), 4, 8)
    cam = Camera((-L / 2 + 1.0, 1.7, 0.3), (1.0, -0.02, -0.01), width, height)
    return sb.build(), cam


def _building(pb, w, d, h, seed, lods):
    # 4 sides + roof, each 8x8 patches, normals outward
    pb.add_surface(plane_surface((-w / 2, 0, d / 2), (w, 0, 0), (0, h, 0), seed + 0, 0.15, 0.8), 8, 8, lods)      # +z face
    pb.add_surface(plane_surface((w / 2, 0, d / 2), (0, 0, -d), (0, h, 0), seed + 1, 0.15, 0.8), 8, 8, lods)     # +x face
    pb.add_surface(plane_surface((w / 2, 0, -d / 2), (-w, 0, 0), (0, h, 0), seed + 2, 0.15, 0.8), 8, 8, lods)    # -z face
    pb.add_surface(plane_surface((-w / 2, 0, -d / 2), (0, 0, d), (0, h, 0), seed + 3, 0.15, 0.8), 8, 8, lods)    # -x face
    pb.add_surface(plane_surface((-w / 2, h, d / 2), (w, 0, 0), (0, 0, -d), seed + 4, 0.10, 0.5), 8, 8, lods)    # roof


def _street_primitives(sb, lods=3, uv_tiles=None):
    """Unique geometry of one 'street' block: ground + 40 buildings + 311 props. Returns [(prim, l2w)].
    uv_tiles: (buildings, props) texture repeats per surface for the masked variant (texture coordinates are generated either way)."""
    out = []
    pb = PrimitiveBuilder()
    pb.add_surface(plane_surface((-64, 0, 64), (128, 0, 0), (0, 0, -128), 3000, 0.08, 0.9), 64, 64, lods)
    out.append((sb.add_primitive(pb), np.eye(4)))
    # 4 rows x 10 buildings along x; rows at z = -34, -14 | +14, +34 (street down the middle)
    b = 0
    for row, z in enumerate((-36.0, -15.0, 15.0, 36.0)):
        for k in range(10):
            r = rand01(3100, np.arange(b * 4, b * 4 + 4))
            w, d, h = 9.0 + 2.0 * r[0], 9.0 + 2.0 * r[1], 10.0 + 14.0 * r[2]
            pb = PrimitiveBuilder()
            if uv_tiles:
                pb.uv_scale = (uv_tiles[0], uv_tiles[0])
            _building(pb, w, d, h, 3200 + b * 8, lods)
            x = -58.0 + k * 12.8 + (r[3] - 0.5)
            out.append((sb.add_primitive(pb), translate(x, 0.0, z) @ rotate_y((r[3] - 0.5) * 0.2)))
            b += 1
    # 311 props (lamp posts / bollards / planters): 4x4-patch cylinders scattered over the street and sidewalks
    for p in range(311):
        r = rand01(3900, np.arange(p * 5, p * 5 + 5))
        rad, hgt = 0.15 + 0.5 * r[0], 0.8 + 3.5 * r[1]
        pb = PrimitiveBuilder()
        if uv_tiles:
            pb.uv_scale = (uv_tiles[1], uv_tiles[1])
        pb.add_surface(cylinder_surface((0, 0, 0), rad, hgt, 4000 + p, 0.02), 4, 4, lods)
        x, z = -60.0 + 120.0 * r[2], -8.0 + 16.0 * r[3]
        out.append((sb.add_primitive(pb), translate(x, 0.0, z)))
    return out


def config3_street(width=3840, height=2160, lods=3, masked=False):
    """BASELINE config 3 (Bistro-class): 21 872 LOD0 patches = 2 799 616 triangles, 352 objects, 3 LOD levels.
    masked: the SAME geometry with alpha-tested materials (mesh_raster.hlsl:34-38,107-112,198-204) on every prop and every other
    building -- two-sided foliage-style cut-outs (a noise and a disc texture, trilinear / nearest samplers): the workload of
    bench.py --workload street_4k_masked, triangle for triangle the opaque scene."""
    sb = SceneBuilder("config3_street" + ("_masked" if masked else ""))
    mats = [0]
    if masked:
        tex = [sb.add_texture(t) for t in _alpha_textures(11)]
        smp = [sb.add_sampler(T.FILTER_LINEAR_MIPMAP_LINEAR, T.FILTER_LINEAR, T.WRAP_REPEAT, T.WRAP_REPEAT),
               sb.add_sampler(T.FILTER_NEAREST, T.FILTER_NEAREST, T.WRAP_REPEAT, T.WRAP_MIRRORED_REPEAT)]
        # masked == "twin": the same two-sided materials WITHOUT the alpha test -- the opaque frame of equal triangle count the
        # masked frame is measured against (two-sided materials skip the cone cull, so the plain scene submits fewer clusters)
        mode = T.ALPHA_MASK if masked != "twin" else 0
        mats = [sb.add_material(1, mode, tex[1], smp[0], 0.35, 1.0),    # noise: ~2/3 of the surface survives
                sb.add_material(1, mode, tex[2], smp[1], 0.30, 1.0),    # discs
                sb.add_material(1, mode, tex[0], smp[0], 0.5, 1.0)]     # checker
    for k, (prim, l2w) in enumerate(_street_primitives(sb, lods, uv_tiles=(6.0, 3.0) if masked else None)):
        # object 0: the ground (opaque); 1..40: buildings (every other one masked); 41..: props (all masked)
        m = 0 if (not masked or k == 0 or (k <= 40 and k % 2 == 0)) else mats[k % len(mats)]
        sb.add_object(prim, l2w, material=m)
    # second-floor view down the street: ~8.9k clusters pass LOD/frustum/cone culling, ~3.8k survive the
    # two-pass HZB test (~0.49 M triangles submitted, ~11.7 M fragments at 4K)
    cam = Camera((-62.0, 12.0, 3.0), (1.0, -0.18, -0.04), width, height)
    return sb.build(), cam


def config4_street_x64(width=3840, height=2160, grid=8, lods=3):
    """BASELINE config 4: config 3 instanced on a grid x grid lattice (shared geometry), ~179 M triangles at 8x8."""
    sb = SceneBuilder("config4_street_x%d" % (grid * grid))
    prims = _street_primitives(sb, lods)
    pitch = 132.0
    for gz in range(grid):
        for gx in range(grid):
            off = translate((gx - (grid - 1) / 2) * pitch, 0.0, (gz - (grid - 1) / 2) * pitch)
            for prim, l2w in prims:
                sb.add_object(prim, off @ l2w)
    half = (grid - 1) / 2 * pitch
    cam = Camera((-half - 60.0, 45.0, -half - 20.0), (1.0, -0.28, 0.75), width, height)
    return sb.build(), cam


def small_test_scene(width=160, height=96, lods=3, seed=7, two_sided_every=3):
    """A small multi-object scene with LODs for parity tests (a few hundred meshlets)."""
    sb = SceneBuilder("small_test_scene")
    two = sb.add_material(1)
    pb = PrimitiveBuilder()
    pb.add_surface(plane_surface((-8, 0, 8), (16, 0, 0), (0, 0, -16), seed, 0.2, 0.7), 8, 8, lods)
    sb.add_object(sb.add_primitive(pb))
    for k in range(6):
        r = rand01(seed + 1, np.arange(k * 4, k * 4 + 4))
        pb = PrimitiveBuilder()
        if k % 2 == 0:
            _building(pb, 1.5 + r[0], 1.5 + r[1], 1.0 + 2.5 * r[2], seed * 100 + k * 8, min(lods, 3))
        else:
            pb.add_surface(cylinder_surface((0, 0, 0), 0.3 + 0.4 * r[0], 1.0 + 2.0 * r[1], seed * 100 + k, 0.03), 4, 4, lods)
        prim = sb.add_primitive(pb)
        m = translate(-5.0 + 10.0 * r[2], 0.0, -5.0 + 10.0 * r[3]) @ rotate_y(r[0] * 3.0) @ scale(1.0 + 0.5 * r[1])
        sb.add_object(prim, m, material=two if (k % two_sided_every == 1) else 0)
        if k == 2:      # an instanced copy
            sb.add_object(prim, translate(3.0, 0.0, 4.0) @ m)
    cam = Camera((-7.0, 1.6, 6.5), (0.8, -0.12, -0.6), width, height)
    return sb.build(), cam


def _alpha_textures(seed):
    """Three procedural RGBA8 textures whose alpha the masked materials test: a checker, an odd-sized noise, a disc."""
    yy, xx = np.mgrid[0:64, 0:64]
    checker = np.where(((xx // 8) + (yy // 8)) % 2 == 0, 255, 0).astype(np.uint8)
    noise = (pcg_hash(np.arange(37 * 21, dtype=np.uint32) + np.uint32(seed * 7919)) & 0xFF).astype(np.uint8).reshape(21, 37)
    yy, xx = np.mgrid[0:16, 0:16]
    disc = np.clip(255.0 - 40.0 * np.hypot(xx - 7.5, yy - 7.5), 0, 255).astype(np.uint8)
    out = []
    for a in (checker, noise, disc):
        img = np.zeros(a.shape + (4,), np.uint8)
        img[..., 0:3] = 200
        img[..., 3] = a
        out.append(img)
    return out


def masked_test_scene(width=320, height=200, lods=2, seed=3, position=(-6.5, 2.2, 6.0), front=(0.75, -0.22, -0.62)):
    """small_test_scene's layout with alpha-tested, blended and white-fallback materials (mesh_raster.hlsl:34-38,107-112,
    198-204; mesh_raster.cpp:224): holes in the masked surfaces show the geometry behind them, blended objects draw
    nothing.  Three textures x three samplers (every wrap mode, nearest and linear, with and without minification)."""
    sb = SceneBuilder("masked_test_scene")
    tex = [sb.add_texture(t) for t in _alpha_textures(seed)]
    smp = [sb.add_sampler(T.FILTER_LINEAR_MIPMAP_LINEAR, T.FILTER_LINEAR, T.WRAP_REPEAT, T.WRAP_REPEAT),
           sb.add_sampler(T.FILTER_NEAREST, T.FILTER_NEAREST, T.WRAP_CLAMP_TO_EDGE, T.WRAP_MIRRORED_REPEAT),
           sb.add_sampler(T.FILTER_LINEAR_MIPMAP_NEAREST, T.FILTER_NEAREST, T.WRAP_MIRRORED_REPEAT, T.WRAP_CLAMP_TO_EDGE)]
    mats = [sb.add_material(0, T.ALPHA_MASK, tex[0], smp[0], 0.5, 1.0),        # one-sided checker
            sb.add_material(1, T.ALPHA_MASK, tex[1], smp[1], 0.4, 0.9),        # two-sided noise, nearest, clamp / mirror
            sb.add_material(1, T.ALPHA_MASK, tex[2], smp[2], 0.35, 1.0),       # two-sided disc, mirrored / clamp
            sb.add_material(0, T.ALPHA_BLEND),                                  # blended: draws nothing
            sb.add_material(0, T.ALPHA_MASK, 0xFFFFFFFF, 99, 0.5, 1.0),        # no texture: white fallback, opaque in effect
            sb.add_material(1, T.ALPHA_MASK, tex[0], smp[0], 0.5, 0.4)]        # alpha factor below the cut-off: nothing survives
    pb = PrimitiveBuilder()
    pb.add_surface(plane_surface((-8, 0, 8), (16, 0, 0), (0, 0, -16), seed, 0.2, 0.7), 8, 8, lods)
    sb.add_object(sb.add_primitive(pb))                                          # opaque ground
    for k in range(8):
        r = rand01(seed + 1, np.arange(k * 4, k * 4 + 4))
        pb = PrimitiveBuilder()
        pb.uv_scale = (1.0 + 3.0 * r[3], 0.5 + 2.5 * r[2]) if k % 3 else (-2.0, 3.0)     # tiling, incl. negative coordinates
        if k % 2 == 0:
            _building(pb, 1.5 + r[0], 1.5 + r[1], 1.0 + 2.5 * r[2], seed * 100 + k * 8, lods)
        else:
            pb.add_surface(cylinder_surface((0, 0, 0), 0.3 + 0.4 * r[0], 1.0 + 2.0 * r[1], seed * 100 + k, 0.03), 4, 4, lods)
        prim = sb.add_primitive(pb)
        m = translate(-5.0 + 10.0 * r[2], 0.0, -5.0 + 10.0 * r[3]) @ rotate_y(r[0] * 3.0) @ scale(1.0 + 0.5 * r[1])
        sb.add_object(prim, m, material=mats[k % len(mats)])
        if k in (1, 4):                                                          # an opaque copy behind a masked one
            sb.add_object(prim, translate(0.6, 0.0, -1.2) @ m, material=0)
    # a masked screen right in front of the camera: magnified texels, clipped by the near plane at the edges
    pb = PrimitiveBuilder()
    pb.uv_scale = (3.0, 2.0)
    f = np.array(front, dtype=np.float64) / np.linalg.norm(front)
    side = np.cross(f, (0.0, 1.0, 0.0)); side /= np.linalg.norm(side)
    org = np.array(position) + 0.9 * f - 0.7 * side - np.array((0.0, 0.45, 0.0))
    pb.add_surface(plane_surface(tuple(org), tuple(1.4 * side), (0.0, 0.9, 0.0), seed + 5, 0.0, 1.0), 4, 4, 1)
    sb.add_object(sb.add_primitive(pb), material=mats[1])
    cam = Camera(position, front, width, height)
    return sb.build(), cam


def floor_under_camera(position=(0.3, 0.25, 0.2), front=(0.1, -0.6, -1.0), width=128, height=96):
    """One coarse 16 m floor patch (2 m cells) with the camera just above it: its triangles straddle
    the w = 0 plane and exercise the homogeneous clipper."""
    pb = PrimitiveBuilder()
    pb.add_surface(plane_surface((-8, 0, 8), (16, 0, 0), (0, 0, -16)), 1, 1)
    sb = SceneBuilder("floor_under_camera")
    sb.add_object(sb.add_primitive(pb))
    return sb.build(), Camera(position, front, width, height)


def masked_floor_under_camera(position=(0.3, 0.25, 0.2), front=(0.1, -0.6, -1.0), width=160, height=120, seed=4):
    """floor_under_camera with an alpha-tested checker on the (two-sided) floor and an opaque floor 1 m below it: the
    triangles that straddle the w = 0 plane go through the clipper WITH their texture coordinates."""
    sb = SceneBuilder("masked_floor_under_camera")
    tex = sb.add_texture(_alpha_textures(seed)[0])
    smp = sb.add_sampler(T.FILTER_LINEAR_MIPMAP_LINEAR, T.FILTER_LINEAR, T.WRAP_REPEAT, T.WRAP_MIRRORED_REPEAT)
    mat = sb.add_material(1, T.ALPHA_MASK, tex, smp, 0.5, 1.0)
    pb = PrimitiveBuilder()
    pb.uv_scale = (6.0, 6.0)
    pb.add_surface(plane_surface((-8, 0, 8), (16, 0, 0), (0, 0, -16)), 1, 1)
    sb.add_object(sb.add_primitive(pb), material=mat)
    pb = PrimitiveBuilder()
    pb.add_surface(plane_surface((-8, -1, 8), (16, 0, 0), (0, 0, -16)), 2, 2)
    sb.add_object(sb.add_primitive(pb))
    return sb.build(), Camera(position, front, width, height)


def bumpy_sphere_mesh(n=96, seed=1):
    """An indexed triangle mesh (not grid patches): a sphere with low-frequency bumps, open at the poles."""
    u, v = np.meshgrid(np.linspace(0, 2 * np.pi, n, endpoint=False), np.linspace(0.05, np.pi - 0.05, n))
    r = 1.0 + 0.05 * np.sin((4 + seed) * u) * np.sin(7 * v)
    pos = np.stack([r * np.sin(v) * np.cos(u), r * np.cos(v), r * np.sin(v) * np.sin(u)], -1).reshape(-1, 3).astype(np.float32)
    j, i = np.meshgrid(np.arange(n - 1), np.arange(n), indexing="ij")
    a, b, c, d = j * n + i, j * n + (i + 1) % n, (j + 1) * n + i, (j + 1) * n + (i + 1) % n
    idx = np.stack([a, c, b, b, c, d], -1).reshape(-1).astype(np.uint32)
    uv = np.stack([u / (2 * np.pi), v / np.pi], -1).reshape(-1, 2).astype(np.float32)
    return pos, idx, uv


def scene_from_meshes(meshes, local_to_world, prim_of_object=None, two_sided_of_object=None, name="mesh_scene"):
    """A scene of triangle meshes that go through chordvis_nanite_build (own clusterizer / partition / simplifier, SURVEY
    8f-4): `meshes` = [(positions, indices, texcoord0 or None), ...], one primitive each; `local_to_world` = 4x4 matrices, one
    object each (object k instantiates primitive prim_of_object[k], default k mod len(meshes))."""
    from . import lib as L
    prims = [L.nanite_build(pos, idx, uv) for pos, idx, uv in meshes]
    pr = np.zeros(len(prims), dtype=T.PRIMITIVE)
    ml, md, gr, gi, ps, bv, uvs = [], [], [], [], [], [], []
    nv = nm = nd = ng = ni = nb = 0
    for k, a in enumerate(prims):
        pr[k] = a.primitive[0]
        pr[k]["vertexOffset"], pr[k]["meshletOffset"], pr[k]["meshletGroupOffset"] = nv, nm, ng
        pr[k]["meshletGroupIndicesOffset"], pr[k]["bvhNodeOffset"] = ni, nb
        m = a.meshlets.copy(); m["dataOffset"] += nd
        ml.append(m); md.append(a.meshlet_data); gr.append(a.groups); gi.append(a.group_indices); ps.append(a.positions); bv.append(a.bvh_nodes)
        uvs.append(a.texcoord0 if a.texcoord0 is not None else np.zeros((len(a.positions), 2), np.float32))
        nv += len(a.positions); nm += len(m); nd += len(a.meshlet_data); ng += len(a.groups); ni += len(a.group_indices); nb += len(a.bvh_nodes)
    mats = np.concatenate([SceneBuilder._material(0), SceneBuilder._material(1)])
    objects = np.zeros(len(local_to_world), dtype=T.OBJECT)
    objects["GLTFPrimitiveDetail"] = (np.arange(len(local_to_world)) % len(prims)) if prim_of_object is None else np.asarray(prim_of_object)
    objects["GLTFMaterialData"] = 0 if two_sided_of_object is None else np.asarray(two_sided_of_object)
    scene = T.Scene(objects, pr, mats, np.concatenate(ml), np.concatenate(gr), np.concatenate(gi), np.concatenate(md), np.concatenate(ps),
                    name=name, texcoord0=np.concatenate(uvs), bvh_nodes=np.concatenate(bv))
    scene.local_to_world = np.ascontiguousarray(np.stack([np.asarray(m).T.reshape(16) for m in local_to_world]), dtype=np.float64)
    scene.built = prims
    return scene


def built_mesh_scene(width=640, height=360, n=96):
    """Meshes that went through chordvis_nanite_build instead of the grid-patch generator: instances of a bumpy sphere from
    3 m to 400 m so that every LOD level of the DAG is in use."""
    dists = [3.0, 6.0, 14.0, 30.0, 70.0, 160.0, 400.0]
    l2w = [translate(((k % 3) - 1) * 0.3 * dist, 0.08 * dist * ((k // 3) - 1), -dist) @ rotate_y(0.7 * k) @ scale(1.0 + 0.15 * k) for k, dist in enumerate(dists)]
    scene = scene_from_meshes([bumpy_sphere_mesh(n, seed) for seed in (1, 2)], l2w, two_sided_of_object=(np.arange(len(l2w)) // 2) % 2, name="built_mesh_scene")
    return scene, Camera((0.0, 0.4, 1.0), (0.0, -0.05, -1.0), width, height)


def big_built_mesh_scene(width=1280, height=720, n=360):
    """One 258 k-triangle mesh through the builder (7 575 LOD-0 meshlets, 10+ LOD levels), seen close, at mid range and far."""
    l2w = [translate(-0.9, 0.0, -2.4) @ rotate_y(0.3), translate(1.5, 0.2, -7.0) @ rotate_y(1.1), translate(0.0, 3.0, -60.0)]
    scene = scene_from_meshes([bumpy_sphere_mesh(n, 3)], l2w, name="big_built_mesh_scene")
    return scene, Camera((0.0, 0.2, 1.0), (0.0, -0.02, -1.0), width, height)


def config5_subpixel(width=3840, height=2160, prims=1024, patches_per_prim=1024, instances=8, patch_px=8.0, seed=5, hotspot_sigma_px=None):
    """BASELINE config 5: `prims * patches_per_prim` unique camera-facing patches, each ~patch_px x patch_px
    pixels (128 triangles of ~0.5 px^2 at patch_px = 8), centres uniform over the screen, view depth uniform in
    [5, 50], instanced `instances` times with slightly shifted transforms; LOD0 only.  The defaults give
    1 048 576 patches = 134 M unique triangles (1.0 GB of positions + 0.84 GB of meshlet data), x 8 = 1.07 G triangles.
    Camera at the origin looking down -z (so world space = view space).
    hotspot_sigma_px: SURVEY 8d's variant "hotspot" -- the centres are Gaussian around the screen centre with that sigma in
    pixels (Box-Muller on the same counter-based random numbers) instead of uniform: every cluster of the frame lands in a
    few dozen screen tiles, which is the atomic-contention case the configuration is named for."""
    assert patches_per_prim % 4 == 0
    cam = Camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), width, height)
    th = math.tan(0.5 * cam.fovy) if hasattr(cam, "fovy") else math.tan(0.5 * math.radians(45.0))
    aspect = width / height
    sb = SceneBuilder("config5_subpixel")
    k = (np.arange(9) / 8.0 - 0.5)
    M = patches_per_prim
    for p in range(prims):
        idx = (np.arange(M, dtype=np.uint64) + np.uint64(p) * np.uint64(M)) * np.uint64(4)
        if hotspot_sigma_px is None:
            sx = rand01(seed, idx + np.uint64(0)) * width
            sy = rand01(seed, idx + np.uint64(1)) * height
        else:
            u1 = np.maximum(rand01(seed, idx + np.uint64(0)), 1.0e-12)
            u2 = rand01(seed, idx + np.uint64(1))
            rad = np.sqrt(-2.0 * np.log(u1)) * float(hotspot_sigma_px)
            sx = np.clip(0.5 * width + rad * np.cos(2.0 * np.pi * u2), 0.0, width - 1.0)
            sy = np.clip(0.5 * height + rad * np.sin(2.0 * np.pi * u2), 0.0, height - 1.0)
        zv = 5.0 + 45.0 * rand01(seed, idx + np.uint64(2))
        size = patch_px * 2.0 * zv * th / height
        cx = (sx / width * 2.0 - 1.0) * zv * th * aspect
        cy = -(sy / height * 2.0 - 1.0) * zv * th
        X = cx[:, None, None] + size[:, None, None] * k[None, None, :]
        Y = cy[:, None, None] + size[:, None, None] * k[None, :, None]
        X, Y = np.broadcast_arrays(X, Y)
        nz = rand01(seed + 1, (np.arange(M * 81, dtype=np.uint64) + np.uint64(p) * np.uint64(M * 81))).reshape(M, 9, 9)
        Z = -zv[:, None, None] + (nz - 0.5) * 0.2 * size[:, None, None]
        pos = np.stack([X, Y, Z], axis=-1).astype(np.float32).reshape(M, 81, 3)
        pb = PrimitiveBuilder()
        ids = pb._add_meshlets(pos, 0)
        pb._add_groups(ids.reshape(-1, 4), 0.0, -1.0, 0.0, FLT_MAX)     # un-parented LOD0 groups of 4 (nanite_builder.cpp:373-390)
        prim = sb.add_primitive(pb)
        for i in range(instances):
            sb.add_object(prim, translate(0.013 * i, 0.007 * i, -0.05 * i))
    return sb.build(), cam
