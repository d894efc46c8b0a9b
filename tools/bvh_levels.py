"""What the hierarchical cull (chordvis_set_cull_mode 1: bvh_cull_kernel) has to work with, level by level: a host replay (numpy
float32, the kernel's expressions) of the walk over every frustum-visible object of a workload -- nodes alive, nodes whose
sphere drops the subtree, leaf groups tested -- beside what the flat pass tests.  No GPU needed.

  python tools/bvh_levels.py [workload ...]      (default: street_4k_hzb street_x64_4k_hzb)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bench
from chord_amd import lib as L, records as R
import orc

MARGIN = np.float32(1.0 - 1.0 / 64.0)
f32 = np.float32


def projected_error(lod_scale, l2v, max_scale, center, radius):
    """kernels_cull.hip projected_error_px, vectorised: l2v (n, 3, 4), center (n, 3), radius (n,)"""
    q = (l2v[:, :, 0] * center[:, None, 0] + l2v[:, :, 1] * center[:, None, 1]).astype(f32)
    q = ((q + l2v[:, :, 2] * center[:, None, 2]).astype(f32) + l2v[:, :, 3]).astype(f32)
    Rr = (max_scale * radius).astype(f32)
    d2 = ((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]).astype(f32) + q[:, 2] * q[:, 2]).astype(f32)
    r2 = (Rr * Rr).astype(f32)
    with np.errstate(invalid="ignore", divide="ignore"):
        pe = (lod_scale * Rr) / np.sqrt((d2 - r2).astype(f32))
    return np.where(d2 <= r2, f32(-1.0), pe.astype(f32))


for wl in (sys.argv[1:] or ["street_4k_hzb", "street_x64_4k_hzb"]):
    scene, cam = bench.build_workload(wl)
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL
    vis = orc.object_cull(scene, iv, flags).astype(bool)
    objs = np.nonzero(vis)[0]
    V = view["translatedWorldToView"][0].reshape(4, 4).T.astype(f32)
    M = scene.objects["localToTranslatedWorld"].reshape(-1, 4, 4).transpose(0, 2, 1).astype(f32)[objs]
    l2v = np.einsum("rk,nkc->nrc", V, M).astype(f32)[:, :3, :]
    max_scale = scene.objects["scaleExtractFromMatrix"][objs][:, 3].astype(f32)
    prim = scene.objects["GLTFPrimitiveDetail"][objs]
    base = scene.primitives["bvhNodeOffset"][prim].astype(np.int64)
    groups_per_obj = scene.primitives["meshletGroupCount"][prim].astype(np.int64)
    nodes = scene.bvh_nodes
    lod_scale = f32(view["lodScale"][0])
    print("%s: %d objects, %d pass the object test, %d group instances beneath them (what the flat pass tests, one thread each)"
          % (wl, len(scene.objects), len(objs), int(groups_per_obj.sum())))
    # root: its own leaves are always tested; its sphere decides for all its children at once
    root = nodes[base]
    tested = int(root["leafMeshletGroupCount"].sum())
    pe = projected_error(lod_scale, l2v, max_scale, root["sphere"][:, :3], root["sphere"][:, 3])
    root_drop = (root["sphere"][:, 3] > 0) & (pe > 0) & (pe <= MARGIN)
    print("  level 0: %7d nodes (the roots), %6d drop every child, %8d leaf groups tested" % (len(objs), int(root_drop.sum()), tested))
    oi = np.repeat(np.arange(len(objs))[~root_drop], 8)
    ch = root["children"][~root_drop].reshape(-1)
    keep = ch != 0xFFFFFFFF
    oi, nd = oi[keep], ch[keep].astype(np.int64)
    level, total_nodes, total_tested = 1, len(objs), tested
    while len(nd):
        n = nodes[base[oi] + nd]
        pe = projected_error(lod_scale, l2v[oi], max_scale[oi], n["sphere"][:, :3], n["sphere"][:, 3])
        drop = (pe > 0) & (pe <= MARGIN)
        t = int(n["leafMeshletGroupCount"][~drop].sum())
        print("  level %d: %7d nodes alive, %6d drop their subtree (%.0f %%), %8d leaf groups tested" % (level, len(nd), int(drop.sum()), 100.0 * drop.mean(), t))
        total_nodes += len(nd); total_tested += t
        oi2 = np.repeat(oi[~drop], 8)
        ch = n["children"][~drop].reshape(-1)
        keep = ch != 0xFFFFFFFF
        oi, nd = oi2[keep], ch[keep].astype(np.int64)
        level += 1
    print("  walk: %d dependent levels, %d node tests + %d group tests (%.1f %% of the groups); flat: %d group tests in one level"
          % (level, total_nodes, total_tested, 100.0 * total_tested / max(1, int(groups_per_obj.sum())), int(groups_per_obj.sum())))
