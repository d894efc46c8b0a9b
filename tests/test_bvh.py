"""The hierarchical cull's node test against the flat group test, on the CPU: for every node whose sphere the kernel
would drop, NO group of its subtree passes orc_group_visible -- over many cameras and every procedural tree."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
import orc
from chord_amd import lib as L, records as R
from chord_amd import scenes

MARGIN = np.float32(1.0 - 1.0 / 64.0)


def _projected_error_px(lod_scale, l2v, max_scale, center, radius):
    """kernels_cull.hip projected_error_px (= base.hlsli:233-241,503-518) in numpy float32, source order."""
    f = np.float32
    q = [f(f(f(f(l2v[r][0] * center[0]) + f(l2v[r][1] * center[1])) + f(l2v[r][2] * center[2])) + f(l2v[r][3] * f(1.0))) for r in range(3)]
    R_ = f(max_scale * radius)
    d2 = f(f(f(q[0] * q[0]) + f(q[1] * q[1])) + f(q[2] * q[2]))
    r2 = f(R_ * R_)
    if d2 <= r2:
        return f(-1.0)
    return f(f(lod_scale * R_) / np.sqrt(f(d2 - r2)))


def _subtree_groups(nodes, n):
    out, stack = [], [n]
    while stack:
        k = stack.pop()
        nd = nodes[k]
        out += list(range(nd["leafMeshletGroupOffset"], nd["leafMeshletGroupOffset"] + nd["leafMeshletGroupCount"]))
        stack += [int(c) for c in nd["children"] if c != 0xFFFFFFFF]
    return out


@pytest.mark.parametrize("builder", [lambda: scenes.small_test_scene(320, 200, seed=13), lambda: scenes.config3_street(640, 360),
                                     lambda: scenes.masked_test_scene(320, 200)], ids=["small", "street", "masked"])
def test_a_dropped_node_never_hides_a_visible_group(builder):
    scene, cam0 = builder()
    f = np.array(cam0.front, dtype=np.float64)
    f /= np.linalg.norm(f)
    dropped_nodes = tested = 0
    for back in (0.0, 5.0, 25.0, 90.0, 500.0):
        cam = cam0.moved(tuple(-back * f))
        L.fill_objects(scene, cam)
        view, iv = L.make_views(cam)
        V = view["translatedWorldToView"][0].reshape(4, 4).T.astype(np.float32)      # glm column-major -> M[r][c]
        lod_scale = np.float32(view["lodScale"][0])
        objs = range(len(scene.objects)) if len(scene.objects) <= 40 else range(0, len(scene.objects), 9)
        for o in objs:
            obj = scene.objects[o]
            prim = scene.primitives[obj["GLTFPrimitiveDetail"]]
            M = obj["localToTranslatedWorld"].reshape(4, 4).T.astype(np.float32)
            l2v = np.zeros((4, 4), np.float32)
            for r in range(4):                                                       # mul(A, B) in source order
                for c in range(4):
                    acc = np.float32(np.float32(V[r][0] * M[0][c]) + np.float32(V[r][1] * M[1][c]))
                    acc = np.float32(acc + np.float32(V[r][2] * M[2][c]))
                    l2v[r][c] = np.float32(acc + np.float32(V[r][3] * M[3][c]))
            nodes = scene.bvh_nodes[prim["bvhNodeOffset"]: prim["bvhNodeOffset"] + scene.bvh_nodes[prim["bvhNodeOffset"]]["bvhNodeCount"]]
            groups = scene.groups[prim["meshletGroupOffset"]: prim["meshletGroupOffset"] + prim["meshletGroupCount"]]
            for n in range(len(nodes)):
                if nodes[n]["sphere"][3] <= 0:
                    continue
                pe = _projected_error_px(lod_scale, l2v, np.float32(obj["scaleExtractFromMatrix"][3]), nodes[n]["sphere"][:3], nodes[n]["sphere"][3])
                tested += 1
                if not (pe > 0 and pe <= MARGIN):
                    continue
                dropped_nodes += 1
                below = _subtree_groups(nodes, n) if n else [g for c in nodes[0]["children"] if c != 0xFFFFFFFF for g in _subtree_groups(nodes, int(c))]
                for g in below:
                    vis = orc.lib.orc_group_visible(view.ctypes.data, scene.objects[o:o + 1].ctypes.data, groups[g:g + 1].ctypes.data)
                    assert vis == 0, "object %d node %d (pe %.4f) hides visible group %d at -%g m" % (o, n, pe, g, back)
    assert dropped_nodes > 0 and tested > dropped_nodes


def test_tree_shape_follows_the_reference_builder():
    scene, _ = scenes.config3_street(320, 180)
    for prim in scene.primitives[:40]:
        base = prim["bvhNodeOffset"]
        nodes = scene.bvh_nodes[base: base + scene.bvh_nodes[base]["bvhNodeCount"]]
        groups = scene.groups[prim["meshletGroupOffset"]: prim["meshletGroupOffset"] + prim["meshletGroupCount"]]
        assert nodes[0]["bvhNodeCount"] == len(nodes)                                # nanite_builder.cpp:415
        listed = np.concatenate([np.arange(n["leafMeshletGroupOffset"], n["leafMeshletGroupOffset"] + n["leafMeshletGroupCount"]) for n in nodes])
        assert np.array_equal(listed, np.arange(len(groups)))                        # groups are stored in node order (flattenBVH)
        for k, n in enumerate(nodes):
            kids = [c for c in n["children"] if c != 0xFFFFFFFF]
            assert len(kids) in (0, 8) and all(c > k for c in kids)                  # 2 x 2 x 2 split or a leaf; breadth first
            leaf = groups[n["leafMeshletGroupOffset"]: n["leafMeshletGroupOffset"] + n["leafMeshletGroupCount"]]
            if k:
                assert (leaf["parentError"] < 3e38).all() and len(leaf) < 8
            assert (groups["meshletCount"] <= 4).all()                               # nanite_builder.cpp:411-414
