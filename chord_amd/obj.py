"""Wavefront OBJ reader for user-supplied meshes (the reference imports glTF, source/asset/gltf/asset_gltf_helper.cpp:423-611;
asset import is out of this path's scope -- this is the minimum that lets a real mesh reach chordvis_nanite_build).

read_obj(path) -> (positions float32[V,3], indices uint32[3T], texcoord0 float32[V,2] or None)
  * v / vt / f records; polygons are fan-triangulated; negative (relative) indices are resolved
  * a position that appears with different texture coordinates is split (one output vertex per distinct (v, vt) pair)
  * normals, materials, groups, smoothing, lines and points are ignored
"""
import numpy as np


def read_obj(path):
    v, vt = [], []
    corner = {}                       # (v index, vt index) -> output vertex
    out_v, out_vt, idx = [], [], []
    with open(path, "r", errors="replace") as f:
        for line in f:
            if not line or line[0] == "#":
                continue
            t = line.split()
            if not t:
                continue
            if t[0] == "v" and len(t) >= 4:
                v.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "vt" and len(t) >= 3:
                vt.append((float(t[1]), float(t[2])))
            elif t[0] == "f" and len(t) >= 4:
                poly = []
                for c in t[1:]:
                    p = c.split("/")
                    vi = int(p[0])
                    vi = vi - 1 if vi > 0 else len(v) + vi
                    ti = -1
                    if len(p) > 1 and p[1]:
                        ti = int(p[1])
                        ti = ti - 1 if ti > 0 else len(vt) + ti
                    if not 0 <= vi < len(v) or ti >= len(vt):
                        raise ValueError("%s: face refers to a vertex that is not defined yet: %r" % (path, c))
                    k = (vi, ti)
                    if k not in corner:
                        corner[k] = len(out_v)
                        out_v.append(v[vi])
                        out_vt.append(vt[ti] if ti >= 0 else (0.0, 0.0))
                    poly.append(corner[k])
                for i in range(1, len(poly) - 1):
                    idx.extend((poly[0], poly[i], poly[i + 1]))
    if not idx:
        raise ValueError("%s: no faces" % path)
    pos = np.asarray(out_v, dtype=np.float32).reshape(-1, 3)
    uv = np.asarray(out_vt, dtype=np.float32).reshape(-1, 2) if vt else None
    return pos, np.asarray(idx, dtype=np.uint32), uv


def write_obj(path, positions, indices, texcoord0=None):
    """The inverse, for tests and for exporting the procedural meshes."""
    pos = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
    idx = np.asarray(indices, dtype=np.int64).reshape(-1, 3) + 1
    with open(path, "w") as f:
        for p in pos:
            f.write("v %.9g %.9g %.9g\n" % tuple(p))
        if texcoord0 is not None:
            for t in np.asarray(texcoord0, dtype=np.float64).reshape(-1, 2):
                f.write("vt %.9g %.9g\n" % tuple(t))
            for a, b, c in idx:
                f.write("f %d/%d %d/%d %d/%d\n" % (a, a, b, b, c, c))
        else:
            for a, b, c in idx:
                f.write("f %d %d %d\n" % (a, b, c))
