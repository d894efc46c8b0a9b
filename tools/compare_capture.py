#!/usr/bin/env python3
"""Compare a capture of the REFERENCE renderer (Vulkan, e.g. a RenderDoc raw save of one frame) with this build's frame
through the only quantity both define identically: per pixel, the decoded tuple (objectId, meshletId, triangleId).

Why not the raw texels: the reference's R32_UINT visibility texel holds ((slot + 1) & 0xFFFFFF) << 8 | triangle
(base.hlsli:437-447) where `slot` is the index a wave obtained from a global InterlockedAdd (instance_culling.hlsl:
191-207) -- it changes from run to run -- and its depth comes from the fixed-function rasterizer (sub-pixel snapping,
interpolation and f32 -> D32 rounding are the GPU's).  slot -> (objectId, meshletId) goes through the post-instanceCulling
command list of the SAME frame (check(drawCmd.z == instanceId), visibility_tile.hlsl:54), so the capture needs three raw
buffers:

    --ref-vis    W*H uint32   the visibility target (VK_FORMAT_R32_UINT), row-major, top row first
    --ref-depth  W*H float32  the depth target (D32 part of D32_SFLOAT_S8_UINT), reverse-Z         [optional]
    --ref-cmds   N*3 uint32   the drawMeshletCmd buffer after instanceCulling: (objectId, meshletId, slot) per command

and from this build (chordvis_readback_visibility / chordvis_readback_cmds, e.g. tools/dump_frame.py):

    --our-vis    W*H uint64   packed words: depth bits << 32 | reference texel layout
    --our-cmds   N*3 uint32

Pixels are classified: identical tuple; "edge" (the two images disagree but one of the two tuples appears in the
8-neighbourhood of the other image: sub-pixel snapping / top-left rule differences move a triangle edge by at most one
pixel); "tie" (different tuples whose depths differ by less than --depth-eps relative: draw order vs 64-bit max, SURVEY
8c-8); everything else is a real difference.  Exit status 0 iff the fraction of real differences is <= --max-real.
"""
import argparse
import sys

import numpy as np


def decode(texels, cmds):
    """R32_UINT texels + command list -> (objectId, meshletId, triangle) int64 arrays, -1 where empty."""
    texels = np.asarray(texels, dtype=np.uint32)
    slot = ((texels >> 8) & 0xFFFFFF).astype(np.int64) - 1
    tri = (texels & 0xFF).astype(np.int64)
    cmds = np.asarray(cmds, dtype=np.uint32).reshape(-1, 3)
    by_slot = np.full((int(cmds[:, 2].max()) + 1 if len(cmds) else 1, 2), -1, dtype=np.int64)
    by_slot[cmds[:, 2]] = cmds[:, :2]
    ok = (texels != 0) & (slot >= 0) & (slot < len(by_slot))
    obj = np.where(ok, by_slot[np.clip(slot, 0, len(by_slot) - 1), 0], -1)
    mesh = np.where(ok, by_slot[np.clip(slot, 0, len(by_slot) - 1), 1], -1)
    return obj, mesh, np.where(ok, tri, -1)


def tuple_key(obj, mesh, tri):
    return (obj << 40) | (mesh << 8) | (tri & 0xFF)


def neighbourhood_has(keys, h, w):
    """For every pixel the set test 'key k occurs within the 3x3 neighbourhood' as a function k -> bool array."""
    k2 = keys.reshape(h, w)
    padded = np.pad(k2, 1, mode="edge")
    stack = np.stack([padded[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], axis=0)
    return lambda other: (stack == other.reshape(1, h, w)).any(axis=0).reshape(-1)


def compare(ref_vis, ref_cmds, our_vis, our_cmds, w, h, ref_depth=None, depth_eps=1e-4):
    our_vis = np.asarray(our_vis, dtype=np.uint64)
    our_tex = (our_vis & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    our_depth = (our_vis >> np.uint64(32)).astype(np.uint32).view(np.float32)
    kr = tuple_key(*decode(ref_vis, ref_cmds))
    ko = tuple_key(*decode(our_tex, our_cmds))
    same = kr == ko
    edge = ~same & (neighbourhood_has(ko, h, w)(kr) | neighbourhood_has(kr, h, w)(ko))
    tie = np.zeros_like(same)
    if ref_depth is not None:
        rd = np.asarray(ref_depth, dtype=np.float32)
        with np.errstate(all="ignore"):
            rel = np.abs(rd - our_depth) / np.maximum(np.maximum(np.abs(rd), np.abs(our_depth)), 1e-30)
        tie = ~same & ~edge & (rel < depth_eps)
    real = ~same & ~edge & ~tie
    n = len(same)
    return {"pixels": n, "identical": int(same.sum()), "edge": int(edge.sum()), "tie": int(tie.sum()), "real": int(real.sum()),
            "real_fraction": float(real.sum()) / n, "covered_ref": int((kr >= 0).sum()), "covered_ours": int((ko >= 0).sum()),
            "first_real": [int(i) for i in np.nonzero(real)[0][:8]]}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--width", type=int, required=True)
    ap.add_argument("--height", type=int, required=True)
    ap.add_argument("--ref-vis", required=True)
    ap.add_argument("--ref-cmds", required=True)
    ap.add_argument("--ref-depth")
    ap.add_argument("--our-vis", required=True)
    ap.add_argument("--our-cmds", required=True)
    ap.add_argument("--depth-eps", type=float, default=1e-4)
    ap.add_argument("--max-real", type=float, default=1e-3, help="largest tolerated fraction of real differences")
    a = ap.parse_args()
    n = a.width * a.height
    ref_vis = np.fromfile(a.ref_vis, dtype=np.uint32, count=n)
    our_vis = np.fromfile(a.our_vis, dtype=np.uint64, count=n)
    ref_depth = np.fromfile(a.ref_depth, dtype=np.float32, count=n) if a.ref_depth else None
    res = compare(ref_vis, np.fromfile(a.ref_cmds, dtype=np.uint32), our_vis, np.fromfile(a.our_cmds, dtype=np.uint32),
                  a.width, a.height, ref_depth, a.depth_eps)
    for k, v in res.items():
        print("%-14s %s" % (k, v))
    return 0 if res["real_fraction"] <= a.max_real else 1


if __name__ == "__main__":
    sys.exit(main())
