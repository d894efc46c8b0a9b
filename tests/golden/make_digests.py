#!/usr/bin/env python3
"""Writes tests/golden/oracle_digests.json: SHA-256 of the oracle's outputs (visibility words, HZB min / max, command
list) on reduced-size instances of BASELINE configs 2-5 and on the small test scene, two frames each where the
config is two-pass.  The digests pin the ORACLE (and the procedural scene generators) against accidental change: it is
the anchor every GPU parity test compares with, and nothing else would notice if it drifted.  Regenerate only on a
deliberate spec change (and say so in the commit)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import orc  # noqa: E402
from chord_amd import lib as L, records as R, scenes  # noqa: E402

ALL = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
CASES = {
    "small_320x200": (lambda: scenes.small_test_scene(320, 200, seed=7), ALL),
    "config2_atrium_480x270": (lambda: scenes.config2_atrium(480, 270), R.FLAG_FRUSTUM_CULL),
    "config3_street_640x360": (lambda: scenes.config3_street(640, 360), ALL),
    "config5_subpixel_480x270": (lambda: scenes.config5_subpixel(480, 270, prims=4, patches_per_prim=256, instances=2),
                                 R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL),
    # alpha-tested materials (oracle item 9: the pinned texture fetch) and, below, depth-only views of the same scene (item 10)
    "masked_320x200": (lambda: scenes.masked_test_scene(320, 200), ALL),
}
DEPTH_CASES = {"masked_320x200": dict(cascadeCount=3, realtimeCascadeCount=2, cascadeDim=256, cascadeEndDistance=14.0, farCascadeEndDistance=40.0,
                                      shadowBiasConst=-8.0, shadowBiasSlope=-0.5)}
LIGHT = (0.35, -1.0, 0.25)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def digests(name):
    builder, flags = CASES[name]
    scene, cam = builder()
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    f0 = orc.frame(scene, view, iv, flags)
    out = {"frame0": {"vis": sha(f0["vis"]), "hzb_min": sha(f0["hzb_min"]), "hzb_max": sha(f0["hzb_max"]),
                      "cmds": sha(f0["cmds"]), "counts": [int(x) for x in f0["counts"]],
                      "covered": int((f0["vis"] != 0).sum())}}
    if flags & R.FLAG_HZB_CULL:
        f1 = orc.frame(scene, view, iv, flags, prev_hzb_min=f0["hzb_min"])
        out["frame1"] = {"vis": sha(f1["vis"]), "hzb_min": sha(f1["hzb_min"]), "hzb_max": sha(f1["hzb_max"]),
                         "cmds": sha(f1["cmds"]), "counts": [int(x) for x in f1["counts"]],
                         "covered": int((f1["vis"] != 0).sum())}
    if name in DEPTH_CASES:
        # shadow cascades: per-view instance cull, depth raster with clamp + bias, generic HZB cull against the farther cascade
        cfg = R.default_cascade_config(**DEPTH_CASES[name])
        dim = int(cfg["cascadeDim"][0])
        views = L.cascade_setup(cfg, view, iv, LIGHT)
        campos = np.frombuffer(iv["cameraWorldPos"][0].tobytes(), dtype=np.float64)[:3]
        desc = orc.hzb_desc(dim, dim)
        prev = None
        for k in range(len(views) - 1, -1, -1):
            cmds = orc.instance_culling(scene, view, views[k:k + 1], flags)
            kept = cmds if prev is None else orc.hzb_culling_generic(scene, views[prev[0]:prev[0] + 1], campos, flags, 1.5, False, desc, prev[1], cmds)
            depth, _ = orc.raster_depth(scene, views[k:k + 1], kept, dim, dim, bias_const=float(cfg["shadowBiasConst"][0]), bias_slope=float(cfg["shadowBiasSlope"][0]))
            words = depth.view(np.uint32).astype(np.uint64) << np.uint64(32)
            _, hmin, _, _ = orc.hzb_build(words, dim, dim)
            out["cascade%d" % k] = {"view": sha(views[k:k + 1]), "cmds": sha(cmds), "kept": sha(kept), "depth": sha(depth), "hzb_min": sha(hmin),
                                    "covered": int((depth > 0).sum())}
            prev = (k, hmin)
    return out


def main():
    res = {name: digests(name) for name in CASES}
    with open(os.path.join(HERE, "oracle_digests.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for k, v in res.items():
        print(k, v["frame0"]["covered"], v["frame0"]["vis"][:16])


if __name__ == "__main__":
    main()
