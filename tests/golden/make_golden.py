#!/usr/bin/env python3
"""Writes tests/golden/config1.npz: BASELINE config 1 (one 128-triangle meshlet, 256x256) inputs and
the oracle's outputs.  The reference holds no golden vectors for this path and cannot
run here, so the vector is produced by the build's own oracle; the analytic known-answer tests in
tests/test_oracle_kat.py pin the oracle itself.  Regenerate only on a deliberate spec change."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import orc  # noqa: E402
from chord_amd import lib as L, records as R, scenes  # noqa: E402


def main():
    scene, cam = scenes.config1_single_meshlet()
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL
    out = orc.frame(scene, view, iv, flags)
    vis = out["vis"]
    np.savez_compressed(
        os.path.join(HERE, "config1.npz"),
        positions=scene.positions, meshlets=scene.meshlets.view(np.uint8), groups=scene.groups.view(np.uint8),
        group_indices=scene.group_indices, meshlet_data=scene.meshlet_data, objects=scene.objects.view(np.uint8),
        primitives=scene.primitives.view(np.uint8), materials=scene.materials.view(np.uint8),
        view=view.view(np.uint8), iv=iv.view(np.uint8), flags=np.uint32(flags),
        vis=vis.reshape(256, 256), cmds=out["cmds"].view(np.uint8), hzb_min=out["hzb_min"], hzb_max=out["hzb_max"],
        valid_range=out["valid_range"], sha256=np.frombuffer(hashlib.sha256(vis.tobytes()).digest(), dtype=np.uint8),
    )
    print("config1: covered", int((vis != 0).sum()), "sha256", hashlib.sha256(vis.tobytes()).hexdigest())


if __name__ == "__main__":
    main()
