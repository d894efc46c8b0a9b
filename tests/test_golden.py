"""Committed golden vector (BASELINE config 1) against the oracle; inputs are checked too so a drift of
the procedural generator or of the host camera math is caught, not silently re-baselined."""
import hashlib
import json
import os

import numpy as np
import pytest

import orc
from chord_amd import records as R
from chord_amd import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "config1.npz")


def load_gold():
    g = np.load(GOLD)
    scene = R.Scene(g["objects"].view(R.OBJECT), g["primitives"].view(R.PRIMITIVE), g["materials"].view(R.MATERIAL),
                    g["meshlets"].view(R.MESHLET), g["groups"].view(R.MESHLET_GROUP), g["group_indices"],
                    g["meshlet_data"], g["positions"], name="config1_golden")
    # (the fixture predates ChordCameraView::clipToTranslatedWorldWithZFar_NoJitter, a trailing field only the cascade
    # setup reads: the stored record is the prefix of today's)
    raw = np.zeros(R.CAMERA_VIEW.itemsize, dtype=np.uint8)
    stored = g["view"].view(np.uint8).reshape(-1)
    raw[:len(stored)] = stored
    view = raw.view(R.CAMERA_VIEW).copy()
    iv = g["iv"].view(R.INSTANCE_CULLING_VIEW).copy()
    return g, scene, view, iv, int(g["flags"])


def test_oracle_reproduces_golden_config1():
    g, scene, view, iv, flags = load_gold()
    out = orc.frame(scene, view, iv, flags)
    assert hashlib.sha256(out["vis"].tobytes()).digest() == g["sha256"].tobytes()
    assert np.array_equal(out["vis"].reshape(256, 256), g["vis"])
    assert np.array_equal(out["cmds"].view(np.uint8), g["cmds"])
    assert np.array_equal(out["hzb_min"], g["hzb_min"]) and np.array_equal(out["hzb_max"], g["hzb_max"])
    assert np.array_equal(out["valid_range"], g["valid_range"])
    # config 1 facts: one cluster, 128 triangles, all front facing and large
    st = out["stats"]
    assert (st.clusters, st.trianglesSubmitted, st.trianglesRastered) == (1, 128, 128)
    assert st.fragments == int((g["vis"] != 0).sum()) == 42382


def test_generator_and_host_camera_reproduce_golden_inputs():
    from chord_amd import lib as L
    g, gscene, gview, giv, _ = load_gold()
    scene, cam = scenes.config1_single_meshlet()
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    assert np.array_equal(scene.positions, gscene.positions)
    assert np.array_equal(scene.meshlets.view(np.uint8), gscene.meshlets.view(np.uint8))
    assert np.array_equal(scene.meshlet_data, gscene.meshlet_data)
    assert np.array_equal(scene.groups.view(np.uint8), gscene.groups.view(np.uint8))
    assert np.array_equal(scene.objects.view(np.uint8), gscene.objects.view(np.uint8))
    stored = len(g["view"].view(np.uint8).reshape(-1))                  # the fixture holds the record as it was when committed (a prefix)
    assert np.array_equal(view.view(np.uint8).reshape(-1)[:stored], gview.view(np.uint8).reshape(-1)[:stored])
    assert np.array_equal(iv.view(np.uint8), giv.view(np.uint8))


def test_host_camera_matches_glm_fixture():
    """chordvis_camera_fill_view / chordvis_object_basic_data against numbers produced with the
    reference's vendored glm (tests/golden/make_glm_fixture.cpp; generated in the build container)."""
    import ctypes as C
    from chord_amd import lib as L
    with open(os.path.join(HERE, "golden", "glm_camera.json")) as fh:
        fx = json.load(fh)
    for c in fx:
        cam = scenes.Camera(c["position"], c["front"], c["width"], c["height"], fovy=c["fovy"], z_near=c["zNear"],
                            z_far=c["zFar"], jitter=tuple(c["jitter"]))
        view, iv = L.make_views(cam)
        np.testing.assert_allclose(view["translatedWorldToView"][0], np.float32(c["translatedWorldToView"]), rtol=0, atol=2e-7)
        np.testing.assert_allclose(view["translatedWorldToClip"][0], np.float32(c["translatedWorldToClip"]), rtol=3e-7, atol=3e-7)
        np.testing.assert_allclose(iv["translatedWorldToClip"][0], view["translatedWorldToClip"][0], rtol=0, atol=0)
        np.testing.assert_allclose(iv["clipToTranslatedWorld"][0], np.float32(c["clipToTranslatedWorld"]), rtol=2e-5, atol=1e-6)
        obj = np.zeros(1, dtype=R.OBJECT)
        l2w = np.array(c["localToWorld"], dtype=np.float64)
        pos = (C.c_double * 3)(*c["position"])
        assert L.lib.chordvis_object_basic_data(l2w.ctypes.data, None, pos, None, obj.ctypes.data) == 0
        np.testing.assert_allclose(obj["localToTranslatedWorld"][0], np.float32(c["localToTranslatedWorld"]), rtol=0, atol=0)
        np.testing.assert_allclose(obj["translatedWorldToLocal"][0], np.float32(c["translatedWorldToLocal"]), rtol=2e-6, atol=1e-6)
        # max |scale| (scene_node.cpp:60-69)
        cols = obj["localToTranslatedWorld"][0].reshape(4, 4)[:3, :3]
        assert abs(obj["scaleExtractFromMatrix"][0][3] - np.linalg.norm(cols, axis=1).max()) < 1e-6
        # frustum planes (camera.cpp:80-154): inward unit normals; the view axis is inside all of them
        planes = iv["frustumPlanesRS"][0]
        f = np.array(c["front"]) / np.linalg.norm(c["front"])
        assert np.allclose(np.linalg.norm(planes[:, :3], axis=1), 1.0, atol=1e-5)
        for d in (1.0, 100.0):
            assert (planes[:, :3] @ (f * d) + planes[:, 3] > 0).all()
        assert planes[4, :3] @ f > 0.999 and planes[5, :3] @ f < -0.999       # front / back planes
        assert abs(view["lodScale"][0] - c["height"] * 0.5 / np.tan(0.5 * c["fovy"])) < 1e-2


def test_oracle_reproduces_the_committed_digests():
    """tests/golden/oracle_digests.json (make_digests.py): the oracle and the scene generators have not drifted on
    reduced instances of configs 2, 3, 5 and the small test scene -- the anchor of every GPU parity test."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_digests", os.path.join(os.path.dirname(__file__), "golden", "make_digests.py"))
    md = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(md)
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_digests.json")))
    assert sorted(want) == sorted(md.CASES)
    for name in md.CASES:
        assert md.digests(name) == want[name], name
