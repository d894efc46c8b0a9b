"""tools/compare_capture.py on synthetic captures: an oracle frame re-encoded the way the reference would produce it
(cluster slots in a random order, as its InterlockedAdd compaction yields them) must compare as identical; an image
shifted by one pixel as edge differences; a swapped object as real ones."""
import os
import subprocess
import sys

import numpy as np

import helpers as H
import orc
from chord_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import compare_capture as CC  # noqa: E402


def _frame():
    scene, cam, view, iv = H.setup_scene(lambda: scenes.small_test_scene(320, 200, seed=21))
    fr = orc.frame(scene, view, iv, H.ALL_FLAGS)
    return fr, cam.width, cam.height


def _as_reference(fr, rng):
    """Re-encode with a random slot permutation + the command list in that order."""
    cmds = np.stack([fr["cmds"]["objectId"], fr["cmds"]["meshletId"], fr["cmds"]["slot"]], axis=1).astype(np.uint32)
    perm = rng.permutation(len(cmds)).astype(np.uint32)               # old slot -> new slot
    tex = (fr["vis"] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    slot = ((tex >> 8) & 0xFFFFFF).astype(np.int64) - 1
    new = np.where(tex != 0, ((perm[np.clip(slot, 0, len(perm) - 1)].astype(np.uint32) + 1) << 8) | (tex & 0xFF), 0).astype(np.uint32)
    ref_cmds = cmds.copy()
    ref_cmds[:, 2] = perm[cmds[:, 2]]
    ref_cmds = ref_cmds[rng.permutation(len(ref_cmds))]               # list order is arbitrary too
    depth = (fr["vis"] >> np.uint64(32)).astype(np.uint32).view(np.float32)
    return new, ref_cmds, depth, cmds


def test_permuted_slots_compare_identical_and_differences_are_classified(tmp_path):
    fr, w, h = _frame()
    rng = np.random.default_rng(5)
    ref_vis, ref_cmds, ref_depth, our_cmds = _as_reference(fr, rng)
    res = CC.compare(ref_vis, ref_cmds, fr["vis"], our_cmds, w, h, ref_depth)
    assert res["identical"] == w * h and res["real"] == 0 and res["covered_ref"] == res["covered_ours"] > 0
    # one pixel to the right: differences only along triangle edges
    shifted = np.roll(ref_vis.reshape(h, w), 1, axis=1).reshape(-1)
    res = CC.compare(shifted, ref_cmds, fr["vis"], our_cmds, w, h)
    assert res["edge"] > 0 and res["real"] < 0.02 * w * h and res["identical"] > 0.5 * w * h
    # another object's ids in a region: real differences
    wrong = ref_vis.copy().reshape(h, w)
    wrong[40:80, 60:140] = ref_vis.max()
    wrong_depth = ref_depth.copy().reshape(h, w)
    wrong_depth[40:80, 60:140] *= 0.5
    res = CC.compare(wrong.reshape(-1), ref_cmds, fr["vis"], our_cmds, w, h, wrong_depth.reshape(-1))
    assert res["real"] > 1000
    # the same ids at (nearly) the same depth: a depth tie, resolved by draw order there and by the 64-bit max here
    res = CC.compare(wrong.reshape(-1), ref_cmds, fr["vis"], our_cmds, w, h, ref_depth)
    assert res["real"] == 0 and res["tie"] > 1000
    # the command-line form, on raw files
    files = {}
    for name, arr in (("ref_vis", ref_vis), ("ref_cmds", ref_cmds), ("ref_depth", ref_depth), ("our_vis", fr["vis"]), ("our_cmds", our_cmds)):
        files[name] = str(tmp_path / (name + ".bin"))
        np.ascontiguousarray(arr).tofile(files[name])
    cmd = [sys.executable, os.path.join(ROOT, "tools", "compare_capture.py"), "--width", str(w), "--height", str(h),
           "--ref-vis", files["ref_vis"], "--ref-cmds", files["ref_cmds"], "--ref-depth", files["ref_depth"],
           "--our-vis", files["our_vis"], "--our-cmds", files["our_cmds"]]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0 and "real           0" in out.stdout, out.stdout + out.stderr
