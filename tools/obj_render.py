#!/usr/bin/env python3
"""A user-supplied triangle mesh through the whole path: OBJ -> chordvis_nanite_build (meshlets, LOD DAG, BVH) -> optional asset
container -> scene of `--instances` copies receding from the camera -> frames on the GPU (both cull modes), optionally checked
against the oracle.   python tools/obj_render.py mesh.obj [--instances 5] [--size 1920 1080] [--save out.chordasset] [--check]
(The reference imports glTF and builds with meshoptimizer + METIS; this is the counterpart for meshes on disk: SURVEY 7 step 2.)"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from chord_amd import lib as L, obj, records as R, scenes
from chord_amd.scenes import Camera, translate, rotate_y, scale

ap = argparse.ArgumentParser()
ap.add_argument("path"); ap.add_argument("--instances", type=int, default=5)
ap.add_argument("--size", type=int, nargs=2, default=(1920, 1080)); ap.add_argument("--save"); ap.add_argument("--check", action="store_true")
a = ap.parse_args()
pos, idx, uv = obj.read_obj(a.path)
lo, hi = pos.min(axis=0), pos.max(axis=0)
centre, radius = 0.5 * (lo + hi), 0.5 * float(np.linalg.norm(hi - lo))
print("%s: %d vertices, %d triangles, bounds %s .. %s" % (a.path, len(pos), len(idx) // 3, lo, hi))
t0 = time.time()
# normalised to radius 1 around the origin so that the placements below frame any mesh
l2w = [translate((k % 3 - 1) * 0.8 * (2.5 * 1.8 ** k), 0.0, -2.5 * 1.8 ** k) @ rotate_y(0.5 * k) @ scale(1.0 / radius) @ translate(*(-centre)) for k in range(a.instances)]
scene = scenes.scene_from_meshes([(pos, idx, uv)], l2w, name=os.path.basename(a.path))
b = scene.built[0]
print("built in %.2f s: %d meshlets (%d at LOD 0), %d groups, %d LOD levels, %d BVH nodes" % (time.time() - t0, len(b.meshlets), int((b.meshlets["lod"] == 0).sum()), len(b.groups), b.lod_count, len(b.bvh_nodes)))
if a.save:
    h = L.nanite_build(pos, idx, uv, keep_handle=True)
    assert L.lib.chordvis_save_asset(h, a.save.encode()) == 0
    L.lib.chordvis_free_built_asset(h)
    print("saved", a.save)
cam = Camera((0.0, 0.3, 1.5), (0.0, -0.05, -1.0), a.size[0], a.size[1])
L.fill_objects(scene, cam)
view, iv = L.make_views(cam)
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
from chord_amd.renderer import VisibilityRenderer
for hier in (0, 1):
    r = VisibilityRenderer(0); r.set_cull_mode(hier); r.upload_scene(scene); r.allocate_gbuffer(*a.size); r.set_view(view, iv, flags)
    for _ in range(2):
        r.render_frame()
    st, vis = r.stats(), r.read_visibility()
    print("cull mode %d: %d clusters after instance culling, %d + %d rastered, %d triangles submitted, %d of %d pixels covered"
          % (hier, st["countInstanceCulled"], st["countStage0Visible"], st["countStage1Visible"], st["trianglesSubmitted"], int(np.count_nonzero(vis)), vis.size))
    if a.check:
        import orc
        w0 = orc.frame(scene, view, iv, flags)
        w1 = orc.frame(scene, view, iv, flags, prev_hzb_min=w0["hzb_min"])
        print("   vs oracle: %s" % ("bit-exact" if np.array_equal(vis, w1["vis"]) else "%d pixels differ" % int((vis != w1["vis"]).sum())))
    r.close()
