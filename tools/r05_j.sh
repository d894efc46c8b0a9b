#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0, '.')
import bench, numpy as np
from chord_amd import lib as L, records as R
from chord_amd.renderer import VisibilityRenderer
for wl in ("street_4k_hzb", "street_x64_4k_hzb", "street_4k_masked", "atrium_1080p", "subpixel_64m"):
    scene, cam = bench.build_workload(wl)
    f = np.array(cam.front); f = f / np.linalg.norm(f); cam_b = cam.moved(tuple(0.5 * f))
    va0, _ = L.make_views(cam); vb0, _ = L.make_views(cam_b)
    views = [L.make_views(cam, vb0), L.make_views(cam_b, va0)]
    objs = [L.fill_objects(scene, cam, cam_b).copy(), L.fill_objects(scene, cam_b, cam).copy()]
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | (0 if wl.startswith("subpixel") else R.FLAG_HZB_CULL)
    r = VisibilityRenderer(0); r.upload_scene(scene); r.allocate_gbuffer(cam.width, cam.height)
    for i in range(4):
        r.update_objects(objs[i & 1]); r.set_view(views[i & 1][0], views[i & 1][1], flags); r.render_frame()
        st = r.stats()
    print(wl, "large records per pass", st["largeRecords"], "clip triangles per pass", st["clipTriangles"], "records", st["triangleRecords"], "bin entries", st["binEntries"], "tiles touched", st["tilesTouched"])
    r.close()
PY
