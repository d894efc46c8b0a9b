import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from chord_amd import lib as L, records as R
from chord_amd.renderer import VisibilityRenderer
scene, cam = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "street_4k_hzb")
f = np.array(cam.front); f = f / np.linalg.norm(f); cam_b = cam.moved(tuple(0.5 * f))
va0, _ = L.make_views(cam); vb0, _ = L.make_views(cam_b)
views = [L.make_views(cam, vb0), L.make_views(cam_b, va0)]
objs = [L.fill_objects(scene, cam, cam_b).copy(), L.fill_objects(scene, cam_b, cam).copy()]
dev = torch.device("cuda", 0); stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
d_obj = [torch.from_numpy(o.view(np.uint8).reshape(-1)).to(dev) for o in objs]
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
r = VisibilityRenderer(0, stream.cuda_stream); r.upload_scene(scene); r.allocate_gbuffer(cam.width, cam.height); r.enable_timers(0)
def frame(i):
    r.bind_objects(d_obj[i & 1].data_ptr(), len(scene.objects)); r.set_view(views[i & 1][0], views[i & 1][1], flags); r.render_frame()
for i in range(400): frame(i)
torch.cuda.synchronize()
for n in (20, 200, 1000):
    torch.cuda.synchronize(); a = time.perf_counter()
    for i in range(n): frame(i)
    b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    print("n %4d: host enqueue %.1f us per frame, total %.1f us per frame" % (n, (b - a) / n * 1e6, (c - a) / n * 1e6))
# the pieces of the host's frame
n = 2000
a = time.perf_counter()
for i in range(n): r.bind_objects(d_obj[i & 1].data_ptr(), len(scene.objects))
b = time.perf_counter()
for i in range(n): r.set_view(views[i & 1][0], views[i & 1][1], flags)
c = time.perf_counter()
print("bind_objects %.2f us, set_view %.2f us per call" % ((b - a) / n * 1e6, (c - b) / n * 1e6))
torch.cuda.synchronize()
a = time.perf_counter()
for i in range(200): r.render_frame()
b = time.perf_counter(); torch.cuda.synchronize()
print("render_frame alone: host %.1f us per call" % ((b - a) / 200 * 1e6))
