"""Host side of chordvis_group_render_frame: wall time of the CALL (all ranks' work enqueued, nothing waited for) with N ranks
on one device, unpipelined and pipelined -- what the rank threads spend on launches plus on each other (round 2: two
mutex + condition-variable barriers per exchange; round 3: lock-free generation counters, group_all_gather).
  python tools/group_host_time.py [ranks] [workload]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from chord_amd import lib as L, records as R
from chord_amd.renderer import VisibilityGroup
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = sys.argv[2] if len(sys.argv) > 2 else "street_720p_hzb"
scene, cam = bench.build_workload(wl)
view, iv = L.make_views(cam)
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | R.FLAG_HZB_CULL
objs = L.fill_objects(scene, cam, cam)
g = VisibilityGroup([0] * ranks)
g.upload_scene(scene)
g.allocate_gbuffer(cam.width, cam.height)
g.update_objects(objs)
g.set_view(view, iv, flags)
for pipelined in (False, True):
    g.set_pipelined(pipelined)
    for _ in range(5):
        g.render_frame()
    g.sync()
    g.enqueue_ms()               # (reset)
    n, calls = 50, []
    t0 = time.perf_counter()
    for _ in range(n):
        c0 = time.perf_counter()
        g.render_frame()
        calls.append(time.perf_counter() - c0)
    g.sync()
    total = (time.perf_counter() - t0) / n * 1e3
    calls.sort()
    enq = g.enqueue_ms()         # per worker thread: its time inside the frame job (launches, copies, waits for the peers' hand-shakes)
    print("%d ranks on one device, %s, %s: render_frame call %.1f us median (%.1f min, %.1f p90); %.3f ms per frame end to end (%d ranks' kernels share the one GPU); per-worker time inside the call, us: %s"
          % (ranks, wl, "pipelined" if pipelined else "unpipelined", calls[n // 2] * 1e6, calls[0] * 1e6, calls[int(n * 0.9)] * 1e6, total, ranks,
             " ".join("%.0f" % (v * 1e3) for v in enq)))
g.close()
