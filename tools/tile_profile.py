"""Per-tile timing of the raster tile kernel (measurement aid).  Needs the profiling build of the library:
  python chord_amd/build.py --tag prof -DRASTER_PROFILE=1;  CHORDVIS_LIB=chord_amd/_build/libchordvis_prof.so python tools/tile_profile.py hzb"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chord_amd import lib as L, records as R, scenes
from chord_amd.renderer import VisibilityRenderer
import bench
scene, cam = bench.build_workload(os.environ.get("WL", "street_4k_hzb"))       # WL=street_x64_4k_hzb: config 4
import numpy as _np
f = _np.array(cam.front); f = f / _np.linalg.norm(f)
cam_b = cam.moved(tuple(0.5 * f))
va0, _ = L.make_views(cam); vb0, _ = L.make_views(cam_b)
views = [L.make_views(cam, vb0), L.make_views(cam_b, va0)]
objs = [L.fill_objects(scene, cam, cam_b).copy(), L.fill_objects(scene, cam_b, cam).copy()]
view, iv = views[0]
flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | (R.FLAG_HZB_CULL if len(sys.argv) > 1 and sys.argv[1] == "hzb" else 0)
r = VisibilityRenderer(0); r.upload_scene(scene); r.allocate_gbuffer(cam.width, cam.height); r.set_view(view, iv, flags)
r.set_debug(16 | (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
for i in range(5):      # alternate the two bench cameras; profile the last frame (view A after view B)
    r.update_objects(objs[i & 1]); r.set_view(views[i & 1][0], views[i & 1][1], flags); r.render_frame()
tx, ty = (cam.width + 63) // 64, (cam.height + 63) // 64
for p in (0, 1):
    ticks = np.zeros(tx * ty * 9, np.uint64); cnt = np.zeros(tx * ty, np.uint32)
    assert L.lib.chordvis_debug_tile_profile(r._ctx, p, ticks.ctypes.data, cnt.ctypes.data, tx * ty * 9) == 0
    raw = ticks[tx * ty:].reshape(tx * ty, 8)
    # phase words: low half = the phase's ticks (10 ns), high half = the eight waves' ticks spent AT the phase's closing barrier (summed)
    barrier = (raw[:, :6] >> np.uint64(32)).astype(np.float64) / 100.0 / 8.0          # per wave, us
    raw = raw.copy(); raw[:, :6] &= np.uint64(0xFFFFFFFF)
    phase = raw.astype(np.float64) / 100.0
    ticks = ticks[:tx * ty]
    us = ticks.astype(np.float64) / 100.0
    units, trips = (raw[:, 6] >> np.uint64(32)).astype(np.int64), (raw[:, 6] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    tiny, tinypx = (raw[:, 7] >> np.uint64(32)).astype(np.int64), (raw[:, 7] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    print("work: entries %d  tiny %d (bbox px %d)  units %d  row-loop trips %d (%.1f per unit)" % (cnt.sum(), tiny.sum(), tinypx.sum(), units.sum(), trips.sum(), trips.sum() / max(1, units.sum())))
    print("phase sums (us) in/load/setup/scan/units/out:", np.round(phase[:, :6].sum(axis=0), 0))
    print("  of which a wave's mean time at the phase's closing barrier (waiting for the workgroup's slowest wave):", np.round(barrier.sum(axis=0), 0),
          " = %.1f %% of the phases' time" % (100.0 * barrier.sum() / max(phase[:, :6].sum(), 1e-9)))
    order = np.argsort(-us)[:4]
    print("pass", p, "entries", int(cnt.sum()), "max bin", int(cnt.max()), "tile us: sum %.0f mean %.1f max %.1f" % (us.sum(), us.mean(), us.max()))
    for t in order: print("   tile (%2d,%2d) %7.1f us  bin %5d  phases" % (t % tx, t // tx, us[t], cnt[t]), np.round(phase[t, :6], 1))
    print("   corr(us, bin) = %.3f" % np.corrcoef(us, cnt)[0, 1])
    o2 = np.argsort(-phase[:, 4])[:6]
    print("   by units phase:")
    for t in o2: print("   tile (%2d,%2d) %7.1f us  bin %5d  units %6d trips %7d tiny %5d  phases" % (t % tx, t // tx, us[t], cnt[t], units[t], trips[t], tiny[t]), np.round(phase[t, :6], 1))
    h = np.histogram(us, bins=[0, 5, 10, 20, 30, 40, 50, 60, 80, 200])
    print("   tile-time histogram (us):", list(zip(h[1][:-1].astype(int), h[0])))
    for lo, hi in ((0, 1), (1, 64), (64, 256), (256, 1024), (1024, 100000)):
        m = (cnt >= lo) & (cnt < hi)
        print("   bins [%d,%d): %d tiles, time sum %.0f, units %d trips %d tiny %d, phases" % (lo, hi, m.sum(), us[m].sum(), units[m].sum(), trips[m].sum(), tiny[m].sum()), np.round(phase[m][:, :6].sum(axis=0), 0))
