import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers as H, orc
from chord_amd import scenes, records as R, lib as L
from chord_amd.renderer import VisibilityRenderer, decode_visibility
scene, cam, view, iv = H.setup_scene(scenes.config4_street_x64)
W, Hh = cam.width, cam.height
want0 = orc.frame(scene, view, iv, H.ALL_FLAGS)
want1 = orc.frame(scene, view, iv, H.ALL_FLAGS, prev_hzb_min=want0["hzb_min"])
r = VisibilityRenderer(0); r.upload_scene(scene); r.allocate_gbuffer(W, Hh); r.set_view(view, iv, H.ALL_FLAGS)
r.set_debug(int(os.environ.get('DBG', '0')))
r.render_frame()
v0 = r.read_visibility(); print("frame0 equal", np.array_equal(v0, want0["vis"]))
mn, mx, rng = r.read_hzb(r.history_hzb()); print("hzb0 min equal", np.array_equal(mn, want0["hzb_min"]), "max", np.array_equal(mx, want0["hzb_max"]))
r.render_frame()
v1 = r.read_visibility()
import ctypes as C
nz = C.c_uint64(0); L.lib.chordvis_debug_slab_nonzero(r._ctx, C.byref(nz)); print('slab nonzero words after frame 1:', nz.value)
st = r.stats(); print("gpu stats", {k: st[k] for k in ("overflow", "countInstanceCulled", "countStage0Visible", "countStage0Rejected", "countStage1Visible", "trianglesSubmitted", "triangleRecords", "binEntries")})
print("oracle counts", want1["counts"], want1["stats"].trianglesSubmitted, want1["stats"].trianglesRastered)
bad = np.nonzero(v1 != want1["vis"])[0]
print("bad pixels", len(bad))
d, slot, tri = decode_visibility(want1["vis"][bad])
gd, gslot, gtri = decode_visibility(v1[bad])
print("missing slots (oracle):", np.unique(slot)[:20], "gpu slots there:", np.unique(gslot)[:20])
ys, xs = bad // W, bad % W
print("x range", xs.min(), xs.max(), "y range", ys.min(), ys.max(), "tiles", sorted(set(zip((xs // 64).tolist(), (ys // 64).tolist())))[:20])
cmds = want1["cmds"]
for s_ in np.unique(slot)[:5]:
    c = cmds[s_]; print("slot", s_, "cmd", c)

d = orc.hzb_desc(W, Hh)
vis0, rej0 = orc.hzb_culling(scene, view, H.ALL_FLAGS, 0, d, want0["hzb_min"], want1["cmds"])
s0 = set(vis0["slot"].tolist())
ms = np.unique(slot)
print("missing slots in stage0-visible:", sum(int(x) in s0 for x in ms), "of", len(ms))

# ---- is the missing cluster in its tile's pass-1 bin? ----
tilesX, tilesY = (W + 63) // 64, (Hh + 63) // 64
tiles = tilesX * tilesY
STRIDE, BINCAP, MAXCH, CH = 16, 16384, 240, 1024
def rd(which, off, nbytes, dtype):
    buf = np.zeros(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
    assert L.lib.chordvis_debug_read(r._ctx, which, off, nbytes, buf.ctypes.data) == 0
    return buf
counts = rd(0, 0, 2 * tiles * STRIDE * 4, np.uint32).reshape(2, tiles, STRIDE)
recC_dt = np.dtype([("X0", "<i4"), ("Y0", "<i4"), ("d", "<i2", 4), ("z", "<f4", 3), ("payload", "<u4")])
recW_dt = np.dtype([("X", "<i4", 3), ("Y", "<i4", 3), ("z", "<f4", 3), ("payload", "<u4"), ("two", "<u4"), ("pad", "<u4")])
seen_tiles = 0
for (tx, ty) in sorted(set(zip((xs // 64).tolist(), (ys // 64).tolist())))[:6]:
    t = ty * tilesX + tx
    n = int(counts[1, t, 0]); tick = int(counts[1, t, 1])
    ent = rd(1, (tiles + t) * BINCAP * 4, min(n, BINCAP) * 4, np.uint32)
    if n > BINCAP:
        tab = rd(2, ((tiles + t) * MAXCH) * 8, MAXCH * 8, np.uint64)
        extra = []
        for j in range((n - BINCAP + CH - 1) // CH):
            cid = int(tab[j] & 0xFFFFFFFF)
            m = min(CH, n - BINCAP - j * CH)
            extra.append(rd(3, (32768 * CH + cid * CH) * 4, m * 4, np.uint32))
        ent = np.concatenate([ent] + extra)
    wide = (ent & 0x80000000) != 0
    pay = np.zeros(len(ent), np.uint32)
    ci = ent[~wide].astype(np.int64)
    if len(ci):
        lo, hi = int(ci.min()), int(ci.max())
        rc = rd(4, lo * 32, (hi - lo + 1) * 32, recC_dt)
        pay[~wide] = rc["payload"][ci - lo]
    wi = (ent[wide] & 0x7FFFFFFF).astype(np.int64)
    if len(wi):
        lo, hi = int(wi.min()), int(wi.max())
        rw = rd(5, lo * 48, (hi - lo + 1) * 48, recW_dt)
        pay[wide] = rw["payload"][wi - lo]
    slots_in_bin = ((pay >> 8) & 0xFFFFFF).astype(np.int64) - 1
    miss_here = np.unique(slot[(xs // 64 == tx) & (ys // 64 == ty)])
    pos = {int(m): np.nonzero(slots_in_bin == m)[0] for m in miss_here}
    print("tile", (tx, ty), "pass-1 bin n", n, "ticket", tick, "slices", (n + 2047) // 2048 if n > 6144 else 1,
          {m: (len(p), (int(p.min()), int(p.max())) if len(p) else None) for m, p in pos.items()})
