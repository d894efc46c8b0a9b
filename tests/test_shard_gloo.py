"""The N > 1 frame protocol on CPU: world_size-2 `gloo`, every rank renders its own screen tiles with the
ORACLE (the checker stands in for the HIP kernels here), the exchanges of DESIGN.md 6 run through
torch.distributed, and every rank must end with the single-rank frame.  Plus the tile map itself (host code of the library)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _border_edges(lay):
    o = lay.owners.reshape(lay.tiles_y, lay.tiles_x)
    return int((o[:, 1:] != o[:, :-1]).sum() + (o[1:] != o[:-1]).sum())


def test_default_tile_map_is_compact_and_within_the_chunk(built_lib):
    """chordvis_tile_layout without loads: every tile owned, every rank at most ceil(tiles / ranks) tiles and at least one fewer,
    and the regions are compact -- the border between ranks stays within 2x of what squares of equal area would share."""
    from chord_amd.sharding import TileLayout
    for (w, h) in ((3840, 2160), (1920, 1080), (1280, 720), (257, 131), (640, 360), (4096, 4096), (64, 64), (2160, 3840)):
        for n in (1, 2, 3, 4, 5, 7, 8, 16):
            lay = TileLayout(w, h, n)
            cnt = np.bincount(lay.owners, minlength=n)
            assert cnt.max() <= lay.slots_per_rank and cnt.sum() == lay.tiles
            if lay.tiles >= n:
                assert cnt.max() - cnt.min() <= 1, (w, h, n, cnt)
            # slots are a bijection onto [0, ranks * slots_per_rank) restricted to the rank's chunk
            assert len(set(lay.slot.tolist())) == lay.tiles
            assert ((lay.slot // lay.slots_per_rank) == lay.owners).all()
            if n > 1 and lay.tiles >= 64 * n:
                # n squares of area tiles / n share about 2 sqrt(n) (sqrt(n) - 1) sides of sqrt(tiles / n) tile edges each ... loosely:
                ideal = 2.0 * np.sqrt(lay.tiles) * (np.sqrt(n) - 1.0)
                assert _border_edges(lay) <= 2.0 * ideal + 8, (w, h, n, _border_edges(lay), ideal)
    lay = TileLayout(3840, 2160, 8)
    assert lay.slots_per_rank == 255 and lay.words == 8 * 255 * 4096
    img = (np.arange(2160 * 3840, dtype=np.uint64).reshape(2160, 3840) * np.uint64(2654435761)) | np.uint64(1)
    assert np.array_equal(lay.from_rank_major(lay.to_rank_major(img)), img)
    lay = TileLayout(257, 131, 3)
    img = np.arange(131 * 257, dtype=np.uint64).reshape(131, 257) + np.uint64(7)
    assert np.array_equal(lay.from_rank_major(lay.to_rank_major(img)), img)


def test_weighted_tile_map_balances_hotspots_and_skies(built_lib):
    """chordvis_tile_layout with loads: the heaviest rank stays near the mean where the tiles allow it (no tile can be split),
    nobody exceeds its chunk, uniform loads give the default map, and the result is a pure function of its inputs."""
    from chord_amd.sharding import TileLayout, tile_layout
    w, h, n = 3840, 2160, 8
    base = TileLayout(w, h, n)
    tiles, tx, ty = base.tiles, base.tiles_x, base.tiles_y
    xs, ys = np.meshgrid(np.arange(tx) * 64 + 32, np.arange(ty) * 64 + 32)

    from chord_amd import lib as L
    cap = int(L.lib.chordvis_tile_slot_capacity(w, h, n))
    assert cap == -(-tiles * 5 // (4 * n))                                                 # a quarter more than ceil(tiles / ranks)

    def check(loads, limit):
        loads = loads.astype(np.uint32).reshape(-1)
        owners = tile_layout(w, h, n, loads, cap)
        assert np.array_equal(owners, tile_layout(w, h, n, loads.copy(), cap))            # deterministic
        cnt = np.bincount(owners, minlength=n)
        assert cnt.max() <= cap and cnt.sum() == tiles
        per = np.bincount(owners, weights=loads.astype(np.float64), minlength=n)
        ratio = per.max() / per.mean()
        bound = max(limit, loads.max() / per.mean() * 1.02)                               # (a single tile cannot be split)
        assert ratio <= bound, (ratio, bound, per.tolist())
        return owners, ratio

    # uniform: the default map
    owners, _ = check(np.full(tiles, 1000), 1.01)
    assert np.array_equal(owners, base.owners)
    # BASELINE config 5's hotspot: cluster centres Gaussian, sigma = 64 px, around the screen centre (12 % of the frame in one tile)
    g = np.exp(-((xs - w / 2) ** 2 + (ys - h / 2) ** 2) / (2 * 64.0 ** 2))
    from math import erf
    cdf = lambda v: 0.5 * (1 + erf(v / np.sqrt(2)))
    px = np.array([cdf((x * 64 + 64 - w / 2) / 64) - cdf((x * 64 - w / 2) / 64) for x in range(tx)])
    py = np.array([cdf((y * 64 + 64 - h / 2) / 64) - cdf((y * 64 - h / 2) / 64) for y in range(ty)])
    hot = np.outer(py, px) * 8.4e6
    owners, ratio = check(hot, 1.15)
    assert ratio <= 1.15, ratio                                                            # stripes of 136 rows: 3.7
    # a frame whose upper third is sky: all ranks share the loaded part
    sky = np.where(ys < h / 3, 0, 500 + (xs // 64) % 7 * 40)
    check(sky, 1.05)
    # a smooth gradient with noise (tile loads 1 : 3.4 across the screen: balancing it needs regions of unequal area, and a
    # rank's area is capped at 1.25x the mean)
    rng = np.random.default_rng(5)
    check((200 + 3 * (xs // 64) + 9 * (ys // 64)) * rng.uniform(0.7, 1.3, size=xs.shape), 1.12)
    check((400 + 2 * (xs // 64) + 3 * (ys // 64)) * rng.uniform(0.7, 1.3, size=xs.shape), 1.03)
    # everything in two tiles
    two = np.zeros(tiles); two[100] = 7; two[1500] = 9
    check(two, 1.0)
    # measured loads (chordvis_read_tile_loads on the GPU box, committed): BASELINE config 5's hotspot and config 4 at 4K
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    check(np.load(os.path.join(prof, "r04_tile_loads_config5_hotspot.npy")), 1.06)        # (its heaviest tile alone is a mean rank's load)
    check(np.load(os.path.join(prof, "r04_tile_loads_config4.npy")), 1.08)
    # the light tiles (fillers) join the region next to them on the curve instead of going out rank by rank: the border between
    # ranks -- clusters on it are set up twice -- of the sky frame (378 tile edges when the fillers were dealt in rank order),
    # of config 4's measured frame (615) and of the hotspot's (275)
    def border(o):
        g = o.reshape(ty, tx)
        return int((g[:, 1:] != g[:, :-1]).sum() + (g[1:] != g[:-1]).sum())
    assert border(check(sky, 1.05)[0]) <= 340
    assert border(check(np.load(os.path.join(prof, "r04_tile_loads_config4.npy")), 1.08)[0]) <= 560
    assert border(check(np.load(os.path.join(prof, "r04_tile_loads_config5_hotspot.npy")), 1.06)[0]) <= 265


# ---- the sharded group cull's exchange on the host (mirrors kernels_cull.hip: group_cull_masks_kernel / group_mask_unpack_kernel) ----
def _group_instance_table(scene):
    """The flattened (object, group) instances in the library's order (DGroupRef, chordvis_upload_scene): (owner object [G], global
    meshlet ids [G, 4], -1 = no such meshlet)."""
    O = scene.objects
    prim = scene.primitives[O["GLTFPrimitiveDetail"]]
    counts = prim["meshletGroupCount"].astype(np.int64)
    owner = np.repeat(np.arange(len(O)), counts)
    gi = np.arange(len(owner)) - np.repeat(np.cumsum(counts) - counts, counts)
    p = prim[owner]
    g = scene.groups[p["meshletGroupOffset"].astype(np.int64) + gi]
    meshlet = np.full((len(owner), 4), -1, dtype=np.int64)
    for i in range(4):
        has = g["meshletCount"] > i
        idx = p["meshletGroupIndicesOffset"][has].astype(np.int64) + g["meshletOffset"][has].astype(np.int64) + i
        meshlet[has, i] = p["meshletOffset"][has].astype(np.int64) + scene.group_indices[idx]
    return owner, meshlet


def _cluster_rank_mask(scene, iv, lay, obj, mid, nobody):
    """Bit r = the cluster's pixel rectangle (8 projected AABB corners, one pixel of slack) touches a tile of rank r; a corner at or
    behind the camera plane: every rank; a rectangle off screen: the one rank `nobody` (the cluster still takes a slot)."""
    import spec_np as S
    f32 = np.float32
    M = S.mat(scene.objects["localToTranslatedWorld"][obj][None])[0]
    VP = S.mat(iv["translatedWorldToClip"])[0]
    mvp = S.mul_mm(VP, M)
    m = scene.meshlets[mid]
    lo, hi = m["posMin"].astype(f32), m["posMax"].astype(f32)
    xs, ys = [], []
    W, H = f32(lay.width), f32(lay.height)
    with np.errstate(all="ignore"):
        for q in range(8):
            c = [hi[k] if (q >> k) & 1 else lo[k] for k in range(3)]
            h = S.mul_mv(mvp, c[0], c[1], c[2])
            x = (h[0] / h[3] * f32(0.5) + f32(0.5)) * W
            y = (h[1] / h[3] * f32(-0.5) + f32(0.5)) * H
            if not (h[3] > f32(1e-6)) or not (abs(y) < f32(1e7)) or not (abs(x) < f32(1e7)):
                return (1 << lay.ranks) - 1
            xs.append(float(x)); ys.append(float(y))
    x0, x1 = max(int(np.floor(min(xs))) - 1, 0), min(int(np.ceil(max(xs))) + 1, lay.width - 1)
    y0, y1 = max(int(np.floor(min(ys))) - 1, 0), min(int(np.ceil(max(ys))) + 1, lay.height - 1)
    if x1 < x0 or y1 < y0:
        return 1 << nobody
    mask = 0
    for ty in range(y0 // 64, y1 // 64 + 1):
        for tx in range(x0 // 64, x1 // 64 + 1):
            mask |= 1 << int(lay.owners[ty * lay.tiles_x + tx])
    return mask


def _cull_exchange_chunk(scene, iv, lay, cmds, table, rank, chunk_blocks):
    """This rank's chunk of the cull exchange buffer: chunk_blocks x 256 words (byte i = the rank set of meshlet i of the group
    instance, 0 = culled) followed by chunk_blocks triangle sums -- from the commands of the rank's OWN range of group instances."""
    owner, meshlet = table
    first, last = rank * chunk_blocks * 256, (rank + 1) * chunk_blocks * 256
    where = {}
    for t in range(first, min(last, len(owner))):
        for i in range(4):
            if meshlet[t, i] >= 0:
                assert (int(owner[t]), int(meshlet[t, i])) not in where
                where[(int(owner[t]), int(meshlet[t, i]))] = (t, i)
    chunk = np.zeros(chunk_blocks * 257, dtype=np.uint32)
    for cmd in cmds:
        key = (int(cmd["objectId"]), int(cmd["meshletId"]))
        if key in where:
            t, i = where[key]
            chunk[t - first] |= np.uint32(_cluster_rank_mask(scene, iv, lay, key[0], key[1], t % lay.ranks) << (8 * i))
            chunk[chunk_blocks * 256 + (t - first) // 256] += (int(scene.meshlets["vertexTriangleCount"][key[1]]) >> 8) & 0xFF
    return chunk


def _cull_exchange_unpack(words, table, world, chunk_blocks, rank):
    """All ranks' chunks -> (the full command list in (object, group, meshlet) order with its slots, this rank's own commands, the
    triangle total): what prefix + scatter make of the exchanged words."""
    from chord_amd import records as R
    owner, meshlet = table
    full, mine, tris = [], [], 0
    for t in range(len(owner)):
        src, lb = divmod(t // 256, chunk_blocks)
        w = int(words[src * chunk_blocks * 257 + lb * 256 + t % 256])
        for i in range(4):
            byte = (w >> (8 * i)) & 0xFF
            if byte:
                cmd = (int(owner[t]), int(meshlet[t, i]), len(full))
                full.append(cmd)
                if (byte >> rank) & 1:
                    mine.append(cmd)
    for src in range(world):
        tris += int(words[src * chunk_blocks * 257 + chunk_blocks * 256: (src + 1) * chunk_blocks * 257].sum())
    return np.array(full, dtype=R.DRAW_CMD).reshape(-1), np.array(mine, dtype=R.DRAW_CMD).reshape(-1), tris


def _worker(rank, world, port, tmp):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers as H
    import orc
    from chord_amd import lib as L, scenes, sharding as S

    scene, cam = scenes.small_test_scene(256, 160, seed=29)
    w, h, flags = cam.width, cam.height, H.ALL_FLAGS
    L.fill_objects(scene, cam)
    view, iv = L.make_views(cam)
    # an interleaved map (every tile border is a rank border) on frame 0, the library's default on frame 1
    lay_default = S.TileLayout(w, h, world)
    lay_checker = S.TileLayout(w, h, world, owners=[(t % lay_default.tiles_x + t // lay_default.tiles_x) % world for t in range(lay_default.tiles)])
    desc = orc.hzb_desc(w, h)
    prev_hzb = None

    def all_gather_rows(rows):
        """all-gather of a [ranks * n, k] array by rank chunks (raw bytes: neither gloo nor NCCL/RCCL has a 16-bit integer type)"""
        n = rows.shape[0] // world
        mine = np.ascontiguousarray(rows[rank * n:(rank + 1) * n]).view(np.uint8).reshape(n, -1)
        parts = [torch.zeros(mine.shape, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(mine.copy()))
        return torch.cat(parts).numpy().view(rows.dtype).reshape(rows.shape)

    for frame in range(2):
        lay = lay_checker if frame == 0 else lay_default
        shard = (lay.owners, lay.tiles_x, world, rank)
        own_px = lay.owner_of_pixels() == rank
        want = orc.frame(scene, view, iv, flags, prev_hzb_min=prev_hzb)       # the single-rank frame
        # ---- phase cull + exchange #0: the group cull sharded by ranges of group instances; what travels is one word per group
        #      instance (byte i = the set of ranks whose tiles meshlet i touches, 0 = culled), a fixed-size all-gather ----------
        cmds = orc.instance_culling(scene, view, iv, flags)                  # (what ONE rank culling every group gets: the reference)
        table = _group_instance_table(scene)
        chunk_blocks = -(-(-(-len(table[0]) // 256)) // world)
        ex0 = np.zeros((world, chunk_blocks * 257), dtype=np.uint32)
        ex0[rank] = _cull_exchange_chunk(scene, iv, lay, cmds, table, rank, chunk_blocks)
        ex0 = all_gather_rows(ex0)
        full_cmds, my_cmds, tris = _cull_exchange_unpack(ex0.reshape(-1), table, world, chunk_blocks, rank)
        assert np.array_equal(full_cmds, cmds), "the list rebuilt from the exchanged rank masks differs from the single-rank cull (slots included)"
        assert tris == int(((scene.meshlets["vertexTriangleCount"][cmds["meshletId"]] >> 8) & 0xFF).sum())
        assert 0 < len(my_cmds) <= len(cmds)
        # ---- phase a: phase-0 HZB cull and raster of the rank's OWN commands (the clusters that touch its tiles) --------
        if prev_hzb is not None:
            vis_list, rej = orc.hzb_culling(scene, view, flags, 0, desc, prev_hzb, my_cmds)
            full_vis_list, _ = orc.hzb_culling(scene, view, flags, 0, desc, prev_hzb, cmds)
        else:
            vis_list, rej = my_cmds, my_cmds[:0]
            full_vis_list = cmds
        mine, _ = orc.raster(scene, iv, vis_list, w, h, shard=shard)
        # (dropping the clusters that touch none of the rank's tiles changes nothing in its tiles)
        assert np.array_equal(mine, orc.raster(scene, iv, full_vis_list, w, h, shard=shard)[0])
        mine = mine.reshape(h, w)
        assert not mine[~own_px].any()                                         # only owned tiles written
        if prev_hzb is not None:
            # ---- exchange #1: the owned tiles' HZB texels (mips 0..5 of the min chain), one slot per tile ---------
            _, mn_local, _, _ = orc.hzb_build(mine.reshape(-1), w, h)
            slots = all_gather_rows(lay.pack_hzb_slots(desc, mn_local, rank))
            chain = np.zeros(desc.totalTexels, dtype=np.uint16)
            lay.unpack_hzb_slots(desc, slots, chain)
            S.hzb_tail(desc, chain)
            # must equal the HZB of the full stage-0 image, which the single-rank oracle builds internally; rebuild it here to compare
            full0, _ = orc.raster(scene, iv, full_vis_list, w, h)
            _, want_mn, _, _ = orc.hzb_build(full0, w, h)
            for l in range(desc.mipCount):
                mw, mh = desc.mip_dims(l); vw, vh = desc.valid_dims(l)
                a = chain[desc.mipOffset[l]: desc.mipOffset[l] + mw * mh].reshape(mh, mw)[:vh, :vw]
                b = want_mn[desc.mipOffset[l]: desc.mipOffset[l] + mw * mh].reshape(mh, mw)[:vh, :vw]
                assert np.array_equal(a, b), "mid-frame chain level %d" % l
            # ---- phase b: phase-1 cull against the shared HZB, raster own tiles -------------------
            vis1, _ = orc.hzb_culling(scene, view, flags, 1, desc, want_mn, rej)
            mine = orc.raster(scene, iv, vis1, w, h, vis=mine.reshape(-1).copy(), shard=shard)[0].reshape(h, w)
        # ---- exchange #2 (small): min | max texels and the valid range of the owned tiles -> the history chain ----------
        _, mn_local, mx_local, _ = orc.hzb_build(mine.reshape(-1), w, h, want_max=True)
        fin = np.zeros((world * lay.slots_per_rank, S.HZB_FINAL_SLOT_HALVES), dtype=np.uint16)
        lay.pack_hzb_slots(desc, mn_local, rank, out=fin)
        lay.pack_hzb_slots(desc, mx_local, rank, out=fin, offset=S.HZB_FINAL_MAX_OFFSET)
        rngs = S.tile_ranges(lay, mine)
        for t in np.nonzero(lay.owners == rank)[0]:
            fin[lay.slot[t], S.HZB_FINAL_RANGE_OFFSET: S.HZB_FINAL_RANGE_OFFSET + 4] = rngs[t].view(np.uint16)
        fin = all_gather_rows(fin)
        hmin = np.zeros(desc.totalTexels, dtype=np.uint16); hmax = np.zeros(desc.totalTexels, dtype=np.uint16)
        lay.unpack_hzb_slots(desc, fin, hmin); lay.unpack_hzb_slots(desc, fin, hmax, offset=S.HZB_FINAL_MAX_OFFSET)
        S.hzb_tail(desc, hmin); S.hzb_tail(desc, hmax, is_max=True)
        pairs = np.stack([fin[lay.slot[t], S.HZB_FINAL_RANGE_OFFSET: S.HZB_FINAL_RANGE_OFFSET + 4].view(np.uint32) for t in range(lay.tiles)])
        got_range = np.array([pairs[:, 0].min(), pairs[:, 1].max()], dtype=np.uint32)
        for l in range(desc.mipCount):
            mw, mh = desc.mip_dims(l); vw, vh = desc.valid_dims(l)
            sl = slice(desc.mipOffset[l], desc.mipOffset[l] + mw * mh)
            assert np.array_equal(hmin[sl].reshape(mh, mw)[:vh, :vw], want["hzb_min"][sl].reshape(mh, mw)[:vh, :vw]), "history min level %d" % l
            assert np.array_equal(hmax[sl].reshape(mh, mw)[:vh, :vw], want["hzb_max"][sl].reshape(mh, mw)[:vh, :vw]), "history max level %d" % l
        assert np.array_equal(got_range, want["valid_range"])
        # ---- exchange #3: the visibility words, rank-major tile slots, then de-tile -------------------------
        rm = all_gather_rows(lay.to_rank_major(mine).reshape(world * lay.slots_per_rank, -1))
        full = lay.from_rank_major(rm)
        assert np.array_equal(full.reshape(-1), want["vis"]), "rank %d frame %d differs from the single-rank frame" % (rank, frame)
        prev_hzb = want["hzb_min"]
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")


def test_two_rank_frame_over_gloo(tmp_path, built_lib):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
