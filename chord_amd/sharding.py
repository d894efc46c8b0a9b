"""Host-side bookkeeping of the screen-tile shard (mirrors ShardInfo in csrc/device_layer.h and the slot layout of DESIGN.md 6).

The screen is cut into 64 x 64 tiles; every tile has one owner (a table, the same on every rank: `tile_layout`, which is the
library's own host function chordvis_tile_layout -- there is ONE implementation of the map).  The visibility buffer of a
sharded context is rank-major and tile-linear: tile t lives in slot  owner * slots_per_rank + (index among the owner's
tiles, by tile index),  a slot is 64 rows of 64 words, so ONE in-place all-gather of slots_per_rank * 4096 words per rank
reassembles the frame.  The HZB texels a tile covers (mips 0..5: 32^2 + 16^2 + ... + 1 = 1365 halves) travel in slots of the
same numbering.
"""
import numpy as np

TILE = 64
HZB_TILE_TEXELS = 1365
HZB_SLOT_HALVES = 1408            # mid-frame exchange: the min chain's texels of one tile, padded
HZB_FINAL_SLOT_HALVES = 2832      # end of frame: min | max (at 1408) | {range min, range max, bin entries, 0} as uint32 (at 2816) | padding
HZB_FINAL_MAX_OFFSET = 1408
HZB_FINAL_RANGE_OFFSET = 2816


def level_offset(l):
    """Offset of level l's (32 >> l)^2 texels inside a tile's HZB slot."""
    return HZB_TILE_TEXELS - (HZB_TILE_TEXELS >> (2 * l))


def tile_layout(width, height, ranks, loads=None, max_tiles_per_rank=0):
    """chordvis_tile_layout: owner of every tile (row-major over the tile grid)."""
    from . import lib as L
    tiles = int(L.lib.chordvis_tile_count(width, height))
    owners = np.zeros(tiles, dtype=np.uint8)
    lp = None
    if loads is not None:
        loads = np.ascontiguousarray(loads, dtype=np.uint32)
        assert len(loads) == tiles
        lp = loads.ctypes.data
    rc = L.lib.chordvis_tile_layout(width, height, ranks, lp, int(max_tiles_per_rank), owners.ctypes.data)
    if rc != L.OK:
        raise L.ChordvisError("chordvis_tile_layout(%d, %d, %d) failed with %d" % (width, height, ranks, rc))
    return owners


class TileLayout:
    def __init__(self, width, height, ranks, owners=None):
        self.width, self.height, self.ranks = width, height, ranks
        self.tiles_x, self.tiles_y = -(-width // TILE), -(-height // TILE)
        self.tiles = self.tiles_x * self.tiles_y
        self.owners = np.asarray(owners, dtype=np.uint8) if owners is not None else tile_layout(width, height, ranks)
        assert len(self.owners) == self.tiles and int(self.owners.max()) < ranks
        # a rank's chunk of every rank-major buffer: as many slots as the largest rank owns (what an all-gather moves per rank)
        self.slots_per_rank = int(np.bincount(self.owners, minlength=ranks).max())
        used = np.zeros(ranks, dtype=np.int64)
        self.slot = np.zeros(self.tiles, dtype=np.int64)
        for t in range(self.tiles):
            o = int(self.owners[t])
            self.slot[t] = o * self.slots_per_rank + used[o]
            used[o] += 1
        assert int(used.max()) <= self.slots_per_rank
        self.words = ranks * self.slots_per_rank * TILE * TILE if ranks > 1 else width * height
        self.chunk_words = self.words // ranks

    def tile_of(self, x, y):
        return (np.asarray(y) // TILE) * self.tiles_x + np.asarray(x) // TILE

    def owner_of_pixels(self):
        """(H, W) owner of every pixel."""
        ys, xs = np.mgrid[0:self.height, 0:self.width]
        return self.owners[self.tile_of(xs, ys)]

    def to_rank_major(self, image):
        """(H, W) row-major -> flat rank-major tile slots (padding zero)."""
        out = np.zeros(self.ranks * self.slots_per_rank * TILE * TILE, dtype=image.dtype)
        slots = out.reshape(-1, TILE, TILE)
        for t in range(self.tiles):
            ty, tx = divmod(t, self.tiles_x)
            blk = image[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
            slots[self.slot[t], :blk.shape[0], :blk.shape[1]] = blk
        return out

    def from_rank_major(self, buf):
        slots = np.asarray(buf).reshape(-1, TILE, TILE)
        out = np.zeros((self.height, self.width), dtype=slots.dtype)
        for t in range(self.tiles):
            ty, tx = divmod(t, self.tiles_x)
            h, w = min(TILE, self.height - ty * TILE), min(TILE, self.width - tx * TILE)
            out[ty * TILE:ty * TILE + h, tx * TILE:tx * TILE + w] = slots[self.slot[t], :h, :w]
        return out

    # ---- HZB texels of a tile in exchange slots (what raster_tile_kernel writes and hzb_untile_kernel reads) -----------------
    def pack_hzb_slots(self, desc, chain, rank, slot_halves=HZB_SLOT_HALVES, out=None, offset=0):
        """Texels (mips 0..5) of `rank`'s tiles out of a flat uint16 chain -> [ranks * slots_per_rank, slot_halves] uint16."""
        if out is None:
            out = np.zeros((self.ranks * self.slots_per_rank, slot_halves), dtype=np.uint16)
        for t in np.nonzero(self.owners == rank)[0]:
            ty, tx = divmod(int(t), self.tiles_x)
            for l in range(min(6, desc.mipCount)):
                side = 32 >> l
                mw, _ = desc.mip_dims(l)
                vw, vh = desc.valid_dims(l)
                lvl = chain[desc.mipOffset[l]: desc.mipOffset[l] + mw * desc.mip_dims(l)[1]].reshape(-1, mw)
                x0, y0 = tx * side, ty * side
                w, h = max(0, min(side, vw - x0)), max(0, min(side, vh - y0))
                blk = np.zeros((side, side), dtype=np.uint16)
                if w and h:
                    blk[:h, :w] = lvl[y0:y0 + h, x0:x0 + w]
                out[self.slot[t], offset + level_offset(l): offset + level_offset(l) + side * side] = blk.reshape(-1)
        return out

    def unpack_hzb_slots(self, desc, slots, chain, offset=0):
        """All tiles' slots -> mips 0..5 of the flat chain (in place), valid extents only."""
        for t in range(self.tiles):
            ty, tx = divmod(t, self.tiles_x)
            for l in range(min(6, desc.mipCount)):
                side = 32 >> l
                mw, mh = desc.mip_dims(l)
                vw, vh = desc.valid_dims(l)
                lvl = chain[desc.mipOffset[l]: desc.mipOffset[l] + mw * mh].reshape(mh, mw)
                x0, y0 = tx * side, ty * side
                w, h = max(0, min(side, vw - x0)), max(0, min(side, vh - y0))
                if w and h:
                    blk = slots[self.slot[t], offset + level_offset(l): offset + level_offset(l) + side * side].reshape(side, side)
                    lvl[y0:y0 + h, x0:x0 + w] = blk[:h, :w]
        return chain


def hzb_tail(desc, chain, is_max=False, first=6):
    """Levels first.. of a flat uint16 chain from level first - 1 (2 x 2 taps clamped to the valid extent of the level below;
    depths are non-negative, so binary16 order is integer order).  The max chain's +1 ulp is applied at level 5 only (hzb.hlsl:67-71),
    so levels 6.. are plain reductions of stored values."""
    for l in range(first, desc.mipCount):
        mw, mh = desc.mip_dims(l)
        vw, vh = desc.valid_dims(l)
        pmw, pmh = desc.mip_dims(l - 1)
        gw, gh = desc.valid_dims(l - 1)
        prev = chain[desc.mipOffset[l - 1]: desc.mipOffset[l - 1] + pmw * pmh].reshape(pmh, pmw)
        lvl = chain[desc.mipOffset[l]: desc.mipOffset[l] + mw * mh].reshape(mh, mw)
        ys, xs = np.arange(vh), np.arange(vw)
        taps = [prev[np.minimum(2 * ys + j, gh - 1)[:, None], np.minimum(2 * xs + i, gw - 1)[None, :]] for j in (0, 1) for i in (0, 1)]
        red = np.maximum.reduce(taps) if is_max else np.minimum.reduce(taps)
        lvl[:vh, :vw] = red
    return chain


def tile_ranges(layout, vis):
    """Per tile {min bits of depths in (0, 1), max bits of depths > 0} of a row-major uint64 image (hzb.hlsl:163-176)."""
    depth = (np.asarray(vis, dtype=np.uint64).reshape(layout.height, layout.width) >> np.uint64(32)).astype(np.uint32)
    f = depth.view(np.float32)
    out = np.zeros((layout.tiles, 2), dtype=np.uint32)
    out[:, 0] = 0xFFFFFFFF
    for t in range(layout.tiles):
        ty, tx = divmod(t, layout.tiles_x)
        d = depth[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
        ff = f[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
        drawn = ff > 0
        if drawn.any():
            out[t, 1] = d[drawn].max()
            inner = drawn & (ff < 1.0)
            if inner.any():
                out[t, 0] = d[inner].min()
    return out
