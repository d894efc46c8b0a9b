"""Host-side bookkeeping of the row-stripe shard (mirrors ShardInfo / row_base in csrc/kernels_raster.hip).

Rows are cut into stripes of `stripe_rows` (even); stripe s belongs to rank s % ranks; the visibility
buffer is stored rank-major (rank r's stripes contiguous, `stripes_per_rank` stripes each, the tail
padded) so that ONE in-place all-gather reassembles the frame.
"""
import numpy as np


def pick_stripe_rows(height, ranks):
    """chordvis_pick_stripe_rows: even stripe height in [32, 256], at least two stripes per rank, that minimises
    padding / height + 18 / rows -- the idle share of the rank holding the padded stripe plus the share of (16-pixel) clusters
    that straddle two stripes and are set up by both owners; ties towards the taller stripe."""
    ranks = max(1, int(ranks))
    best = None
    for s in range(32, 258, 2):
        stripes = -(-height // s)
        per = -(-stripes // ranks)
        if ranks > 1 and per < 2 and s > 32:
            continue
        pad = per * ranks * s - height
        num = pad * s + 18 * height                       # cost = num / (height * s)
        if best is None or num * best[1] < best[0] * s or (num * best[1] == best[0] * s and s > best[1]):
            best = (num, s)
    return best[1]


class StripeLayout:
    def __init__(self, width, height, stripe_rows, ranks):
        assert stripe_rows >= 2 and stripe_rows % 2 == 0 and ranks >= 1
        self.width, self.height, self.stripe_rows, self.ranks = width, height, stripe_rows, ranks
        stripes = -(-height // stripe_rows)
        self.stripes_per_rank = -(-stripes // ranks)
        self.rows_padded = ranks * self.stripes_per_rank * stripe_rows if ranks > 1 else height
        self.words = self.rows_padded * width
        self.chunk_words = self.words // ranks

    def owner(self, y):
        return (np.asarray(y) // self.stripe_rows) % self.ranks

    def rank_major_row(self, y):
        """Row index of pixel row y inside the rank-major buffer."""
        y = np.asarray(y)
        if self.ranks == 1:
            return y
        stripe = y // self.stripe_rows
        return ((stripe % self.ranks) * self.stripes_per_rank + stripe // self.ranks) * self.stripe_rows + y % self.stripe_rows

    def to_rank_major(self, image):
        """(H, W) row-major -> (rows_padded, W) rank-major (padding rows zero)."""
        out = np.zeros((self.rows_padded, self.width), dtype=image.dtype)
        out[self.rank_major_row(np.arange(self.height))] = image
        return out

    def from_rank_major(self, buf):
        return buf.reshape(self.rows_padded, self.width)[self.rank_major_row(np.arange(self.height))]

    # mid-frame HZB mip-0 exchange: half-resolution rows, same stripe map
    def exchange_rows(self):
        return self.ranks * self.stripes_per_rank * (self.stripe_rows // 2)

    def exchange_row(self, y0):
        y0 = np.asarray(y0)
        half = self.stripe_rows // 2
        stripe = y0 // half
        return ((stripe % self.ranks) * self.stripes_per_rank + stripe // self.ranks) * half + y0 % half
