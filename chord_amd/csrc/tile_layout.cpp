// Which rank owns which 64x64 screen tile of a sharded frame ("screen tiles (e.g. 64x64) assigned ... to ranks",
// north_star: "frames shard across the 8 GPUs of one node by screen tile").  Host code, no device work; every rank computes the
// same table from the same inputs (integer arithmetic only, every tie broken by index).  The reference is single-device
// (graphics.cpp:524-548): nothing here has a counterpart in it.
//
// What a map has to do, measured in round 3 on row stripes (profiles/r03_shard_time_*):
//   * a cluster that touches tiles of two ranks is set up by both: ownership regions should be COMPACT -- a cluster of ~10 px
//     straddles a border of a region 16 tiles across with probability ~4 %, a 136-row stripe's with 7 %, a single tile's with 34 %;
//   * a frame waits for its slowest rank: regions should carry EQUAL WORK, and a hotspot two tiles wide (BASELINE config 5's
//     hotspot variant: one tile holds 12 % of the frame) can only be spread tile by tile;
//   * one in-place all-gather reassembles the image, rank chunks of equal size: every rank owns AT MOST `cap` tiles -- its chunk
//     of slots.  The gather moves ranks x (the largest rank's tile count) slots, so cap = ceil(tiles / ranks) moves no padding
//     but ties load to area; chordvis contexts allocate a quarter more (CHORD_TILE_SLACK) and a weighted map may use it: a
//     region of light tiles can then be up to 1.25x the mean area (smooth gradients balance to within a few percent), at the
//     price of up to 25 % more bytes in the image gather.
// So: tiles are ordered along a space-filling curve over the tile grid (a generalised Hilbert curve: contiguous runs of it are
// compact blobs whatever the grid's aspect), and
//   * without load figures the curve is cut into `ranks` runs of equal length;
//   * with them (bin entries per tile of a rendered frame, chordvis_read_tile_loads) tiles heavier than a quarter of a rank's
//     share are placed one by one, heaviest first, on the least loaded rank; the others are cut into contiguous runs along the
//     curve such that the heaviest rank is as light as possible (bisection over the bound); the lightest tiles -- together a
//     sixteenth of the load at most, each a quarter of the mean tile at most -- are fillers that bring every rank to its tile
//     count: each joins the region next to it on the curve while that region is below its count, the rest go out in runs.

#include "device_layer.h"

#include <algorithm>
#include <cstdint>
#include <vector>

namespace {

inline int sgn(int v) { return (v > 0) - (v < 0); }
inline int fdiv2(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }      // floor(v / 2)

// Generalised Hilbert curve over a w x h rectangle: visits every cell once, consecutive cells are neighbours (with at most
// one diagonal step on odd-by-odd rectangles), and every contiguous run is a compact region.  (x, y): start; (ax, ay): the
// major axis vector; (bx, by): the minor one.
void gilbert(std::vector<uint32_t>& out, uint32_t pitch, int x, int y, int ax, int ay, int bx, int by)
{
    const int w = std::abs(ax + ay), h = std::abs(bx + by);
    const int dax = sgn(ax), day = sgn(ay), dbx = sgn(bx), dby = sgn(by);
    if (h == 1) { for (int i = 0; i < w; i++) { out.push_back((uint32_t)y * pitch + (uint32_t)x); x += dax; y += day; } return; }
    if (w == 1) { for (int i = 0; i < h; i++) { out.push_back((uint32_t)y * pitch + (uint32_t)x); x += dbx; y += dby; } return; }
    int ax2 = fdiv2(ax), ay2 = fdiv2(ay), bx2 = fdiv2(bx), by2 = fdiv2(by);
    const int w2 = std::abs(ax2 + ay2), h2 = std::abs(bx2 + by2);
    if (2 * w > 3 * h) {
        if ((w2 & 1) && w > 2) { ax2 += dax; ay2 += day; }                 // prefer even steps
        gilbert(out, pitch, x, y, ax2, ay2, bx, by);
        gilbert(out, pitch, x + ax2, y + ay2, ax - ax2, ay - ay2, bx, by);
    } else {
        if ((h2 & 1) && h > 2) { bx2 += dbx; by2 += dby; }
        gilbert(out, pitch, x, y, bx2, by2, ax2, ay2);
        gilbert(out, pitch, x + bx2, y + by2, ax, ay, bx - bx2, by - by2);
        gilbert(out, pitch, x + (ax - dax) + (bx2 - dbx), y + (ay - day) + (by2 - dby), -bx2, -by2, -(ax - ax2), -(ay - ay2));
    }
}

} // namespace

namespace chord {

int tile_layout(uint32_t tilesX, uint32_t tilesY, uint32_t ranks, const uint32_t* loads, uint32_t cap, uint8_t* owners)
{
    if (!owners || tilesX == 0 || tilesY == 0 || ranks == 0 || ranks > 255u) return CHORDVIS_E_INVALID;
    const uint32_t tiles = tilesX * tilesY, N = ranks;
    const uint32_t S = std::max(cap, (tiles + N - 1u) / N);                 // tiles a rank may own
    std::vector<uint32_t> order;
    order.reserve(tiles);
    if (tilesX >= tilesY) gilbert(order, tilesX, 0, 0, (int)tilesX, 0, 0, (int)tilesY);
    else                  gilbert(order, tilesX, 0, 0, 0, (int)tilesY, (int)tilesX, 0);
    if (order.size() != tiles) return CHORDVIS_E_INVALID;                  // (cannot happen: the curve visits every cell once)

    uint64_t W = 0;
    if (loads) for (uint32_t t = 0; t < tiles; t++) W += loads[t];
    if (!loads || W == 0) {
        // runs of equal length (tiles / N, the first tiles % N ranks one more): compact regions of equal area
        for (uint32_t i = 0; i < tiles; i++) owners[order[i]] = (uint8_t)std::min<uint64_t>(N - 1u, (uint64_t)i * N / tiles);
        return CHORDVIS_OK;
    }

    std::vector<int> own(tiles, -1);
    std::vector<uint32_t> count(N, 0);
    std::vector<uint64_t> load(N, 0);
    auto give = [&](uint32_t t, uint32_t r) { own[t] = (int)r; count[r]++; load[r] += loads[t]; };

    // ---- heavy tiles, heaviest first, each to the least loaded rank
    std::vector<uint32_t> heavy;
    for (uint32_t t = 0; t < tiles; t++) if ((uint64_t)loads[t] * 4u * N > W) heavy.push_back(t);
    std::stable_sort(heavy.begin(), heavy.end(), [&](uint32_t a, uint32_t b) { return loads[a] > loads[b]; });   // (stable: ties by tile index)
    for (uint32_t t : heavy) {
        uint32_t best = N;
        for (uint32_t r = 0; r < N; r++) if (count[r] < S && (best == N || load[r] < load[best])) best = r;
        if (best == N) return CHORDVIS_E_INVALID;                           // (cannot happen: fewer than `tiles` tiles are placed and N * S >= tiles)
        give(t, best);
    }

    // ---- medium tiles: contiguous runs along the curve, the heaviest rank as light as possible
    // fillers: the lightest tiles -- each at most a quarter of the mean tile -- that together hold at most a sixteenth of the load
    std::vector<uint8_t> isLight(tiles, 0);
    {
        std::vector<uint32_t> cand;
        for (uint32_t t : order) if (own[t] < 0 && (uint64_t)loads[t] * tiles * 4u <= W) cand.push_back(t);
        std::stable_sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) { return loads[a] < loads[b]; });   // (stable: ties in curve order)
        uint64_t cum = 0;
        for (uint32_t t : cand) { cum += loads[t]; if (cum * 16u > W) break; isLight[t] = 1; }
    }
    std::vector<uint32_t> medium, light;
    uint64_t Lrem = 0;
    for (uint32_t t : order) {
        if (own[t] >= 0) continue;
        if (isLight[t]) light.push_back(t); else { medium.push_back(t); Lrem += loads[t]; }
    }
    {
        // The smallest bound B such that, walking the curve, rank 0, 1, ... each take tiles while their load stays within B
        // and their tile count within the chunk, and the last rank ends the curve: bisection over B with that greedy walk as the
        // feasibility test -- the optimal contiguous partition under the count limit (a region of light tiles that fills its
        // chunk pushes load to ALL the others, not to the ranks that happen to follow it on the curve).
        auto walk = [&](uint64_t B, bool commit) -> bool {
            uint32_t k = 0;
            uint64_t acc = 0;
            uint32_t cnt = 0;
            for (uint32_t t : medium) {
                const uint64_t w = loads[t];
                while (k < N && (count[k] + cnt >= S || load[k] + acc + w > B)) {
                    if (commit) { /* (tiles were given as they were taken) */ }
                    k++; acc = 0; cnt = 0;
                }
                if (k == N) return false;
                if (commit) give(t, k); else { acc += w; cnt++; }
            }
            return true;
        };
        uint64_t lo = 0, hi = W;
        for (uint32_t r = 0; r < N; r++) hi = std::max(hi, load[r] + Lrem);
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2u;
            if (walk(mid, false)) hi = mid; else lo = mid + 1u;
        }
        // commit: `give` updates load[] / count[] itself, so the walk's own accumulators stay zero
        {
            uint32_t k = 0;
            for (uint32_t t : medium) {
                const uint64_t w = loads[t];
                while (k + 1u < N && (count[k] >= S || load[k] + w > lo)) k++;
                give(t, k);
            }
        }
    }

    // ---- fillers: every rank to its tile count (tiles / N, the first tiles % N one more), then whatever room is left.
    //      A filler goes to the rank that owns the tile next to it ON THE CURVE while that rank is below its count -- a sweep
    //      forwards (the tile before it), then one backwards (the tile after it), so a run of light tiles between two regions is
    //      shared by those two regions from both ends and the regions stay compact; what neither neighbour has room for (a sky
    //      larger than its neighbours' counts) is handed out in curve order, rank by rank, as before.
    {
        auto quota = [&](uint32_t r) { return tiles / N + (r < tiles % N ? 1u : 0u); };
        for (int dir = 0; dir < 2; dir++) {
            int prev = -1;
            for (uint32_t i = 0; i < tiles; i++) {
                const uint32_t t = order[dir == 0 ? i : tiles - 1u - i];
                if (own[t] < 0 && prev >= 0 && count[(uint32_t)prev] < quota((uint32_t)prev)) give(t, (uint32_t)prev);
                prev = own[t];
            }
        }
        std::vector<uint32_t> rest;
        for (uint32_t t : light) if (own[t] < 0) rest.push_back(t);
        size_t next = 0;
        for (int round = 0; round < 2 && next < rest.size(); round++)
            for (uint32_t r = 0; r < N && next < rest.size(); r++) {
                const uint32_t q = round == 0 ? quota(r) : S;
                while (count[r] < q && next < rest.size()) give(rest[next++], r);
            }
    }

    // ---- repair: a rank above its chunk (the last run of the medium pass, or heavy tiles on a frame of very few tiles) hands
    //      its lightest tiles to the least loaded rank with room
    for (uint32_t r = 0; r < N; r++) {
        while (count[r] > S) {
            uint32_t pick = tiles;
            for (uint32_t t = 0; t < tiles; t++) if (own[t] == (int)r && (pick == tiles || loads[t] < loads[pick])) pick = t;
            uint32_t to = N;
            for (uint32_t q = 0; q < N; q++) if (q != r && count[q] < S && (to == N || load[q] < load[to])) to = q;
            if (pick == tiles || to == N) return CHORDVIS_E_INVALID;
            count[r]--; load[r] -= loads[pick]; give(pick, to);
        }
    }
    for (uint32_t t = 0; t < tiles; t++) { if (own[t] < 0) return CHORDVIS_E_INVALID; owners[t] = (uint8_t)own[t]; }
    return CHORDVIS_OK;
}

} // namespace chord

extern "C" {

uint32_t chordvis_tile_count(uint32_t width, uint32_t height)
{
    return ((width + CHORD_TILE - 1u) >> CHORD_TILE_SHIFT) * ((height + CHORD_TILE - 1u) >> CHORD_TILE_SHIFT);
}

uint32_t chordvis_tile_slots_per_rank(uint32_t width, uint32_t height, uint32_t ranks)
{
    const uint32_t tiles = chordvis_tile_count(width, height);
    return ranks ? (tiles + ranks - 1u) / ranks : tiles;
}

uint32_t chordvis_tile_slot_capacity(uint32_t width, uint32_t height, uint32_t ranks)
{
    const uint32_t tiles = chordvis_tile_count(width, height);
    if (ranks <= 1) return tiles;
    return (uint32_t)(((uint64_t)tiles * (1000u + CHORD_TILE_SLACK_PERMILLE) + (uint64_t)ranks * 1000u - 1u) / ((uint64_t)ranks * 1000u));
}

int chordvis_tile_layout(uint32_t width, uint32_t height, uint32_t ranks, const uint32_t* loads, uint32_t maxTilesPerRank, uint8_t* ownersOut)
{
    if (width == 0 || height == 0 || width > 4096u || height > 4096u) return CHORDVIS_E_INVALID;
    return chord::tile_layout((width + CHORD_TILE - 1u) >> CHORD_TILE_SHIFT, (height + CHORD_TILE - 1u) >> CHORD_TILE_SHIFT, ranks, loads, maxTilesPerRank, ownersOut);
}

} // extern "C"
