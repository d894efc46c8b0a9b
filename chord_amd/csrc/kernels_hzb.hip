// HZB build + de-stripe kernels, hand-written for gfx950.
//
// Replaces the single-dispatch HZB shaders (hzb_one.hlsl:126-372, hzb.hlsl:127-389) driven by
// buildHZB (hzb.cpp:38-227).  The reference keeps everything in one dispatch with a
// globallycoherent mip 5 + atomic ticket so the last workgroup can finish mips 6..11; on MI355X
// an agent-scope release/acquire pair costs more (~3.5 us) than a kernel boundary (~1.5 us), so
// the chain is built by three stream-ordered launches instead:
//   hzb_mip0_kernel   depth (high half of the visibility words) -> mip 0 min[/max] (+ valid range)
//                     HBM-bound: reads 8*W*H bytes once, 16 B per lane, fully coalesced
//   hzb_mips_kernel   one block per 32x32 mip-0 tile -> mips 1..5 through LDS
//   hzb_tail_kernel   one block -> mips 6..n
// Texel semantics: mip l texel = min (max) over the 2^(l+1) square of
// edge-clamped source depth, stored as binary16 (RNE); max chain carries +1 ulp from mip 5 up
// (hzb.hlsl:67-71).  Only the sampled extent of each mip (x <= ((W-1)>>1)>>l) is defined.
//
// Sharded (multi-GPU) frames do not come here at all: the tile kernel reduces every owned tile to its HZB texels (mips 0..5)
// into per-tile slots of an exchange buffer, the slots are all-gathered, and hzb_untile_kernel copies them to the chain
// (DESIGN.md 6).  chordvis_build_hzb on a sharded context reads the resolved (row-major) image.

#include "hzb_device.h"

#include <algorithm>

namespace chord {

// full frame, row-major visibility words -> mip 0 (min, optional max, optional range)
__global__ __launch_bounds__(256) void hzb_mip0_kernel(HzbParams p, int wantMax, int wantRange)
{
    const uint32_t vw = valid_w(p.desc, 0), vh = valid_h(p.desc, 0);
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t y = blockIdx.y * 4u + (threadIdx.x >> 6);
    float mn = 0.0f, mx = 0.0f;
    uint32_t rmin = 0xFFFFFFFFu, rmax = 0u;
    const bool act = x < vw && y < vh;
    if (act) {
        const uint32_t sx0 = min(2u * x, (uint32_t)p.W - 1u), sx1 = min(2u * x + 1u, (uint32_t)p.W - 1u);
        const uint32_t sy0 = min(2u * y, (uint32_t)p.H - 1u), sy1 = min(2u * y + 1u, (uint32_t)p.H - 1u);
        const size_t r0 = (size_t)sy0 * (uint32_t)p.W, r1 = (size_t)sy1 * (uint32_t)p.W;
        const float d00 = __uint_as_float((uint32_t)(p.vis[r0 + sx0] >> 32));
        const float d10 = __uint_as_float((uint32_t)(p.vis[r0 + sx1] >> 32));
        const float d01 = __uint_as_float((uint32_t)(p.vis[r1 + sx0] >> 32));
        const float d11 = __uint_as_float((uint32_t)(p.vis[r1 + sx1] >> 32));
        mn = fminf(fminf(fminf(d00, d10), d01), d11);
        mx = fmaxf(fmaxf(fmaxf(d00, d10), d01), d11);
        {
            const uint32_t mw = max(1u, p.desc.width);
            p.hzbMin[p.desc.mipOffset[0] + y * mw + x] = f32_to_f16(mn);
            if (wantMax) p.hzbMax[p.desc.mipOffset[0] + y * mw + x] = f32_to_f16(mx);
        }
        if (wantRange) {                                          // hzb.hlsl:163-176
            const float dd[4] = {d00, d10, d01, d11};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (dd[i] > 0.0f) {
                    const uint32_t b = __float_as_uint(dd[i]);
                    if (dd[i] < 1.0f) rmin = min(rmin, b);
                    rmax = max(rmax, b);
                }
            }
        }
    }
    if (wantRange) {
        // The reference does one InterlockedMin/Max per wave on a single word (hzb.hlsl:168-176); on
        // MI355X one word sustains only ~88 atomics/us (32k waves at 4K = 0.7 ms), so each block
        // writes one partial instead and the single-block tail kernel reduces them.
        __shared__ uint32_t sMn[4], sMx[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            rmin = min(rmin, (uint32_t)__shfl_down(rmin, off, 64));
            rmax = max(rmax, (uint32_t)__shfl_down(rmax, off, 64));
        }
        if ((threadIdx.x & 63u) == 0u) { sMn[threadIdx.x >> 6] = rmin; sMx[threadIdx.x >> 6] = rmax; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t b = blockIdx.y * gridDim.x + blockIdx.x;
            p.rangePartials[2u * b] = min(min(sMn[0], sMn[1]), min(sMn[2], sMn[3]));
            p.rangePartials[2u * b + 1u] = max(max(sMx[0], sMx[1]), max(sMx[2], sMx[3]));
        }
    }
}

// One block per 32x32 mip-0 tile: mips 1..5.
__global__ __launch_bounds__(256) void hzb_mips_kernel(HzbParams p, int wantMax)
{
    __shared__ float sMin[16][17], sMax[16][17];
    const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
    const uint32_t bx = blockIdx.x, by = blockIdx.y;
    const ChordHZBDesc& d = p.desc;

    // level 1 from level 0
    float mn = 0.0f, mx = 0.0f;
    {
        const uint32_t pw = valid_w(d, 0), ph = valid_h(d, 0), pmw = max(1u, d.width);
        const uint32_t X = bx * 16u + tx, Y = by * 16u + ty;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const uint32_t cx = min(2u * X + i, pw - 1u), cy = min(2u * Y + j, ph - 1u);
                const float a = f16_to_f32(p.hzbMin[d.mipOffset[0] + cy * pmw + cx]);
                const float b = wantMax ? f16_to_f32(p.hzbMax[d.mipOffset[0] + cy * pmw + cx]) : 0.0f;
                if (i == 0 && j == 0) { mn = a; mx = b; } else { mn = fminf(mn, a); mx = fmaxf(mx, b); }
            }
        if (d.mipCount > 1 && X < valid_w(d, 1) && Y < valid_h(d, 1)) {
            const uint32_t mw = max(1u, d.width >> 1);
            p.hzbMin[d.mipOffset[1] + Y * mw + X] = f32_to_f16(mn);
            if (wantMax) p.hzbMax[d.mipOffset[1] + Y * mw + X] = f32_to_f16(mx);
        }
        sMin[ty][tx] = mn; sMax[ty][tx] = mx;
    }
    // levels 2..5 through LDS: level l tile is (32 >> l) wide
#pragma unroll
    for (uint32_t l = 2; l <= 5; l++) {
        __syncthreads();
        const uint32_t side = 32u >> l;                     // 8, 4, 2, 1
        const bool act = tx < side && ty < side && l < d.mipCount;
        float rmn = 0.0f, rmx = 0.0f;
        if (act) {
            const uint32_t pw = valid_w(d, l - 1), ph = valid_h(d, l - 1);
            const uint32_t originX = bx * (64u >> l), originY = by * (64u >> l);     // tile origin at level l-1
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    // children clamped to the valid extent of level l-1, in tile-local coordinates
                    const uint32_t gx = min(originX + 2u * tx + i, pw - 1u), gy = min(originY + 2u * ty + j, ph - 1u);
                    const uint32_t lx = gx >= originX ? gx - originX : 0u, ly = gy >= originY ? gy - originY : 0u;
                    const float a = sMin[ly][lx], b = sMax[ly][lx];
                    if (i == 0 && j == 0) { rmn = a; rmx = b; } else { rmn = fminf(rmn, a); rmx = fmaxf(rmx, b); }
                }
        }
        __syncthreads();
        if (act) {
            const uint32_t X = bx * side + tx, Y = by * side + ty;
            // keep what the stored halves hold (the max chain's +1 ulp at mip 5 is applied on store)
            uint16_t hmn = f32_to_f16(rmn), hmx = f32_to_f16(rmx);
            if (l == 5) hmx = (uint16_t)(hmx + 1u);                                  // storeHZBMip5
            if (X < valid_w(d, l) && Y < valid_h(d, l)) {
                const uint32_t mw = max(1u, d.width >> l);
                p.hzbMin[d.mipOffset[l] + Y * mw + X] = hmn;
                if (wantMax) p.hzbMax[d.mipOffset[l] + Y * mw + X] = hmx;
            }
            sMin[ty][tx] = rmn; sMax[ty][tx] = rmx;
        }
    }
}

// One block: mips 6..mipCount-1 from the stored mip 5, and the valid-range reduction (hzb_device.h).
__global__ __launch_bounds__(256) void hzb_tail_kernel(HzbParams p, int wantMax, int wantRange, uint32_t firstLevel)
{
    hzb_tail_block(p, wantMax, wantRange, firstLevel);
}

// rank-major tile slots (64 rows of 64 words per tile) -> row-major: one block per tile, 16-byte loads and stores
__global__ __launch_bounds__(256) void detile_kernel(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst,
                                                     uint32_t W, uint32_t H, ShardInfo shard)
{
    const uint32_t tile = blockIdx.y * shard.tilesX + blockIdx.x;
    const ulonglong2* __restrict__ s2 = reinterpret_cast<const ulonglong2*>(src + (size_t)shard.tileSlot[tile] * (CHORD_TILE * CHORD_TILE));
    const uint32_t ox = blockIdx.x * CHORD_TILE, oy = blockIdx.y * CHORD_TILE;
    for (uint32_t i = threadIdx.x; i < CHORD_TILE * CHORD_TILE / 2u; i += 256u) {
        const uint32_t ly = i >> (CHORD_TILE_SHIFT - 1), lx = (i & (CHORD_TILE / 2u - 1u)) * 2u;
        if (oy + ly >= H || ox + lx >= W) continue;
        const ulonglong2 v = s2[i];
        unsigned long long* d = dst + (size_t)(oy + ly) * W + ox + lx;
        if (ox + lx + 1u < W) *reinterpret_cast<ulonglong2*>(d) = v;       // (W even or not: 16-byte aligned iff the row start is; checked by the launcher)
        else d[0] = v.x;
    }
}
__global__ __launch_bounds__(256) void detile_scalar_kernel(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst,
                                                            uint32_t W, uint32_t H, ShardInfo shard)
{
    const uint32_t tile = blockIdx.y * shard.tilesX + blockIdx.x;
    const unsigned long long* __restrict__ s1 = src + (size_t)shard.tileSlot[tile] * (CHORD_TILE * CHORD_TILE);
    const uint32_t ox = blockIdx.x * CHORD_TILE, oy = blockIdx.y * CHORD_TILE;
    for (uint32_t i = threadIdx.x; i < CHORD_TILE * CHORD_TILE; i += 256u) {
        const uint32_t ly = i >> CHORD_TILE_SHIFT, lx = i & (CHORD_TILE - 1u);
        if (oy + ly < H && ox + lx < W) dst[(size_t)(oy + ly) * W + ox + lx] = s1[i];
    }
}

// Exchanged tile slots -> the chain.  The tile kernel of a sharded frame reduces every tile it owns to the HZB texels the tile
// covers (mips 0..5; tile_out_and_hzb in kernels_raster.hip) and stores them in the tile's slot of an exchange buffer; after the
// all-gather every rank copies all slots -- its own included -- to their places in the chain.  One block per tile.
// FINAL: the end-of-frame slots (min | max | valid range | bin entries): both chains, the per-tile range partials the tail
// reduces, and every tile's load for the next re-balancing of the tile map.
template <bool FINAL>
__global__ __launch_bounds__(256) void hzb_untile_kernel(const uint16_t* __restrict__ exchange, ShardInfo shard, ChordHZBDesc d,
                                                         uint16_t* __restrict__ hzbMin, uint16_t* __restrict__ hzbMax,
                                                         uint32_t* __restrict__ tileRange, uint32_t* __restrict__ tileLoads)
{
    const uint32_t tX = blockIdx.x, tY = blockIdx.y, tile = tY * shard.tilesX + tX;
    const uint16_t* __restrict__ slot = exchange + (size_t)shard.tileSlot[tile] * (FINAL ? CHORD_HZB_FINAL_SLOT_HALVES : CHORD_HZB_SLOT_HALVES);
    for (uint32_t l = 0; l < 6u && l < d.mipCount; l++) {
        const uint32_t side = 32u >> l, off = hzb_slot_level_offset(l);
        const uint32_t vw = valid_w(d, l), vh = valid_h(d, l), mw = max(1u, d.width >> l);
        for (uint32_t i = threadIdx.x; i < side * side; i += 256u) {
            const uint32_t lx = i & (side - 1u), ly = i >> (5u - l);
            const uint32_t gx = tX * side + lx, gy = tY * side + ly;
            if (gx >= vw || gy >= vh) continue;
            const size_t o = d.mipOffset[l] + (size_t)gy * mw + gx;
            hzbMin[o] = slot[off + i];
            if (FINAL) hzbMax[o] = slot[CHORD_HZB_FINAL_MAX_OFFSET + off + i];
        }
    }
    if (FINAL && threadIdx.x == 0u) {
        const uint32_t* __restrict__ tail = reinterpret_cast<const uint32_t*>(slot + CHORD_HZB_FINAL_RANGE_OFFSET);
        tileRange[2u * tile] = tail[0]; tileRange[2u * tile + 1u] = tail[1];
        tileLoads[tile] = tail[2];
    }
}

static HzbParams make_params(ChordCtx* c, HzbBuffers& out)
{
    HzbParams p;
    p.vis = (const unsigned long long*)(c->shard.ranks > 1 ? c->dVisResolved : c->dVis); p.W = (int32_t)c->width; p.H = (int32_t)c->height;
    p.desc = out.desc;
    p.hzbMin = out.minTexels; p.hzbMax = out.maxTexels; p.validRange = out.validRange;
    p.rangePartials = c->dRangePartials; p.rangePartialCount = 0;
    return p;
}

void launch_hzb_build(ChordCtx* c, HzbBuffers& out, bool bMin, bool bMax, bool bValidRange)
{
    (void)bMin;
    HzbParams p = make_params(c, out);
    const int wantMax = bMax ? 1 : 0, wantRange = bValidRange ? 1 : 0;
    const uint32_t vw = min(p.desc.width, (((uint32_t)p.W - 1u) >> 1) + 1u), vh = min(p.desc.height, (((uint32_t)p.H - 1u) >> 1) + 1u);
    const dim3 g0((vw + 63u) / 64u, (vh + 3u) / 4u);
    p.rangePartialCount = g0.x * g0.y;
    CHORD_LAUNCH(c, hzb_mip0_kernel, g0, dim3(256), 0, c->stream, p, wantMax, wantRange);
    const dim3 g1((vw + 31u) / 32u, (vh + 31u) / 32u);
    CHORD_LAUNCH(c, hzb_mips_kernel, g1, dim3(256), 0, c->stream, p, wantMax);
    if (p.desc.mipCount > 6 || wantRange) CHORD_LAUNCH(c, hzb_tail_kernel, dim3(1), dim3(256), 0, c->stream, p, wantMax, wantRange, 6u);
    out.valid = true;
}

// Sharded frames: the all-gathered tile slots of an exchange buffer -> mips 0..5 of `out` (the tail, mips 6.. and the valid
// range, is left to whoever needs the chain next: the phase-1 cull reduces it itself, the history chain's rides on the next
// frame's first kernel -- as in the single-GPU frame, whose tile kernel writes the same texels straight into the chain).
void launch_hzb_untile(ChordCtx* c, HzbBuffers& out, bool finalChain)
{
    const dim3 g(c->tilesX, c->tilesY);
    if (finalChain) CHORD_LAUNCH(c, hzb_untile_kernel<true>, g, dim3(256), 0, c->stream, (const uint16_t*)c->dHzbFinalExchange, c->shard, out.desc,
                                       out.minTexels, out.maxTexels, c->dTileRange, c->dTileLoads);
    else            CHORD_LAUNCH(c, hzb_untile_kernel<false>, g, dim3(256), 0, c->stream, (const uint16_t*)c->dHzbExchange, c->shard, out.desc,
                                       out.minTexels, (uint16_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr);
    out.valid = true;
}

HzbParams make_hzb_tail_params(ChordCtx* c, HzbBuffers& out)
{
    HzbParams p = make_params(c, out);
    p.rangePartials = c->dTileRange;
    p.rangePartialCount = c->tilesX * c->tilesY;
    return p;
}

// After a raster pass with the fused reduction (mips 0..5 written by raster_tile_kernel): only the tail is left.
void launch_hzb_tail(ChordCtx* c, HzbBuffers& out, bool bMax, bool bValidRange)
{
    HzbParams p = make_hzb_tail_params(c, out);
    if (p.desc.mipCount > (uint32_t)CHORD_TILE_SHIFT || bValidRange)
        CHORD_LAUNCH(c, hzb_tail_kernel, dim3(1), dim3(256), 0, c->stream, p, bMax ? 1 : 0, bValidRange ? 1 : 0, (uint32_t)CHORD_TILE_SHIFT);
    out.valid = true;
}

__global__ __launch_bounds__(256) void depth_extract_kernel(const unsigned long long* __restrict__ vis, float* __restrict__ depth, size_t words)
{
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < words; i += (size_t)gridDim.x * 256u)
        depth[i] = __uint_as_float((uint32_t)(vis[i] >> 32));
}
__global__ __launch_bounds__(256) void depth_expand_kernel(const float* __restrict__ depth, unsigned long long* __restrict__ vis, size_t words)
{
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < words; i += (size_t)gridDim.x * 256u)
        vis[i] = (unsigned long long)__float_as_uint(depth[i]) << 32;
}
void launch_depth_extract(ChordCtx* c, const unsigned long long* vis, float* depth, size_t words)
{
    const uint32_t blocks = (uint32_t)std::min<size_t>((words + 255u) / 256u, (size_t)c->numCUs * 16u);
    CHORD_LAUNCH(c, depth_extract_kernel, dim3(blocks ? blocks : 1u), dim3(256), 0, c->stream, vis, depth, words);
}
void launch_depth_expand(ChordCtx* c, const float* depth, unsigned long long* vis, size_t words)
{
    const uint32_t blocks = (uint32_t)std::min<size_t>((words + 255u) / 256u, (size_t)c->numCUs * 16u);
    CHORD_LAUNCH(c, depth_expand_kernel, dim3(blocks ? blocks : 1u), dim3(256), 0, c->stream, depth, vis, words);
}

void launch_detile(ChordCtx* c, hipStream_t stream)
{
    const dim3 g(c->tilesX, c->tilesY);
    // (16-byte stores need every row start 16-byte aligned: an even width)
    if ((c->width & 1u) == 0u) CHORD_LAUNCH(c, detile_kernel, g, dim3(256), 0, stream ? stream : c->stream, (const unsigned long long*)c->dVis,
                                                  (unsigned long long*)c->dVisResolved, c->width, c->height, c->shard);
    else CHORD_LAUNCH(c, detile_scalar_kernel, g, dim3(256), 0, stream ? stream : c->stream, (const unsigned long long*)c->dVis,
                            (unsigned long long*)c->dVisResolved, c->width, c->height, c->shard);
}

} // namespace chord
