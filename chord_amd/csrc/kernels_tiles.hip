// chordvis — consumers' first step on the visibility buffer: the tile marker and the shading tile lists
// (SURVEY 8f-1).  Reference: install/resource/shader/visibility_tile.hlsl:39-219 (tilerMarkerCS,
// tilePrepareCS, prepareTileParamCS) recorded by source/renderer/visibility_tile.cpp:20-110.
//
//   visibility_mark_kernel   a wave per 128 x 8 pixels: 16-byte loads, 1 KB contiguous per wave instruction ->
//                            shading type of each pixel (visibility id -> draw command -> object ->
//                            material type, the last two folded into a per-object table at upload) ->
//                            128-bit mask per lane; the 4 lanes of a tile meet through two xor-shuffles.
//                            HBM-bound: reads every visibility word once (8 W H bytes; the reference's
//                            R32_UINT target would be 4 W H).
//   shading_tiles_kernel     a thread per 4 marker texels, block-wide scan, one reservation per 1024 texels;
//                            then the dispatch argument.
//
// The reference walks a 32x32-pixel region per 64-thread group through an 8x8 quad swizzle and reduces
// 4x4-pixel partial masks through groupshared memory in three barrier steps; none of that shapes the
// result (an OR over the tile), so the layout here is chosen for coalescing on 64-wide waves instead.

#include "device_layer.h"
#include "device_math.h"

namespace chord {

// A wave covers 128 x 8 pixels = 16 tiles: per pixel row every lane loads the two words at x = 2 lane (16 bytes,
// 1 KB contiguous per wave instruction), accumulates the types of its 2-pixel column strip over the 8 rows, and
// the four lanes of a tile meet through two xor-shuffles.
__global__ __launch_bounds__(256) void visibility_mark_kernel(const unsigned long long* __restrict__ vis, uint32_t W, uint32_t H,
                                                              const ChordDrawCmd* __restrict__ cmds, const uint32_t* __restrict__ cmdCount,
                                                              const DObjStatic* __restrict__ objStatic, uint4* __restrict__ marker,
                                                              uint32_t mW, uint32_t mH)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t groupsX = (mW + 15u) / 16u;
    const uint32_t waves = gridDim.x * 4u, total = groupsX * mH;
    const uint32_t n = *cmdCount;
    for (uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6); g < total; g += waves) {
        const uint32_t my = g / groupsX, mx = (g % groupsX) * 16u + (lane >> 2);
        const uint32_t x = (g % groupsX) * 128u + 2u * lane;
        uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        uint32_t lastId = 0xFFFFFFFFu, lastType = 0;
        auto add = [&](uint32_t pack) {
            uint32_t type;
            if ((pack >> 8) == lastId) type = lastType;                // same cluster as the pixel before
            else {
                type = 0u;                                             // kLightingType_None
                if (pack != 0u) {
                    const uint32_t instanceId = ((pack >> 8) & CHORD_MAX_INSTANCE_ID) - 1u;    // base.hlsli:443-447
                    if (instanceId < n) type = objStatic[cmds[instanceId].objectId].shadingType;   // visibility_tile.hlsl:51-60
                }
                lastId = pack >> 8; lastType = type;
            }
            const uint32_t bit = 1u << (type & 31u);
            const uint32_t w = (type >> 5) & 3u;
            m0 |= w == 0u ? bit : 0u; m1 |= w == 1u ? bit : 0u; m2 |= w == 2u ? bit : 0u; m3 |= w == 3u ? bit : 0u;
        };
        // pixels past the right / bottom edge: the reference's clamp-to-edge Gather repeats edge pixels of this same
        // tile, which adds nothing to an OR (visibility_tile.hlsl:83-87)
        if (x < W) {
            const bool pair = x + 1u < W && (((size_t)W & 1u) == 0u);   // 16-byte aligned pair loads need an even row pitch
#pragma unroll
            for (uint32_t r = 0; r < 8u; r++) {
                const uint32_t y = my * 8u + r;
                if (y >= H) break;
                const unsigned long long* src = vis + (size_t)y * W + x;
                if (pair) {
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(src);
                    add((uint32_t)v.x); add((uint32_t)v.y);
                } else {
                    add((uint32_t)src[0]);
                    if (x + 1u < W) add((uint32_t)src[1]);
                }
            }
        }
#pragma unroll
        for (int d = 1; d < 4; d <<= 1) {
            m0 |= (uint32_t)__shfl_xor((int)m0, d, 64); m1 |= (uint32_t)__shfl_xor((int)m1, d, 64);
            m2 |= (uint32_t)__shfl_xor((int)m2, d, 64); m3 |= (uint32_t)__shfl_xor((int)m3, d, 64);
        }
        if ((lane & 3u) == 0u && mx < mW) marker[(size_t)my * mW + mx] = make_uint4(m0, m1, m2, m3);
    }
}

// One thread per 4 consecutive marker texels, block-wide exclusive scan of the per-thread counts, ONE reservation
// per 1024 texels: the reference's one InterlockedAdd per wave (visibility_tile.hlsl:184-190) on a single word would
// be 2 025 returning atomics at 4K = 23 us here (~88/us per address, measured).
__global__ __launch_bounds__(256) void shading_tiles_kernel(const uint4* __restrict__ marker, uint32_t mW, uint32_t mH,
                                                            uint32_t index, uint32_t bit, uint2* __restrict__ tiles, uint32_t* __restrict__ count)
{
    __shared__ uint32_t sWave[4], sBase;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, total = mW * mH;
    for (uint32_t base = blockIdx.x * 1024u; base < total; base += gridDim.x * 1024u) {
        const uint32_t t0 = base + threadIdx.x * 4u;
        uint32_t flags = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4u; i++) {
            if (t0 + i < total) {
                const uint4 m = marker[t0 + i];
                const uint32_t word = index == 0u ? m.x : index == 1u ? m.y : index == 2u ? m.z : m.w;
                if (word & bit) flags |= 1u << i;                      // visibility_tile.hlsl:169-170
            }
        }
        const uint32_t mine = (uint32_t)__popc(flags);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t nb = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= (uint32_t)d) incl += nb; }
        if (lane == 63u) sWave[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4u; w++) { const uint32_t c = sWave[w]; if (w < wave) before += c; all += c; }
        if (threadIdx.x == 0) sBase = all ? atomicAdd(count, all) : 0u;
        __syncthreads();
        uint32_t slot = sBase + before + incl - mine;
#pragma unroll
        for (uint32_t i = 0; i < 4u; i++)
            if (flags & (1u << i)) { const uint32_t t = t0 + i; tiles[slot++] = make_uint2((t % mW) * 8u, (t / mW) * 8u); }   // :174,204
        __syncthreads();                                               // sWave / sBase are rewritten by the next chunk
    }
}

__global__ void shading_tile_args_kernel(const uint32_t* __restrict__ count, uint4* __restrict__ args)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *args = make_uint4((*count + 3u) / 4u, 1u, 1u, 1u);   // :208-219
}

void launch_visibility_mark(ChordCtx* c, const unsigned long long* vis, const ChordDrawCmd* cmds, const uint32_t* cmdCount, uint32_t* marker)
{
    const uint32_t mW = (c->width + 7u) / 8u, mH = (c->height + 7u) / 8u;
    const uint32_t groups = ((mW + 15u) / 16u) * mH;
    uint32_t blocks = (groups + 3u) / 4u;
    const uint32_t maxBlocks = (uint32_t)c->numCUs * 8u;
    if (blocks > maxBlocks) blocks = maxBlocks;
    if (blocks < 1u) blocks = 1u;
    CHORD_LAUNCH(c, visibility_mark_kernel, dim3(blocks), dim3(256), 0, c->stream, vis, c->width, c->height, cmds, cmdCount,
                       c->dObjStatic, reinterpret_cast<uint4*>(marker), mW, mH);
}

void launch_shading_tiles(ChordCtx* c, const uint32_t* marker, uint32_t shadingType, uint32_t* tiles, uint32_t* count, uint32_t* args)
{
    const uint32_t mW = (c->width + 7u) / 8u, mH = (c->height + 7u) / 8u, total = mW * mH;
    (void)hipMemsetAsync(count, 0, sizeof(uint32_t), c->stream);        // queue.clearUAV(countBuffer), visibility_tile.cpp:68
    uint32_t blocks = (total + 1023u) / 1024u;
    if (blocks > (uint32_t)c->numCUs * 4u) blocks = (uint32_t)c->numCUs * 4u;
    if (blocks < 1u) blocks = 1u;
    CHORD_LAUNCH(c, shading_tiles_kernel, dim3(blocks), dim3(256), 0, c->stream, reinterpret_cast<const uint4*>(marker), mW, mH,
                       (shadingType >> 5) & 3u, 1u << (shadingType & 31u), reinterpret_cast<uint2*>(tiles), count);
    CHORD_LAUNCH(c, shading_tile_args_kernel, dim3(1), dim3(64), 0, c->stream, count, reinterpret_cast<uint4*>(args));
}

} // namespace chord
