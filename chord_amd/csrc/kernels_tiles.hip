// chordvis — consumers' first step on the visibility buffer: the tile marker and the shading tile lists
// (SURVEY 8f-1).  Reference: install/resource/shader/visibility_tile.hlsl:39-219 (tilerMarkerCS,
// tilePrepareCS, prepareTileParamCS) recorded by source/renderer/visibility_tile.cpp:20-110.
//
//   visibility_mark_kernel   one lane per (8x8 tile, pixel row): a 64-byte line of visibility words ->
//                            shading type of each pixel (visibility id -> draw command -> object ->
//                            material type, the last two folded into a per-object table at upload) ->
//                            128-bit mask; the 8 rows of a tile meet through three xor-shuffles.
//                            HBM-bound: reads every visibility word once (8 W H bytes; the reference's
//                            R32_UINT target would be 4 W H).
//   shading_tiles_kernel     one lane per marker texel, wave64 ballot compaction, one atomic per wave
//                            (same structure as visibility_tile.hlsl:184-193); then the dispatch argument.
//
// The reference walks a 32x32-pixel region per 64-thread group through an 8x8 quad swizzle and reduces
// 4x4-pixel partial masks through groupshared memory in three barrier steps; none of that shapes the
// result (an OR over the tile), so the layout here is chosen for coalescing on 64-wide waves instead.

#include "device_layer.h"
#include "device_math.h"

namespace chord {

__global__ __launch_bounds__(256) void visibility_mark_kernel(const unsigned long long* __restrict__ vis, uint32_t W, uint32_t H,
                                                              const ChordDrawCmd* __restrict__ cmds, const uint32_t* __restrict__ cmdCount,
                                                              const DObjStatic* __restrict__ objStatic, uint4* __restrict__ marker,
                                                              uint32_t mW, uint32_t mH)
{
    const uint32_t lane = threadIdx.x & 63u, row = lane & 7u, tcol = lane >> 3;
    const uint32_t groupsX = (mW + 7u) / 8u;                          // a wave covers 8 tiles of one marker row
    const uint32_t waves = gridDim.x * 4u, total = groupsX * mH;
    const uint32_t n = *cmdCount;
    for (uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6); g < total; g += waves) {
        const uint32_t my = g / groupsX, mx = (g % groupsX) * 8u + tcol;
        const uint32_t y = my * 8u + row, x0 = mx * 8u;
        uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        if (mx < mW && y < H) {
            // pixels past the right / bottom edge: the reference's clamp-to-edge Gather repeats edge pixels of this
            // same tile, which adds nothing to an OR (visibility_tile.hlsl:83-87)
            const unsigned long long* src = vis + (size_t)y * W + x0;
            uint32_t lastPack = 0xFFFFFFFFu, lastType = 0;
#pragma unroll
            for (uint32_t i = 0; i < 8u; i++) {
                if (x0 + i >= W) break;
                const uint32_t pack = (uint32_t)src[i];                // the R32_UINT visibility texel
                uint32_t type;
                if ((pack >> 8) == (lastPack >> 8)) type = lastType;   // same cluster as the pixel before
                else {
                    type = 0u;                                         // kLightingType_None
                    if (pack != 0u) {
                        const uint32_t instanceId = ((pack >> 8) & CHORD_MAX_INSTANCE_ID) - 1u;    // base.hlsli:443-447
                        if (instanceId < n) type = objStatic[cmds[instanceId].objectId].shadingType;   // :51-60
                    }
                    lastPack = pack; lastType = type;
                }
                const uint32_t bit = 1u << (type & 31u);
                switch ((type >> 5) & 3u) { case 0: m0 |= bit; break; case 1: m1 |= bit; break; case 2: m2 |= bit; break; default: m3 |= bit; }
            }
        }
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
            m0 |= (uint32_t)__shfl_xor((int)m0, d, 64); m1 |= (uint32_t)__shfl_xor((int)m1, d, 64);
            m2 |= (uint32_t)__shfl_xor((int)m2, d, 64); m3 |= (uint32_t)__shfl_xor((int)m3, d, 64);
        }
        if (row == 0u && mx < mW) marker[(size_t)my * mW + mx] = make_uint4(m0, m1, m2, m3);
    }
}

__global__ __launch_bounds__(256) void shading_tiles_kernel(const uint4* __restrict__ marker, uint32_t mW, uint32_t mH,
                                                            uint32_t index, uint32_t bit, uint2* __restrict__ tiles, uint32_t* __restrict__ count)
{
    const uint32_t lane = threadIdx.x & 63u, total = mW * mH;
    for (uint32_t base = blockIdx.x * 256u; base < total; base += gridDim.x * 256u) {
        const uint32_t t = base + threadIdx.x;
        bool has = false;
        if (t < total) {
            const uint4 m = marker[t];
            const uint32_t word = index == 0u ? m.x : index == 1u ? m.y : index == 2u ? m.z : m.w;
            has = (word & bit) != 0u;                                  // visibility_tile.hlsl:169
        }
        const unsigned long long mask = __ballot(has);
        if (mask == 0ull) continue;
        uint32_t slot = 0;
        if (lane == 0u) slot = atomicAdd(count, (uint32_t)__popcll(mask));   // :184-190
        slot = (uint32_t)__shfl((int)slot, 0, 64) + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (has) tiles[slot] = make_uint2((t % mW) * 8u, (t / mW) * 8u);     // :174,204
    }
}

__global__ void shading_tile_args_kernel(const uint32_t* __restrict__ count, uint4* __restrict__ args)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *args = make_uint4((*count + 3u) / 4u, 1u, 1u, 1u);   // :208-219
}

void launch_visibility_mark(ChordCtx* c, const unsigned long long* vis, const ChordDrawCmd* cmds, const uint32_t* cmdCount, uint32_t* marker)
{
    const uint32_t mW = (c->width + 7u) / 8u, mH = (c->height + 7u) / 8u;
    const uint32_t groups = ((mW + 7u) / 8u) * mH;
    uint32_t blocks = (groups + 3u) / 4u;
    const uint32_t maxBlocks = (uint32_t)c->numCUs * 8u;
    if (blocks > maxBlocks) blocks = maxBlocks;
    if (blocks < 1u) blocks = 1u;
    hipLaunchKernelGGL(visibility_mark_kernel, dim3(blocks), dim3(256), 0, c->stream, vis, c->width, c->height, cmds, cmdCount,
                       c->dObjStatic, reinterpret_cast<uint4*>(marker), mW, mH);
}

void launch_shading_tiles(ChordCtx* c, const uint32_t* marker, uint32_t shadingType, uint32_t* tiles, uint32_t* count, uint32_t* args)
{
    const uint32_t mW = (c->width + 7u) / 8u, mH = (c->height + 7u) / 8u, total = mW * mH;
    (void)hipMemsetAsync(count, 0, sizeof(uint32_t), c->stream);        // queue.clearUAV(countBuffer), visibility_tile.cpp:68
    uint32_t blocks = (total + 255u) / 256u;
    if (blocks > (uint32_t)c->numCUs * 4u) blocks = (uint32_t)c->numCUs * 4u;
    if (blocks < 1u) blocks = 1u;
    hipLaunchKernelGGL(shading_tiles_kernel, dim3(blocks), dim3(256), 0, c->stream, reinterpret_cast<const uint4*>(marker), mW, mH,
                       (shadingType >> 5) & 3u, 1u << (shadingType & 31u), reinterpret_cast<uint2*>(tiles), count);
    hipLaunchKernelGGL(shading_tile_args_kernel, dim3(1), dim3(64), 0, c->stream, count, reinterpret_cast<uint4*>(args));
}

} // namespace chord
