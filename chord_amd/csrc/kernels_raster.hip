// Software rasterizer of the visibility path, hand-written for gfx950 (wave64, LDS tiles).
//
// Replaces meshRasterPassMS + the fixed-function rasterizer + depth test + meshRasterPassPS
// (mesh_raster.hlsl:51-210; state mesh_raster.cpp:141-156, helper.h:7-13,304-324,395-407).
//
// Measured on MI355X (tools/microbench/atomics.hip): a 64-bit global atomicMax costs one L2/fabric
// request per 64-byte line touched — 26 Gop/s when every lane hits its own line (small triangles),
// 214 Gop/s for an 8x8 footprint — while ds_max_u64 in LDS sustains ~1 050 Gop/s and plain coalesced
// stores ~6.6 TB/s.  So fragments are resolved in LDS and the visibility words leave the CU once:
//
//   raster_setup_kernel   one wave per visible meshlet, software-pipelined over the wave's meshlets: coalesced
//                         meshletData / position stream -> clip-space transform -> LDS (SoA x,y,w,u,v,depth) ->
//                         per-triangle culls (mesh_raster.hlsl:143-179) -> snapped setup -> a 32-byte (vertices
//                         at most 64 px apart) or 48-byte triangle record + one bin entry per 64x64 screen tile
//                         touched; every reservation of a meshlet in one memory round trip
//   raster_setup_blocks_kernel  dense launches (a cluster per 16 pixels or more: sub-pixel geometry) run this one first: a
//                         cluster that fits a 16x16-pixel window is resolved in LDS and leaves as a pixel block per
//                         touched tile (one bin entry each) instead of a record per triangle; the clusters it cannot
//                         take go to the record kernel through a leftover list.  <true>: bin slots of hot tiles drawn ahead
//   raster_clip_and_bin_large_kernel  one launch, two roles: (a) homogeneous Sutherland-Hodgman clipper for
//                         triangles touching the near / guard planes (rare; emits + bins its pieces itself),
//                         (b) records touching more than 2x2 tiles: one wave per record, one lane per tile
//   raster_tile_order_kernel  work items of the tile kernel (tiles, or slices of tiles with long bins), heaviest first
//   raster_tile_kernel    one 512-thread workgroup per work item: the tile's 4096 packed words live in LDS;
//                         every binned triangle is scan-converted with ds_max_u64 (tiny: one lane per triangle;
//                         others: cut into (triangle, row) units that a block-wide prefix sum deals out one row
//                         per lane, each row a branch-free loop over an fp32-bounded span).  Tile-out: 16-byte
//                         coalesced stores (on the first pass of a frame this is also the clear) fused with the
//                         reduction of the tile to HZB mips 0..5; slices of a split tile meet in a per-tile slab.
// Bins hold 16 384 entries per tile and continue in 1024-entry pool chunks (bin_store).
// The packed word is (asuint(depth) << 32) | ((slot+1)&0xFFFFFF)<<8 | tri: reverse-Z "greater wins"
// and the id in one 64-bit max (GREATER_OR_EQUAL depth test + id write of the reference).
//
// Arithmetic is the canonical restatement of SURVEY.md §8c: snapped 24.8 coordinates, pixel
// centres at +0.5, top-left rule, integer edge functions, depth = (d0 + l1*(d1-d0)) + l2*(d2-d0).
// Integer / fp32 VALU + LDS atomics; no MFMA.  Built with -ffp-contract=off.

#include "device_layer.h"
#include "device_math.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace chord {

#define GUARD_BAND 1024.0f
#define LDS_VERTS 256
#define TILE CHORD_TILE         // pixels per tile side
#define TILE_SHIFT CHORD_TILE_SHIFT
// LDS row pitch of the tile in 8-byte words: one word of padding rotates the banks by 2 per row, so lanes
// that walk different rows at the same x (the row-unit loop) do not all hit the same bank pair
#define TPITCH (TILE + 1)
#define SMALL_AREA 256          // clipped bbox pixels a single lane scans on its own
#define TC_STRIDE CHORD_TILECOUNT_STRIDE   // one bin counter per 64-byte line (no false sharing between tiles)
// words of a tile's counter line: 0 = bin entries, 1 = of which pixel blocks (words 0 and 1 are ONE 64-bit counter for the
// block path: a block's bin slot and the block count travel in a single atomic), 2 = ticket of the slices of a split tile
#define TC_BLOCKS 1u
#define TC_TICKET 2u
// Where alpha-tested triangles are scan-converted: 0 = in raster_masked_tile_kernel, a pass of their own in front of the tile kernel
// (whose only instantiations are then the opaque ones); 1 = inside the tile kernel's MASKED instantiations, in a pass of their own
// over the batch's units (the form of rounds 2-4, kept as a build switch for the A/B: profiles/r05_masked_variants.txt)
#ifndef CHORD_MASKED_FUSED
#define CHORD_MASKED_FUSED 1
#endif
#define TC_MASKED 3u            // 3 = the bin holds entries of alpha-tested triangles (a flag, plain stores: raster_masked_tile_kernel's work list)

struct RasterParams {
    const uint32_t* count; const ChordDrawCmd* cmds;
    uint32_t cmdCap;                                    // entries the list `cmds` has room for (>= 1): a wave may read its first command before it knows the count
    uint32_t* leftCount; ChordDrawCmd* leftCmds;         // clusters of a dense launch that do not become pixel blocks (raster_setup_blocks_kernel -> raster_setup_kernel), or NULL
    const DObjFrame* objFrame; const DObjStatic* objStatic;
    const DMeshlet* meshlets; const uint32_t* meshletData; const float* positions;
    const DMaterial* materials; const uint8_t* texAlpha; const float* texcoords;   // masked materials (texcoords may be null: uv = 0)
    unsigned long long* vis;
    float* depthOut;                                    // depth-only views: the fused tile-out writes the D32 image itself (row-major floats) instead of the 64-bit words
    float W, H; int32_t Wi, Hi;
    ShardInfo shard;
    TriRec* tris; uint32_t triCap;                      // 48-byte records, capacity per list shard
    TriRecC* trisC; uint32_t triCapC;                   // 32-byte records
    uint32_t* tileCount; uint32_t* tileBins; uint32_t binCap; uint32_t tilesX, tilesY;
    uint32_t* binPool; uint32_t binPoolChunks; uint32_t* binPoolCount;   // overflow chunks of this pass
    unsigned long long* binChunkTab; uint32_t binStamp; uint32_t binMaxChunks;   // [tile][binMaxChunks] serial << 32 | chunk
    unsigned long long* blockPool; uint32_t blockCap;   // pixel blocks of small clusters, granules of 16 bytes per list shard (0: path off)
    uint32_t blockForce;                                // debug: every launch takes the setup kernel's BLOCKS body
    ClipTri* clipTris; uint32_t clipTriCap; uint32_t pass;   // raster pass of the frame (0 / 1): clip / large count slot
    uint32_t* largeList; uint32_t largeCap;                  // records touching more than 2x2 tiles (binned by raster_bin_large_kernel)
    DeviceCounters* counters;
    uint2* tileOrder;                                   // [0].x = work item count, [1..] = {tile | slice << 12 | (slices-1) << 22, bin count}, heaviest first
    unsigned long long* tileSlabs;                      // one TILE x TILE accumulation slab per tile (all zero between uses)
    // fused HZB (single-GPU frames): the tile kernel reduces its finished 64x64 tile to mips 0..5
    uint32_t hzbFused;                                  // 0: off (later passes merge with global atomicMax)
    ChordHZBDesc hzbDesc;
    uint16_t* hzbMinA;                                  // temporary chain for stage 1 (first pass only) or NULL
    uint16_t* hzbMinB; uint16_t* hzbMaxB;               // chain kept as history
    // sharded frames: the texels go to the tile's slot of the exchange buffers instead (all-gathered, then hzb_untile_kernel)
    uint16_t* hzbExA;                                   // mid-frame exchange (min chain after the first pass), or NULL
    uint16_t* hzbExB;                                   // end-of-frame exchange (min | max | valid range | bin entries)
    uint32_t* tileRange;                                // per tile {min bits, max bits} of valid depth
    unsigned long long* tileClocks;                     // debug: per-tile elapsed wall clock ticks (DBG_TILE_CLOCKS)
    unsigned long long* tilePhase;                      // debug: 8 phase accumulators per tile
    uint32_t depthOnly, depthClamp;                     // PASS_TYPE_DEPTH (renderMeshDepth, mesh_raster.cpp:159-206): cull NONE, no id; depth clamp: near / far do not clip
    float biasConst, biasSlope;                         // vkCmdSetDepthBias(const, 0, slope), applied to the vertex depths of a depth-pass triangle
    uint32_t clearTiles;                                // first raster pass of a frame: tiles start from 0
    uint32_t* binHint;                                  // host-visible word: the longest bin of this pass (tile order kernel -> launch_raster of later frames), or NULL
    uint32_t* countHint;                                // host-visible word: the number of clusters this pass set up, or NULL
    uint32_t slotHot;                                   // bin length from which a tile counts as hot (SLOT_HOT; tests lower it)
    uint32_t* hotTiles;                                 // [1 + CHORD_HOT_TILES] this pass's hot tiles of the LAST frame (count, then tile | very hot << 31): written by the tile schedule, read by the block kernel's hot variant
    uint32_t tileSlots;                                 // tile workgroups the device holds at once (2 per CU); a pass with fewer non-empty tiles than that cuts its bins finer (tile_order_part), 0: never
    uint32_t tileSplitMin, tileSliceLen;                // bins longer than tileSplitMin entries are cut into slices of tileSliceLen (TILE_SPLIT_MIN, TILE_SLICE)
    uint2* tileOrderNext;                               // non-null: the tile kernel's extra workgroup makes the NEXT frame's schedule of this pass here, from this frame's bin counts (launch_raster)
    uint32_t orderAll;                                  // the schedule lists every (owned) tile, empty ones last, also in a pass that does not clear: a KEPT schedule of such a pass must name the tiles a later frame touches
    uint32_t orderKept;                                 // 1: tileOrder is the schedule of an EARLIER frame's first pass (launch_raster: TILE_ORDER_KEEP) -- the items and their order are taken from it, a tile's bin length and flags from the counter line of this pass
                                                        // 2: no schedule at all (later passes of a frame: launch_raster TILE_DIRECT) -- work item i is tile i, whole; a tile without entries is left alone
    uint32_t* heavyHint;                                // host-visible words [0] / [2]: this pass's serial (binStamp) stored by whoever finds the pass HEAVY (a bin beyond tileSplitMin entries, more than
                                                        // TILE_DIRECT_MAX_CLUSTERS clusters) / LIGHT (launch_raster: only a pass that was light a moment ago runs without a schedule), or NULL
    uint32_t debug;                                     // ablation switches (chordvis_set_debug), 0 in production
};
// The per-phase clocks of the setup kernels (debug bit 512) and of the tile kernel (bit 16) exist only in a build with
// -DRASTER_PROFILE=1 (python chord_amd/build.py --tag prof -DRASTER_PROFILE=1; the profile tools load that library): their
// accumulators are 64-bit values held across the kernels' main loops, and those loops are at the 102-SGPR limit -- in the
// product build they cost the block kernel 90 of its 145 v_readlane / v_writelane and 5 % of its time.
// The measurement-only ablation switches below (everything but DBG_NO_BLOCKS / DBG_FORCE_BLOCKS / DBG_FORCE_HOT, which tests
// run because they do not change results) exist only in a build with -DRASTER_ABLATION=1: tested at run time they keep
// values alive and conditions in the innermost loops of kernels that are register-bound.  chordvis_set_debug refuses a switch
// the library was not built with (device_layer.h).
#define ABL(p, flag) (RASTER_ABLATION && ((p).debug & (flag)) != 0u)
#define DBG_NO_PIXELS   1u    // skip every visibility write
#define DBG_NO_BIN      2u    // setup only: no records, no bins
#define DBG_TILE_CLOCKS 16u   // tile kernel writes its elapsed wall-clock ticks per tile
#define DBG_NO_OUT      128u  // tile kernel skips tile-out and the HZB reduction
#define DBG_NO_HZB      256u  // tile kernel writes the tile but skips the HZB reduction
#define DBG_SETUP_CLOCKS 512u // setup kernel accumulates per-wave phase ticks (header / vertex / triangle / reserve / emit)
#define DBG_NO_VIS_STORE 1024u // fused tile-out skips the visibility stores (HZB texels still written: culling unchanged)
#define DBG_NO_SPLIT    2048u // tile order kernel never cuts a bin into slices
#define DBG_NO_TINY     32u   // tile kernel skips the per-lane scan of tiny triangles
#define DBG_TILE_EXIT   64u   // tile kernel of passes >= 1 returns at once (launch-floor measurement)
#define DBG_NO_ENTRY    8192u // tile kernel fetches bin entries and records but does nothing with them
#define DBG_NO_BATCH    16384u // tile kernel skips the bin altogether (tile in + tile out only)
#define DBG_NO_UNITS    4096u // tile kernel skips the row units (entries are still fetched, set up, scanned and listed)
#define DBG_SKIP_LIGHT  1048576u // tile kernel: work items of tiles with fewer than 64 bin entries end at once (what the light tiles cost the pass)
#define DBG_SKIP_HEAVY  2097152u // ... of tiles with 64 entries or more

// Scalar loads and the kernel arguments re-read at their point of use.  The raster kernels take ONE by-value RasterParams; the
// compiler loads every field a kernel touches into scalar registers up front and keeps it there, and the loops of these
// kernels sit at the 102-SGPR limit: what does not fit lives in VGPR lanes, one v_readlane / v_writelane per use (the record
// setup kernel: 900 of them, a tenth of its instructions).  Fields needed once per cluster / tile are therefore read again
// from the kernel-argument segment (scalar cache) where they are used; the pointer goes through an empty asm so that the
// loads cannot be hoisted back to the top.
template <typename T>
__device__ __forceinline__ T scalar_load(const T* ptr)
{
    // uniform address, data written by an earlier kernel: the scalar cache is coherent at kernel boundaries
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(ptr));
}
__device__ __forceinline__ const RasterParams* kernel_args()
{
    unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return reinterpret_cast<const RasterParams*>(kp);
}

// number of set bits of a wave mask below this lane
__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ int32_t bcast(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ uint32_t bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ float bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

// Sharded frames: does this rank own tile (tx, ty) / a tile under the pixel rectangle?  (ranks == 1: always)
__device__ __forceinline__ bool owns_tile(const ShardInfo& s, int32_t tx, int32_t ty)
{
    if (s.ranks <= 1) return true;
    return shard_owns_tile(s, (uint32_t)tx, (uint32_t)ty);
}
__device__ __forceinline__ bool owns_rect(const ShardInfo& s, int32_t px0, int32_t py0, int32_t px1, int32_t py1)
{
    if (s.ranks <= 1) return true;
    return shard_owns_any_tile(s, (uint32_t)px0 >> TILE_SHIFT, (uint32_t)py0 >> TILE_SHIFT, (uint32_t)px1 >> TILE_SHIFT, (uint32_t)py1 >> TILE_SHIFT);
}

// depth clamp (shadow views): the near / far planes do not clip
__device__ __forceinline__ bool in_fast_volume_xy(const f4& h)
{
    return h.w > 0.0f && (GUARD_BAND * h.w + h.x) >= 0.0f && (GUARD_BAND * h.w - h.x) >= 0.0f &&
           (GUARD_BAND * h.w + h.y) >= 0.0f && (GUARD_BAND * h.w - h.y) >= 0.0f;
}

__device__ __forceinline__ bool in_fast_volume(const f4& h)
{
    return h.w > 0.0f && (h.w - h.z) >= 0.0f && h.z >= 0.0f &&
           (GUARD_BAND * h.w + h.x) >= 0.0f && (GUARD_BAND * h.w - h.x) >= 0.0f &&
           (GUARD_BAND * h.w + h.y) >= 0.0f && (GUARD_BAND * h.w - h.y) >= 0.0f;
}

// Per-triangle setup shared by every path.  Edge i is opposite vertex i:
// E0 = orient(V1,V2,P), E1 = orient(V2,V0,P), E2 = orient(V0,V1,P), times s so the interior is positive.
struct TriSetup {
    int32_t X[3], Y[3];
    float d0, e1, e2, invA;
    int32_t px0, py0, px1, py1;       // pixel bbox clamped to the screen
    int64_t area;                     // |2A|
    int32_t s;                        // orientation sign
    uint32_t payload;
};

// Returns false when the triangle is rejected (zero area, back face after snapping, empty bbox).
__device__ __forceinline__ bool narrow_extent(const TriSetup& ts)
{
    const int32_t extX = max(ts.X[0], max(ts.X[1], ts.X[2])) - min(ts.X[0], min(ts.X[1], ts.X[2]));
    const int32_t extY = max(ts.Y[0], max(ts.Y[1], ts.Y[2])) - min(ts.Y[0], min(ts.Y[1], ts.Y[2]));
    return extX <= (1 << 14) && extY <= (1 << 14);       // 32-bit edge functions are exact
}

// the 32-byte form holds a triangle iff its 16-bit deltas and 32-bit anchor are exact
__device__ __forceinline__ bool fits_compact(const TriSetup& ts)
{
    return narrow_extent(ts);
}

__device__ __forceinline__ void write_record_c(TriRecC* dst, const TriSetup& ts, const float d[3])
{
    TriRecC r;
    r.X0 = ts.X[0]; r.Y0 = ts.Y[0];
    r.dX1 = (int16_t)(ts.X[1] - ts.X[0]); r.dY1 = (int16_t)(ts.Y[1] - ts.Y[0]);
    r.dX2 = (int16_t)(ts.X[2] - ts.X[0]); r.dY2 = (int16_t)(ts.Y[2] - ts.Y[0]);
    r.d[0] = d[0]; r.d[1] = d[1]; r.d[2] = d[2];
    r.payload = ts.payload;
    *dst = r;
}

__device__ __forceinline__ bool tri_setup(TriSetup& ts, bool twoSided, int32_t Wi, int32_t Hi)
{
    const int32_t dx1 = ts.X[1] - ts.X[0], dy1 = ts.Y[1] - ts.Y[0], dx2 = ts.X[2] - ts.X[0], dy2 = ts.Y[2] - ts.Y[0];
    int64_t area2;
    // deltas below 2^15 (vertices within 128 px of vertex 0 -- nearly every triangle): both products fit 2^30 and their
    // difference an int32, from full-rate 24-bit multiplies; v_mul_lo/hi_u32 of the general form are quarter rate.  The choice is
    // made per WAVE (a scalar branch): as a per-lane select the compiler evaluates both forms for every triangle -- four
    // quarter-rate multiplies here and the int64 -> double -> float conversion below, the price of ~30 plain instructions.
    const bool small = (uint32_t)(dx1 + 32767) < 65535u && (uint32_t)(dy1 + 32767) < 65535u &&
                       (uint32_t)(dx2 + 32767) < 65535u && (uint32_t)(dy2 + 32767) < 65535u;
    const bool allSmall = __ballot(!small) == 0ull;
    if (allSmall) area2 = (int64_t)(__mul24(dx1, dy2) - __mul24(dx2, dy1));
    else area2 = (int64_t)dx1 * (int64_t)dy2 - (int64_t)dx2 * (int64_t)dy1;
    if (area2 == 0) return false;
    if (!twoSided && area2 > 0) return false;            // VK_CULL_MODE_BACK_BIT (mesh_raster.cpp:235)
    ts.s = area2 < 0 ? -1 : 1;
    ts.area = area2 < 0 ? -area2 : area2;
    const int32_t minX = min(ts.X[0], min(ts.X[1], ts.X[2])), maxX = max(ts.X[0], max(ts.X[1], ts.X[2]));
    const int32_t minY = min(ts.Y[0], min(ts.Y[1], ts.Y[2])), maxY = max(ts.Y[0], max(ts.Y[1], ts.Y[2]));
    ts.px0 = max(0, (minX + 127) >> 8);                  // arithmetic shift == floor
    ts.py0 = max(0, (minY + 127) >> 8);
    ts.px1 = min(Wi - 1, (maxX - 128) >> 8);
    ts.py1 = min(Hi - 1, (maxY - 128) >> 8);
    if (ts.px1 < ts.px0 || ts.py1 < ts.py0) return false;
    // (float)(double)area: one rounding of an exact value either way -- an int32 conversion when the area fits (|2A| < 2^31)
    if (allSmall) ts.invA = 1.0f / (float)(int32_t)ts.area;
    else ts.invA = 1.0f / (float)(double)ts.area;
    return true;
}

// Depth bias of a depth-pass triangle (oracle.c header item 10): o = slope * max(|dz/dx|, |dz/dy|) + const * 2^(exponent(max |d|) - 23)
__device__ __forceinline__ float depth_bias(const TriSetup& ts, const float d[3], float biasConst, float biasSlope)
{
    const int64_t s = ts.s;
    const float invA = ts.invA;
    const int64_t a1 = -s * (int64_t)(ts.Y[0] - ts.Y[2]), b1 = s * (int64_t)(ts.X[0] - ts.X[2]);
    const int64_t a2 = -s * (int64_t)(ts.Y[1] - ts.Y[0]), b2 = s * (int64_t)(ts.X[1] - ts.X[0]);
    const float e1 = d[1] - d[0], e2 = d[2] - d[0];
    const float dzdx = ((float)(double)(a1 * 256) * invA) * e1 + ((float)(double)(a2 * 256) * invA) * e2;
    const float dzdy = ((float)(double)(b1 * 256) * invA) * e1 + ((float)(double)(b2 * 256) * invA) * e2;
    const float m = fmaxf(fabsf(dzdx), fabsf(dzdy));
    const float mz = fmaxf(fabsf(d[0]), fmaxf(fabsf(d[1]), fabsf(d[2])));
    const int32_t e = (int32_t)((__float_as_uint(mz) >> 23) & 0xFFu) - 23;
    const float r = (e > 0 && e < 255) ? __uint_as_float((uint32_t)e << 23) : 0.0f;
    return biasSlope * m + biasConst * r;
}

// Setup of a record that raster_setup_kernel / raster_clip_kernel already validated: bbox + stored sign / invA.
__device__ __forceinline__ void tri_setup_from_compact(TriSetup& ts, const TriRecC& r, int32_t Wi, int32_t Hi)
{
    ts.X[0] = r.X0; ts.Y[0] = r.Y0;
    ts.X[1] = r.X0 + r.dX1; ts.Y[1] = r.Y0 + r.dY1;
    ts.X[2] = r.X0 + r.dX2; ts.Y[2] = r.Y0 + r.dY2;
    ts.payload = r.payload;
    // the same integers tri_setup reduced: |deltas| <= 2^14, so the 32-bit product is exact
    const int32_t area2 = (int32_t)r.dX1 * (int32_t)r.dY2 - (int32_t)r.dX2 * (int32_t)r.dY1;
    ts.s = area2 < 0 ? -1 : 1;
    ts.invA = 1.0f / (float)(area2 < 0 ? -area2 : area2);           // (|2A| <= 2^29: the int32 conversion is the canonical (float)(double)|2A|)
    ts.d0 = r.d[0]; ts.e1 = r.d[1] - r.d[0]; ts.e2 = r.d[2] - r.d[0];
    const int32_t minX = min(ts.X[0], min(ts.X[1], ts.X[2])), maxX = max(ts.X[0], max(ts.X[1], ts.X[2]));
    const int32_t minY = min(ts.Y[0], min(ts.Y[1], ts.Y[2])), maxY = max(ts.Y[0], max(ts.Y[1], ts.Y[2]));
    ts.px0 = max(0, (minX + 127) >> 8);
    ts.py0 = max(0, (minY + 127) >> 8);
    ts.px1 = min(Wi - 1, (maxX - 128) >> 8);
    ts.py1 = min(Hi - 1, (maxY - 128) >> 8);
    ts.area = 0;
}

__device__ __forceinline__ void tri_setup_from_record(TriSetup& ts, const TriRec& r, int32_t Wi, int32_t Hi)
{
#pragma unroll
    for (int i = 0; i < 3; i++) { ts.X[i] = r.X[i]; ts.Y[i] = r.Y[i]; }
    ts.payload = r.payload;
    ts.s = (r.twoSided & 2u) ? -1 : 1;
    ts.invA = __uint_as_float(r.pad);
    ts.d0 = r.d[0]; ts.e1 = r.d[1] - r.d[0]; ts.e2 = r.d[2] - r.d[0];
    const int32_t minX = min(ts.X[0], min(ts.X[1], ts.X[2])), maxX = max(ts.X[0], max(ts.X[1], ts.X[2]));
    const int32_t minY = min(ts.Y[0], min(ts.Y[1], ts.Y[2])), maxY = max(ts.Y[0], max(ts.Y[1], ts.Y[2]));
    ts.px0 = max(0, (minX + 127) >> 8);
    ts.py0 = max(0, (minY + 127) >> 8);
    ts.px1 = min(Wi - 1, (maxX - 128) >> 8);
    ts.py1 = min(Hi - 1, (maxY - 128) >> 8);
    ts.area = 0;
}


// ---- record + bin emission --------------------------------------------------------------------

// What the record kernel's emission (list and bin reservations, record stores) needs of the kernel arguments, per cluster.
struct RecordEmitParams {
    DeviceCounters* counters; ClipTri* clipTris; uint32_t clipTriCap; uint32_t pass;
    TriRec* tris; uint32_t triCap; TriRecC* trisC; uint32_t triCapC; uint32_t* largeList; uint32_t largeCap;
    uint32_t* tileCount; uint32_t* tileBins; uint32_t binCap; uint32_t tilesX;
    uint32_t* binPool; uint32_t binPoolChunks; uint32_t* binPoolCount;
    unsigned long long* binChunkTab; uint32_t binStamp; uint32_t binMaxChunks;
    ShardInfo shard;
};
__device__ __forceinline__ RecordEmitParams load_record_emit_params()
{
    const RasterParams* q = kernel_args();
    RecordEmitParams e;
    e.counters = scalar_load(&q->counters); e.clipTris = scalar_load(&q->clipTris); e.clipTriCap = scalar_load(&q->clipTriCap); e.pass = scalar_load(&q->pass);
    e.tris = scalar_load(&q->tris); e.triCap = scalar_load(&q->triCap); e.trisC = scalar_load(&q->trisC); e.triCapC = scalar_load(&q->triCapC);
    e.largeList = scalar_load(&q->largeList); e.largeCap = scalar_load(&q->largeCap);
    e.tileCount = scalar_load(&q->tileCount); e.tileBins = scalar_load(&q->tileBins); e.binCap = scalar_load(&q->binCap); e.tilesX = scalar_load(&q->tilesX);
    e.binPool = scalar_load(&q->binPool); e.binPoolChunks = scalar_load(&q->binPoolChunks); e.binPoolCount = scalar_load(&q->binPoolCount);
    e.binChunkTab = scalar_load(&q->binChunkTab); e.binStamp = scalar_load(&q->binStamp); e.binMaxChunks = scalar_load(&q->binMaxChunks);
    e.shard.ranks = scalar_load(&q->shard.ranks); e.shard.rank = scalar_load(&q->shard.rank); e.shard.slotsPerRank = 0u; e.shard.tilesX = 0u;
    e.shard.ownedRows = scalar_load(&q->shard.ownedRows); e.shard.tileSlot = nullptr;
    return e;
}

// One bin slot per lane, reserved with ONE atomic per distinct tile in the wave: lanes that target the
// same tile elect a leader (pure ALU), all leaders issue their atomicAdd in a single wave instruction, and
// the base is handed back through the lanes.  (64 lanes hitting one counter would serialise at the L2.)
struct BinElect { int leader; uint32_t rank, group; };

__device__ __forceinline__ BinElect wave_bin_elect(bool has, uint32_t tile, uint32_t lane)
{
    BinElect e; e.leader = 0; e.rank = 0; e.group = 0;
    unsigned long long todo = __ballot(has);
    const unsigned long long lt = (1ull << lane) - 1ull;
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        const uint32_t k = bcast(tile, l);
        const unsigned long long same = __ballot(has && tile == k);
        if (has && tile == k) { e.leader = l; e.rank = (uint32_t)__popcll(same & lt); e.group = (uint32_t)__popcll(same); }
        todo &= ~same;
    }
    return e;
}

// ---- tile bins ----------------------------------------------------------------------------------
// Entry `slot` of a tile's bin: the first binCap entries have a fixed home; beyond that, entries live in
// 1024-entry chunks from a pool.  The lane that drew the first slot of a chunk allocates it and publishes
// `serial << 32 | id` in the tile's chunk table; lanes that drew other slots of that chunk wait for the entry
// to carry this pass's serial.  Every allocation for the slots a wave has drawn is issued before any of its lanes
// starts waiting (bin_alloc for all of them, then bin_put), and an allocator never waits, so the wait always ends;
// it is bounded anyway (overflow bit 2) so that a logic error cannot hang the device.
__device__ __forceinline__ uint32_t bin_capacity(const RasterParams& p) { return p.binCap + p.binMaxChunks * CHORD_BIN_CHUNK; }

// step 1 of a bin write: the lane that drew the first slot of an overflow chunk allocates it and publishes it.  Never waits.
template <class P>
__device__ __forceinline__ void bin_alloc(const P& p, uint32_t tile, uint32_t slot)
{
#ifdef EXP_NO_CHUNKS
    return;
#endif
    if (slot < p.binCap) return;
    const uint32_t o = slot - p.binCap, j = o >> CHORD_BIN_CHUNK_SHIFT;
    if (j >= p.binMaxChunks || (o & (CHORD_BIN_CHUNK - 1u)) != 0u) return;
    uint32_t id = atomicAdd(p.binPoolCount, 1u);
    if (id >= p.binPoolChunks) { id = CHORD_BIN_CHUNK_INVALID; atomicOr(&p.counters->overflow, 1u); }
    __hip_atomic_store(p.binChunkTab + (__umul24(tile, p.binMaxChunks) + j), ((unsigned long long)p.binStamp << 32) | id,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// step 2: store the entry; a slot inside an overflow chunk waits for the chunk's allocator (the drawer of the chunk's
// first slot, whose atomicAdd preceded this lane's: it has passed every wait of its own earlier writes and reaches its
// bin_alloc without waiting -- PROVIDED every caller runs bin_alloc for ALL the slots it has drawn before its first
// bin_put; wave_bin_commit draws eight slots per lane at once and therefore allocates for all eight first).
template <class P>
__device__ __forceinline__ void bin_put(const P& p, uint32_t tile, uint32_t slot, uint32_t gi)
{
    // (tile < 4096 and the fixed part of a bin at most 2^20 entries: the index is a full-rate 24-bit multiply and fits 32 bits)
#ifdef EXP_NO_CHUNKS
    slot &= p.binCap - 1u;                                      // measurement only (wrong image): no entry ever lives in a pool chunk
#endif
    if (slot < p.binCap) { p.tileBins[__umul24(tile, p.binCap) + slot] = gi; return; }
    const uint32_t o = slot - p.binCap, j = o >> CHORD_BIN_CHUNK_SHIFT;
    if (j >= p.binMaxChunks) { atomicOr(&p.counters->overflow, 1u); return; }
    const unsigned long long* ent = p.binChunkTab + (__umul24(tile, p.binMaxChunks) + j);
    const unsigned long long stamp = (unsigned long long)p.binStamp << 32;
    unsigned long long e = 0;
    uint32_t spins = 0;
    for (;;) {
        e = __hip_atomic_load(ent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((e & 0xFFFFFFFF00000000ull) == stamp) break;
        if (++spins > (1u << 20)) { atomicOr(&p.counters->overflow, 4u); return; }
        __builtin_amdgcn_s_sleep(4);
    }
    const uint32_t id = (uint32_t)e;
    if (id != CHORD_BIN_CHUNK_INVALID) p.binPool[((size_t)id << CHORD_BIN_CHUNK_SHIFT) + (o & (CHORD_BIN_CHUNK - 1u))] = gi;
}

// one slot drawn, written at once (the looped binners: a slot is drawn, allocated for and stored within one iteration)
__device__ __forceinline__ void bin_store(const RasterParams& p, uint32_t tile, uint32_t slot, uint32_t gi)
{
    bin_alloc(p, tile, slot);
    bin_put(p, tile, slot, gi);
}

// Bin reservations of two records per lane (triangles lane and lane + 64 of the meshlet) whose clamped bboxes touch
// at most 2x2 tiles each, split into ISSUE and COMMIT so that the caller can put every atomic of a cluster -- list
// reservations and bin reservations -- into ONE memory round trip and do all its stores afterwards (returning
// atomics and stores share the in-order vmcnt: a reservation issued behind the 48-byte record stores waits for
// them; reserve + emit were 30 of 40 us per cluster in the profile of config 3):
//   * the primary tile of every record (almost all entries) is reserved once per distinct tile of the wave;
//   * the up to three further tiles of a record that straddles a tile boundary (few lanes) are reserved by the
//     lane itself, one entry each.
struct BinTicket {
    uint32_t tileA[4], tileB[4], slotA[4], slotB[4];
    uint32_t has;                 // bit r: A has tile r, bit 4 + r: B
    uint32_t eA[4], eB[4];        // per tile slot r: leader lane | rank among the wave's lanes with the same tile << 8 (wave_bin_elect)
};

template <class P>
__device__ __forceinline__ void wave_bin_issue(const P& p, bool emitA, const TriSetup& tsA, bool emitB, const TriSetup& tsB,
                                               uint32_t lane, BinTicket& k, const bool masked = false)
{
    auto tile_of = [&](const TriSetup& ts, int r, bool emit, uint32_t& tile) -> bool {
        const int32_t tx0 = ts.px0 >> TILE_SHIFT, tx1 = ts.px1 >> TILE_SHIFT, ty0 = ts.py0 >> TILE_SHIFT, ty1 = ts.py1 >> TILE_SHIFT;
        const int32_t tx = (r & 1) ? tx1 : tx0, ty = (r & 2) ? ty1 : ty0;
        bool has = emit && !((r & 1) && tx1 == tx0) && !((r & 2) && ty1 == ty0);
        if (has) has = owns_tile(p.shard, tx, ty);
        tile = has ? __umul24((uint32_t)ty, p.tilesX) + (uint32_t)tx : 0u;
        return has;
    };
    k.has = 0u;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        k.slotA[r] = 0u; k.slotB[r] = 0u;
        if (tile_of(tsA, r, emitA, k.tileA[r])) k.has |= 1u << r;
        if (tile_of(tsB, r, emitB, k.tileB[r])) k.has |= 16u << r;
    }
    // ONE atomic per distinct tile of the wave, for the primary tile of a record AND for the up to three further tiles of a record that
    // straddles a tile boundary.  (Until round 6 the further tiles were reserved by every lane for itself: 18 % of config 3's bin entries,
    // 85 k returning atomics per frame against 12 k leader atomics, a straddling cluster's 30-60 of them on ONE counter line in one
    // wave instruction -- a line retires ~88 per microsecond: the reservation round trip was 8 of the 19.5 us a cluster took in config
    // 3's first pass, tools/setup_profile.py.)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const BinElect a = wave_bin_elect((k.has & (1u << r)) != 0u, k.tileA[r], lane);
        const BinElect b = wave_bin_elect((k.has & (16u << r)) != 0u, k.tileB[r], lane);
        k.eA[r] = (uint32_t)a.leader | a.rank << 8; k.eB[r] = (uint32_t)b.leader | b.rank << 8;
        if ((k.has & (1u << r)) && (int)lane == a.leader) k.slotA[r] = atomicAdd(&p.tileCount[(size_t)k.tileA[r] * TC_STRIDE], a.group);
        if ((k.has & (16u << r)) && (int)lane == b.leader) k.slotB[r] = atomicAdd(&p.tileCount[(size_t)k.tileB[r] * TC_STRIDE], b.group);
    }
    if (masked) {
        // an alpha-tested cluster: the tiles its triangles are binned into are work items of the masked pass (a flag per tile, plain
        // stores of the same value: one per distinct primary tile of the wave, one per straddled tile of a lane)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if ((k.has & (1u << r)) && lane == (k.eA[r] & 0xFFu)) p.tileCount[(size_t)k.tileA[r] * TC_STRIDE + TC_MASKED] = 1u;
            if ((k.has & (16u << r)) && lane == (k.eB[r] & 0xFFu)) p.tileCount[(size_t)k.tileB[r] * TC_STRIDE + TC_MASKED] = 1u;
        }
    }
}

// (a record that did not fit its list leaves its reserved bin slots unwritten: the frame is reported incomplete
// anyway, and a stale entry of an earlier frame is a valid index)
template <class P>
__device__ __forceinline__ void wave_bin_commit(const P& p, BinTicket& k, bool okA, uint32_t giA, bool okB, uint32_t giB)
{
#pragma unroll
    for (int r = 0; r < 4; r++) {
        k.slotA[r] = __shfl(k.slotA[r], (int)(k.eA[r] & 0xFFu), 64) + (k.eA[r] >> 8);
        k.slotB[r] = __shfl(k.slotB[r], (int)(k.eB[r] & 0xFFu), 64) + (k.eB[r] >> 8);
    }
    // all eight slots of a lane were drawn together (wave_bin_issue): every chunk allocation they owe comes before
    // the first wait -- a wait ahead of a later allocation could close a cycle between two waves (each waiting for a
    // chunk the other allocates in a later step), which costs 2^20 spins and drops entries.  A record that did not fit
    // its list still allocates (its slots are drawn; other lanes may be waiting for the chunk).
    if (k.has) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (k.has & (1u << r)) bin_alloc(p, k.tileA[r], k.slotA[r]);
            if (k.has & (16u << r)) bin_alloc(p, k.tileB[r], k.slotB[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (okA && (k.has & (1u << r))) bin_put(p, k.tileA[r], k.slotA[r], giA);
        if (okB && (k.has & (16u << r))) bin_put(p, k.tileB[r], k.slotB[r], giB);
    }
}

__device__ __forceinline__ bool touches_many_tiles(const TriSetup& ts)
{
    return ((ts.px1 >> TILE_SHIFT) - (ts.px0 >> TILE_SHIFT)) > 1 || ((ts.py1 >> TILE_SHIFT) - (ts.py0 >> TILE_SHIFT)) > 1;
}

__device__ __forceinline__ void write_record(TriRec* dst, const TriSetup& ts, const float d[3], bool twoSided, bool masked = false)
{
    TriRec r;
#pragma unroll
    for (int i = 0; i < 3; i++) { r.X[i] = ts.X[i]; r.Y[i] = ts.Y[i]; r.d[i] = d[i]; }
    r.payload = ts.payload;
    r.twoSided = (twoSided ? 1u : 0u) | (ts.s < 0 ? 2u : 0u) | (masked ? 4u : 0u);   // bit 1: orientation sign of the snapped triangle, bit 2: a TriRecMaskExt follows
    r.pad = __float_as_uint(ts.invA);                             // 1 / float(2A): the tile kernel does not redo the division
    *dst = r;
}

// ---- masked materials (mesh_raster.hlsl:34-38,107-112,198-204) -----------------------------------------------------------
// The reference samples the base-colour texture in the pixel shader and clip()s on its alpha.  Level of detail and
// filtering are the sampler hardware's business there; here they are pinned (oracle.c header item 9):
// perspective-correct uv from u/w, v/w, 1/w; ONE level per triangle from the ratio of its doubled uv area (in level-0
// texels) to its doubled pixel area, level = floor(log2(ratio)) >> 1; nearest or bilinear as the sampler's min / mag
// filter says; wrap modes on the integer texel index.  A masked triangle takes a 48-byte record (bit 2 of `twoSided`)
// plus a TriRecMaskExt in the next two slots of the same list.
__device__ __forceinline__ bool filter_is_linear(uint32_t f)
{
    return f == CHORD_FILTER_LINEAR || f == CHORD_FILTER_LINEAR_MIPMAP_NEAREST || f == CHORD_FILTER_LINEAR_MIPMAP_LINEAR;
}

__device__ __forceinline__ uint32_t mask_level_filter(const DMaterial& m, int64_t absArea2, const float u[3], const float v[3])
{
    uint32_t level = 0u;
    bool linear = filter_is_linear(m.magFilter);
    if (m.texOffset != 0xFFFFFFFFu) {
        const float texels = (float)m.texWidth * (float)m.texHeight;
        const float auv = fabsf((u[1] - u[0]) * (v[2] - v[0]) - (u[2] - u[0]) * (v[1] - v[0])) * texels;
        const float apx = (float)(double)absArea2 * (1.0f / 65536.0f);
        const float ratio = auv / apx;
        if (ratio >= 1.0f) {
            const int32_t e = (int32_t)((__float_as_uint(ratio) >> 23) & 0xFFu) - 127;
            level = min((uint32_t)(e >> 1), m.texMips - 1u);
            linear = filter_is_linear(m.minFilter);
        }
    }
    return level | (linear ? 256u : 0u);
}

__device__ __forceinline__ void write_mask_ext(TriRec* slot, const DMaterial* mp, uint32_t material, int64_t absArea2,
                                               const float u[3], const float v[3], const float w[3])
{
    // (the header fields by value; the one level the triangle samples is read where it is known)
    DMaterial m;
    m.texOffset = mp->texOffset; m.texWidth = mp->texWidth; m.texHeight = mp->texHeight; m.texMips = mp->texMips;
    m.minFilter = mp->minFilter; m.magFilter = mp->magFilter; m.wrapS = mp->wrapS; m.wrapT = mp->wrapT;
    m.alphaFactor = mp->alphaFactor; m.alphaCutOff = mp->alphaCutOff;
    TriRecMaskExt e;
#pragma unroll
    for (int i = 0; i < 3; i++) { e.iw[i] = 1.0f / w[i]; e.uw[i] = u[i] * e.iw[i]; e.vw[i] = v[i] * e.iw[i]; }
    e.levelFilter = mask_level_filter(m, absArea2, u, v);
    (void)material;
    const uint32_t level = e.levelFilter & 0xFFu;
    e.levelBase = 0xFFFFFFFFu; e.dims = 0u; e.magicS = e.biasS = e.magicT = e.biasT = 0u;
    if (m.texOffset != 0xFFFFFFFFu) {
        const DMatLevel L = mp->levels[level];                  // (resolved at upload: no loop over the levels, no division here)
        e.levelBase = L.base; e.dims = L.dims;
        e.magicS = L.magicS; e.biasS = L.biasS; e.magicT = L.magicT; e.biasT = L.biasT;
    }
    e.wraps = (m.wrapS & 0xFFFFu) | m.wrapT << 16;
    e.alphaFactor = m.alphaFactor; e.alphaCutOff = m.alphaCutOff;
    // (the 76 bytes in use: four 16-byte stores and three dwords; the rest of the second slot is never read)
    uint4* dst = reinterpret_cast<uint4*>(slot);
    const uint4* src = reinterpret_cast<const uint4*>(&e);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
    reinterpret_cast<uint32_t*>(slot)[16] = e.biasS; reinterpret_cast<uint32_t*>(slot)[17] = e.magicT; reinterpret_cast<uint32_t*>(slot)[18] = e.biasT;
}

#ifndef EXP_MASKED
#define EXP_MASKED 0            // measurement builds only (results differ): 1 no fetch, 2 nearest everywhere, 4 sizes treated as powers of two, 8 unguarded shared-reciprocal divisions
#endif
// Texel indices are 32-bit here: texel_floor maps everything beyond +-1e9 to 0, so an index and its +1 neighbour fit an int32
// (the oracle's 64-bit arithmetic gives the same values); a 64-bit modulo is ~200 instructions on this GPU and the bilinear
// fetch of round 2 did eight of them per covered pixel.  Power-of-two periods (every level of a power-of-two texture) wrap with
// a mask: i & (n - 1) is the non-negative remainder in two's complement.  Any other period divides by a constant of the level
// (DMatLevel: magic = floor(2^32 / period), bias = a multiple of the period >= 2^30): iu = i + bias is in [0, 2^31) and has
// i's remainder; floor(iu * magic / 2^32) is floor(iu / period) or one less (iu * (2^32 / period - magic) / 2^32 < 1/2), so one
// multiply-high, one multiply and one conditional subtraction replace the ~50 issue slots of a 32-bit signed remainder.
__device__ __forceinline__ int32_t period_mod(int32_t i, int32_t period, uint32_t magic, uint32_t bias)
{
    if ((EXP_MASKED & 4) || magic == 0u) return i & (period - 1);
    const uint32_t iu = (uint32_t)i + bias;
    uint32_t r = iu - __umulhi(iu, magic) * (uint32_t)period;
    if (r >= (uint32_t)period) r -= (uint32_t)period;
    return (int32_t)r;
}
__device__ __forceinline__ int32_t wrap_index(int32_t i, int32_t n, uint32_t mode, uint32_t magic, uint32_t bias)
{
    if (mode == CHORD_WRAP_CLAMP_TO_EDGE) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    if (mode == CHORD_WRAP_MIRRORED_REPEAT) {
        const int32_t m = period_mod(i, 2 * n, magic, bias);
        return m < n ? m : 2 * n - 1 - m;
    }
    return period_mod(i, n, magic, bias);
}

// wrap_index(i) and wrap_index(i + 1) with ONE remainder: the neighbour's follows from the remainder's successor
__device__ __forceinline__ void wrap_pair(int32_t i, int32_t n, uint32_t mode, uint32_t magic, uint32_t bias, int32_t& w0, int32_t& w1)
{
    if (mode == CHORD_WRAP_CLAMP_TO_EDGE) {
        w0 = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
        w1 = i + 1 < 0 ? 0 : (i + 1 > n - 1 ? n - 1 : i + 1);
        return;
    }
    const bool mirror = mode == CHORD_WRAP_MIRRORED_REPEAT;
    const int32_t period = mirror ? 2 * n : n;
    const int32_t m = period_mod(i, period, magic, bias);
    const int32_t m1 = m + 1 == period ? 0 : m + 1;                          // (i + 1) mod period
    w0 = mirror ? (m < n ? m : 2 * n - 1 - m) : m;
    w1 = mirror ? (m1 < n ? m1 : 2 * n - 1 - m1) : m1;
}

__device__ __forceinline__ int32_t texel_floor(float x)
{
    if (!(fabsf(x) < 1.0e9f)) return 0;
    return (int32_t)floorf(x);
}

// One level of a material's alpha texture, resolved once per row unit (not once per pixel): base address, size, wraps, filter.
struct AlphaLevel {
    const uint8_t* base;            // NULL: no texture (white fallback, alpha 1)
    int32_t W, H;
    float fW, fH;
    uint32_t wrapS, wrapT;
    uint32_t magicS, biasS, magicT, biasT;
    bool linear;
};
__device__ __forceinline__ AlphaLevel alpha_level(const uint8_t* __restrict__ texAlpha, const TriRecMaskExt& e)
{
    const uint32_t levelBase = e.levelBase, dims = e.dims;
    AlphaLevel a;
    a.base = nullptr; a.W = 1; a.H = 1; a.fW = 1.0f; a.fH = 1.0f; a.wrapS = e.wraps & 0xFFFFu; a.wrapT = e.wraps >> 16; a.linear = (e.levelFilter & 256u) != 0u;
    a.magicS = e.magicS; a.biasS = e.biasS; a.magicT = e.magicT; a.biasT = e.biasT;
    if (levelBase == 0xFFFFFFFFu) return a;
    a.W = (int32_t)(dims & 0xFFFFu) + 1; a.H = (int32_t)(dims >> 16) + 1;
    a.fW = (float)a.W; a.fH = (float)a.H;
    a.base = texAlpha + levelBase;
    return a;
}
__device__ __forceinline__ float sample_alpha(const AlphaLevel& t, float u, float v)
{
    if (!t.base) return 1.0f;
    if (EXP_MASKED & 1) return 1.0f;
    if (!t.linear || (EXP_MASKED & 2)) {
        const int32_t ix = wrap_index(texel_floor(u * t.fW), t.W, t.wrapS, t.magicS, t.biasS), iy = wrap_index(texel_floor(v * t.fH), t.H, t.wrapT, t.magicT, t.biasT);
        return (float)t.base[iy * t.W + ix] * (1.0f / 255.0f);
    }
    const float x = u * t.fW - 0.5f, y = v * t.fH - 0.5f;
    const int32_t x0 = texel_floor(x), y0 = texel_floor(y);
    float fx = x - (float)x0, fy = y - (float)y0;
    if (!(fabsf(x) < 1.0e9f)) fx = 0.0f;
    if (!(fabsf(y) < 1.0e9f)) fy = 0.0f;
    int32_t ix0, ix1, iy0, iy1;
    wrap_pair(x0, t.W, t.wrapS, t.magicS, t.biasS, ix0, ix1);
    wrap_pair(y0, t.H, t.wrapT, t.magicT, t.biasT, iy0, iy1);
    const float a00 = (float)t.base[iy0 * t.W + ix0] * (1.0f / 255.0f), a10 = (float)t.base[iy0 * t.W + ix1] * (1.0f / 255.0f);
    const float a01 = (float)t.base[iy1 * t.W + ix0] * (1.0f / 255.0f), a11 = (float)t.base[iy1 * t.W + ix1] * (1.0f / 255.0f);
    const float top = a00 + (a10 - a00) * fx, bot = a01 + (a11 - a01) * fx;
    return top + (bot - top) * fy;
}

// extension of a masked triangle of the cluster in flight: texture coordinates from the vertex stream, w from the
// wave's LDS copy of the clip-space vertices.
__device__ __forceinline__ void setup_emit_mask_ext(const RasterParams& p, TriRec* slot, uint32_t triWord, const float uv[6],
                                                 const float* lW, uint32_t material, int64_t absArea2)
{
    float u[3], v[3], w[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        u[i] = uv[2 * i]; v[i] = uv[2 * i + 1];               // (fetched at the start of the triangle phase; 0 without texture coordinates)
        w[i] = lW[(triWord >> (8 * i)) & 0xFFu];
    }
    write_mask_ext(slot, &p.materials[material], material, absArea2, u, v, w);
}

template <int PITCH>
__device__ __forceinline__ void tile_raster_narrow(unsigned long long* tile, const TriSetup& ts, int32_t ox, int32_t oy,
                                                   int32_t x0, int32_t y0, int32_t x1, int32_t y1, bool noPixels, const bool clampZ);

// wave-wide minimum of a and b, maximum of c and d (all 64 lanes active), results wave-uniform.  Each step's operand is a
// DPP-permuted copy (within quads, within rows of 16, then lane 15 / 31 of a row broadcast to the rows above) folded into the
// v_min / v_max itself: 24 VALU instructions for the four reductions and no LDS traffic -- the __shfl_xor form was 24
// ds_bpermute round trips plus ~40 instructions of address arithmetic and min / max.  Written as one asm block because the
// compiler expands the update_dpp builtin to copy + nop + v_mov_dpp + min; the four chains are interleaved, which also keeps
// every DPP read two instructions behind the write it depends on (the hazard the leading s_nop covers for the inputs).
__device__ __forceinline__ void wave_min2_max2(int32_t& a, int32_t& b, int32_t& c, int32_t& d)
{
#define DPP_STEP(ctrl)                                                \
    "v_min_i32_dpp %0, %0, %0 " ctrl "\n\t"                           \
    "v_min_i32_dpp %1, %1, %1 " ctrl "\n\t"                           \
    "v_max_i32_dpp %2, %2, %2 " ctrl "\n\t"                           \
    "v_max_i32_dpp %3, %3, %3 " ctrl "\n\t"
    asm volatile("s_nop 1\n\t"
                 DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 DPP_STEP("row_ror:4 row_mask:0xf bank_mask:0xf")
                 DPP_STEP("row_ror:8 row_mask:0xf bank_mask:0xf")
                 DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef DPP_STEP
    a = __builtin_amdgcn_readlane(a, 63); b = __builtin_amdgcn_readlane(b, 63);
    c = __builtin_amdgcn_readlane(c, 63); d = __builtin_amdgcn_readlane(d, 63);
}

// wave-wide sums of a and b (all 64 lanes active), results wave-uniform: the same DPP ladder with v_add
__device__ __forceinline__ void wave_sum2(uint32_t& a, uint32_t& b)
{
#define DPP_STEP(ctrl)                                                \
    "v_add_u32_dpp %0, %0, %0 " ctrl "\n\t"                           \
    "v_add_u32_dpp %1, %1, %1 " ctrl "\n\t"
    asm volatile("s_nop 1\n\t"
                 DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") "s_nop 0\n\t"
                 DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") "s_nop 0\n\t"
                 DPP_STEP("row_ror:4 row_mask:0xf bank_mask:0xf") "s_nop 0\n\t"
                 DPP_STEP("row_ror:8 row_mask:0xf bank_mask:0xf") "s_nop 0\n\t"
                 DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") "s_nop 0\n\t"
                 DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(a), "+v"(b));
#undef DPP_STEP
    a = (uint32_t)__builtin_amdgcn_readlane((int)a, 63); b = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
}

// ---- the per-cluster setup kernel -------------------------------------------------------------
enum { K_NONE = 0, K_EMIT = 1, K_CLIP = 2 };
#define WIN CHORD_BLOCK_WIN
#define DBG_NO_BLOCKS 32768u     // small clusters take the record path too (A/B of the pixel blocks; results identical)
#define DBG_FORCE_BLOCKS 65536u  // the setup kernel takes its BLOCKS body whatever the cluster count (tests: small scenes)
#define DBG_FORCE_HOT 262144u    // the block kernel's hot-tile variant whatever the hint says, tiles hot from 64 entries (tests: small scenes)

// cmds / count: the list this launch sets up (the input list, or what the block kernel left over: raster_setup_kernel).
// firstCmd: the three words of THIS wave's first command (index blockIdx.x * 4 + wave), requested by the caller together with the count
// -- one round trip instead of two in front of everything else a wave does (a short list is one cluster per wave: the kernel is the
// chain count -> command -> records -> indices -> positions -> matrix -> reservations, seven dependent round trips until round 6).
template <bool MASKED>
__device__ __forceinline__ void raster_setup_body(const RasterParams& p, const ChordDrawCmd* __restrict__ cmds, const uint32_t count, float (*sVert)[4][LDS_VERTS],
                                                  const uint32_t firstCmd0, const uint32_t firstCmd1, const uint32_t firstCmd2, const bool firstCmdValid)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float* lX = sVert[0][wave]; float* lY = sVert[1][wave]; float* lW = sVert[2][wave];
    float* lU = sVert[3][wave]; float* lV = sVert[4][wave]; float* lD = sVert[5][wave];

    const uint32_t listShard = (blockIdx.x * 4u + wave) % CHORD_LIST_SHARDS;

    // Wave-uniform header of a cluster (scalar loads: the addresses are uniform).  The chain command -> meshlet /
    // object records -> index stream -> positions is four dependent memory round trips per cluster; the header of
    // the NEXT cluster is fetched while the current one is processed (command at the top of the iteration, records
    // after the vertex phase), which takes two of them off the critical path.
    struct Header {
        uint32_t objectId, meshletId, slot, V, T, dataOffset, vertexBase, matFlags;
        bool twoSided;
    };
    // (scalar loads through the constant address space: the addresses are wave-uniform, and a vector load + v_readfirstlane per
    // dword was 26 VALU instructions per header -- the record kernel is bound by VALU issue on dense scenes like the block kernel)
    auto header_of = [&](uint32_t objectId, uint32_t meshletId, uint32_t slot) -> Header {
        Header h;
        h.objectId = objectId; h.meshletId = meshletId; h.slot = slot;
        const RasterParams* q = kernel_args();                  // (scene pointers: read where they are used, not held across the loop)
        const DMeshlet* __restrict__ mm = &scalar_load(&q->meshlets)[h.meshletId];
        const uint32_t vt = scalar_load(&mm->vertexTriangleCount);
        h.V = vt & 0xFFu; h.T = (vt >> 8) & 0xFFu;
        h.dataOffset = scalar_load(&mm->dataOffset);
        h.vertexBase = scalar_load(&mm->vertexBase);
        h.matFlags = scalar_load(&scalar_load(&q->objStatic)[h.objectId].matFlags);
        h.twoSided = (h.matFlags & CHORD_MATFLAG_TWO_SIDED) != 0u;
        if (CHORD_MATFLAG_ALPHA(h.matFlags) >= CHORD_ALPHA_BLEND) h.T = 0u;   // blended: in no bucket of renderMesh (mesh_raster.cpp:224)
        return h;
    };
    auto load_header = [&](uint32_t i) -> Header {
        const uint32_t k = __builtin_amdgcn_readfirstlane(min(i, count - 1u));
        const uint32_t* __restrict__ cw = reinterpret_cast<const uint32_t*>(cmds + k);
        return header_of(scalar_load(cw), scalar_load(cw + 1), scalar_load(cw + 2));
    };
    // (the object's matrix is fetched where the cluster's vertex phase starts, not an iteration ahead with the header: sixteen
    // more scalars alive across a whole cluster are lane spills -- a v_readlane each -- in a loop that sits at its 102 SGPRs)
    auto mvp_of = [&](uint32_t objectId) -> Mat4 {
        Mat4 m;
        const float* __restrict__ mv = scalar_load(&kernel_args()->objFrame)[objectId].mvp;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) m.r[r][cc] = scalar_load(mv + r * 4 + cc);
        return m;
    };
    // Software pipeline over clusters (a wave's clusters k, k+1, ... are `stride` apart in the list).  While cluster k is
    // processed, k+1's header is resident, its vertex indices + triangle words are fetched after k's vertex phase, its
    // positions after k's triangle arithmetic, and k+2's header at the end -- so that at the top of an iteration the
    // positions of its first 128 vertices are already in registers (or on their way) and the chain of four dependent
    // round trips command -> records -> indices -> positions is off the critical path.
    const uint32_t stride = gridDim.x * 4u;
    uint32_t c = blockIdx.x * 4u + wave;
    if (c >= count) return;
    Header hdr = firstCmdValid ? header_of(firstCmd0, firstCmd1, firstCmd2) : load_header(c);
    Header hdrN = load_header(c + stride);
    // (the first cluster's matrix travels with its index stream: behind it, a cluster's matrix is asked for while the cluster before it
    // stores its records -- see the end of the loop)
    Mat4 mvpNext = mvp_of(hdr.objectId);
    // geometry of the current cluster: vertices lane and lane + 64 (indices, then positions), triangle words
    uint32_t t0 = 0, t1 = 0;
    float pax, pay, paz, pbx, pby, pbz;
    {
        const uint32_t* __restrict__ md = scalar_load(&kernel_args()->meshletData);
        const float* __restrict__ ps = scalar_load(&kernel_args()->positions);
        const uint32_t ia = md[hdr.dataOffset + min(lane, max(hdr.V, 1u) - 1u)] + hdr.vertexBase;
        const uint32_t ib = md[hdr.dataOffset + min(lane + 64u, max(hdr.V, 1u) - 1u)] + hdr.vertexBase;
        if (lane < hdr.T) t0 = md[hdr.dataOffset + hdr.V + lane];
        if (lane + 64u < hdr.T) t1 = md[hdr.dataOffset + hdr.V + 64u + lane];
        const float* __restrict__ pa = ps + (size_t)(ia * 3u);      // (32-bit: chordvis_upload_scene refuses scenes whose vertex index x 3 would not fit)
        const float* __restrict__ pb = ps + (size_t)(ib * 3u);
        pax = pa[0]; pay = pa[1]; paz = pa[2]; pbx = pb[0]; pby = pb[1]; pbz = pb[2];
    }
    const bool sprof = RASTER_PROFILE && (p.debug & DBG_SETUP_CLOCKS) != 0;
    unsigned long long sph[5] = {0, 0, 0, 0, 0}, stp = sprof ? wall_clock64() : 0ull;
#define SPHASE(i) do { if (sprof) { const unsigned long long tn = wall_clock64(); sph[i] += tn - stp; stp = tn; } } while (0)
    const uint32_t laneTop = lane;
    for (; c < count; c += stride) {
        // (the lane index of this cluster goes through an empty asm: what the body derives from it is invariant over the cluster
        // loop and would otherwise be hoisted out of it and held in registers across the whole kernel -- raster_tile_kernel: 25 VGPRs)
        uint32_t lane = laneTop;
        asm volatile("" : "+v"(lane));
        const uint32_t slot = hdr.slot, V = hdr.V, T = hdr.T, dataOffset = hdr.dataOffset, vertexBase = hdr.vertexBase;
        const bool twoSided = hdr.twoSided || p.depthOnly != 0u;                       // depth passes: cull mode NONE (mesh_raster.cpp:188-190)
        const bool masked = MASKED && CHORD_MATFLAG_ALPHA(hdr.matFlags) == CHORD_ALPHA_MASK;      // (wave-uniform)
        const Mat4 mvp = mvpNext;
        const uint32_t triWord[2] = {t0, t1};

        if (sprof) { volatile uint32_t sink = V + T; (void)sink; }
        SPHASE(0);
        // ---- vertex phase: position stream -> clip space -> LDS (mesh_raster.hlsl:84-105) ----------------
        bool notFast = false;
        auto vertex = [&](uint32_t i, float x, float y, float z) {
            const f4 h = mul_mv(mvp, x, y, z, 1.0f);                             // mesh_raster.hlsl:99
            const float aw = fabsf(h.w);
            lX[i] = h.x; lY[i] = h.y; lW[i] = h.w;
            lU[i] = h.x / aw * 0.5f + 0.5f;                                      // :159-161
            lV[i] = h.y / aw * -0.5f + 0.5f;
            const bool fast = p.depthClamp ? in_fast_volume_xy(h) : in_fast_volume(h);
            lD[i] = fast ? h.z / h.w : __builtin_nanf("");
            notFast = notFast || !fast;
        };
        if (lane < V) vertex(lane, pax, pay, paz);
        if (lane + 64u < V) vertex(lane + 64u, pbx, pby, pbz);
        for (uint32_t i = lane + 128u; i < V; i += 64u) {                        // (meshlets with more than 128 vertices)
            const float* __restrict__ pp = scalar_load(&kernel_args()->positions) + (size_t)(scalar_load(&kernel_args()->meshletData)[dataOffset + i] + vertexBase) * 3;
            vertex(i, pp[0], pp[1], pp[2]);
        }
        const bool allFast = __ballot(notFast) == 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        SPHASE(1);
        // next cluster: indices and triangle words now (its header has been resident for an iteration)
        const uint32_t* __restrict__ mdN = scalar_load(&kernel_args()->meshletData);
        const uint32_t nia = mdN[hdrN.dataOffset + min(lane, max(hdrN.V, 1u) - 1u)] + hdrN.vertexBase;
        const uint32_t nib = mdN[hdrN.dataOffset + min(lane + 64u, max(hdrN.V, 1u) - 1u)] + hdrN.vertexBase;
        const uint32_t nt0 = lane < hdrN.T ? mdN[hdrN.dataOffset + hdrN.V + lane] : 0u;
        const uint32_t nt1 = lane + 64u < hdrN.T ? mdN[hdrN.dataOffset + hdrN.V + 64u + lane] : 0u;

        // ---- triangle phase: the (up to) two triangles of a lane are evaluated first, then emitted together so
        //      that every round of list / bin reservations costs ONE atomic round trip for both ------------------
        // (masked clusters: the texture coordinates of each lane's two triangles are fetched now -- vertex index, then uv: two
        // dependent gathers -- so that they arrive behind the culls and the set-up instead of being waited for in the emission)
        float uvA[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, uvB[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (MASKED && masked && scalar_load(&kernel_args()->texcoords) != nullptr) {
            const float* __restrict__ tc = scalar_load(&kernel_args()->texcoords);
            const uint32_t* __restrict__ mdT = scalar_load(&kernel_args()->meshletData);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                if (lane < T) { const size_t vi = (size_t)(mdT[dataOffset + ((t0 >> (8 * i)) & 0xFFu)] + vertexBase) * 2; uvA[2 * i] = tc[vi]; uvA[2 * i + 1] = tc[vi + 1]; }
                if (lane + 64u < T) { const size_t vi = (size_t)(mdT[dataOffset + ((t1 >> (8 * i)) & 0xFFu)] + vertexBase) * 2; uvB[2 * i] = tc[vi]; uvB[2 * i + 1] = tc[vi + 1]; }
            }
        }
        int kindA = K_NONE, kindB = K_NONE;
        TriSetup tsA, tsB;
        float dA[3] = {0.0f, 0.0f, 0.0f}, dB[3] = {0.0f, 0.0f, 0.0f};
        tsA.px0 = tsA.py0 = tsA.px1 = tsA.py1 = 0; tsB.px0 = tsB.py0 = tsB.px1 = tsB.py1 = 0;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const uint32_t t = (uint32_t)half * 64u + lane;
            int kind = K_NONE;
            TriSetup ts;
            ts.px0 = ts.py0 = ts.px1 = ts.py1 = 0;
            float d[3] = {0.0f, 0.0f, 0.0f};
            if (t < T) {
                const uint32_t packedIdx = triWord[half];
                const uint32_t i0 = packedIdx & 0xFFu, i1 = (packedIdx >> 8) & 0xFFu, i2 = (packedIdx >> 16) & 0xFFu;
                const float x0 = lX[i0], y0 = lY[i0], w0 = lW[i0];
                const float x1 = lX[i1], y1 = lY[i1], w1 = lW[i1];
                const float x2 = lX[i2], y2 = lY[i2], w2 = lW[i2];
                bool culled = false;
                if (!twoSided) {                                                  // #0 mesh_raster.hlsl:143-149
                    const float det = (x0 * (y1 * w2 - w1 * y2) - y0 * (x1 * w2 - w1 * x2)) + w0 * (x1 * y2 - y1 * x2);
                    culled = det <= 0.0f;
                }
                culled = culled || (w0 <= 0.0f && w1 <= 0.0f && w2 <= 0.0f);     // #1 :152-155
                const float u0 = lU[i0], v0 = lV[i0], u1 = lU[i1], v1 = lV[i1], u2 = lU[i2], v2 = lV[i2];
                const float maxU = fmaxf(u0, fmaxf(u1, u2)), maxV = fmaxf(v0, fmaxf(v1, v2));
                const float minU = fminf(u0, fminf(u1, u2)), minV = fminf(v0, fminf(v1, v2));
                culled = culled || ((minU >= 1.0f || minV >= 1.0f) || (maxU <= 0.0f || maxV <= 0.0f));   // #2 :168-171
                culled = culled || (rintf(minU * p.W) == rintf(maxU * p.W) || rintf(minV * p.H) == rintf(maxV * p.H)); // #3 :174-179
                if (!culled) {
                    d[0] = lD[i0]; d[1] = lD[i1]; d[2] = lD[i2];
                    ts.payload = p.depthOnly ? 0u : encode_triangle_instance(t, slot);   // PASS_TYPE_DEPTH writes no id
                    if (!allFast && (d[0] != d[0] || d[1] != d[1] || d[2] != d[2])) {
                        kind = K_CLIP;
                    } else {
                        // (snapped per use: keeping the snapped pair per vertex in LDS, as the block kernel does, makes the workgroup
                        // 32 KB -- this kernel holds 256 vertices per wave -- and costs it its fifth wave per SIMD: config 4's set-up
                        // 154 -> 227 us per frame, measured)
                        ts.X[0] = (int32_t)rintf((u0 * p.W) * 256.0f); ts.Y[0] = (int32_t)rintf((v0 * p.H) * 256.0f);
                        ts.X[1] = (int32_t)rintf((u1 * p.W) * 256.0f); ts.Y[1] = (int32_t)rintf((v1 * p.H) * 256.0f);
                        ts.X[2] = (int32_t)rintf((u2 * p.W) * 256.0f); ts.Y[2] = (int32_t)rintf((v2 * p.H) * 256.0f);
                        if (tri_setup(ts, twoSided, p.Wi, p.Hi) && owns_rect(p.shard, ts.px0, ts.py0, ts.px1, ts.py1)) {
                            kind = K_EMIT;
                            if (p.biasConst != 0.0f || p.biasSlope != 0.0f) { const float o = depth_bias(ts, d, p.biasConst, p.biasSlope); d[0] += o; d[1] += o; d[2] += o; }
                        }
                    }
                }
            }
            if (ABL(p, DBG_NO_BIN)) kind = K_NONE;
            if (half == 0) { kindA = kind; tsA = ts; dA[0] = d[0]; dA[1] = d[1]; dA[2] = d[2]; }
            else           { kindB = kind; tsB = ts; dB[0] = d[0]; dB[1] = d[1]; dB[2] = d[2]; }
        }
        SPHASE(2);
        // next cluster: its positions now (the indices have arrived behind the triangle arithmetic)
        const float* __restrict__ psN = scalar_load(&kernel_args()->positions);
        const float* __restrict__ npa = psN + (size_t)(nia * 3u);
        const float* __restrict__ npb = psN + (size_t)(nib * 3u);
        const float nax = npa[0], nay = npa[1], naz = npa[2], nbx = npb[0], nby = npb[1], nbz = npb[2];
        {
            const unsigned long long lt = (1ull << lane) - 1ull;
            const unsigned long long cmA = __ballot(kindA == K_CLIP), cmB = __ballot(kindB == K_CLIP);
            const unsigned long long emA = __ballot(kindA == K_EMIT), emB = __ballot(kindB == K_EMIT);
            // the 32-byte record form takes every triangle whose vertices are at most 64 px apart; the rest go wide
            // (a masked triangle takes a 48-byte record and its extension: three slots of the wide list)
            const bool cpA = kindA == K_EMIT && !masked && fits_compact(tsA), cpB = kindB == K_EMIT && !masked && fits_compact(tsB);
            const uint32_t wSlots = masked ? 1u + CHORD_MASK_EXT_SLOTS : 1u;
            const unsigned long long ecA = __ballot(cpA), ecB = __ballot(cpB);
            const unsigned long long ewA = emA & ~ecA, ewB = emB & ~ecB;
            const bool lgA = kindA == K_EMIT && touches_many_tiles(tsA), lgB = kindB == K_EMIT && touches_many_tiles(tsB);
            const unsigned long long lmA = __ballot(lgA), lmB = __ballot(lgB);
            const uint32_t nClip = (uint32_t)(__popcll(cmA) + __popcll(cmB));
            const uint32_t nEc = (uint32_t)(__popcll(ecA) + __popcll(ecB)), nEw = (uint32_t)(__popcll(ewA) + __popcll(ewB));
            const uint32_t nLg = (uint32_t)(__popcll(lmA) + __popcll(lmB));
            {
            const RecordEmitParams e = load_record_emit_params();
            // every reservation of the cluster travels together: the list reservations (lane 0) and the bin
            // reservations; nothing is stored before they are back
            uint32_t cbase = 0, ebaseC = 0, ebaseW = 0, lbase = 0;
            if (lane == 0) {
                if (nClip) cbase = atomicAdd(&e.counters->clipTriCount[p.pass], nClip);
                if (nEc) ebaseC = atomicAdd(&e.counters->triCountC[listShard * CHORD_SHARD_STRIDE], nEc);
                if (nEw) ebaseW = atomicAdd(&e.counters->triCount[listShard * CHORD_SHARD_STRIDE], nEw * wSlots);
                if (nLg) lbase = atomicAdd(&e.counters->largeCount[p.pass][listShard * CHORD_SHARD_STRIDE], nLg);
            }
            BinTicket ticket;
            if (emA | emB) wave_bin_issue(e, kindA == K_EMIT && !lgA, tsA, kindB == K_EMIT && !lgB, tsB, lane, ticket, MASKED && masked);
            cbase = bcast(cbase, 0); ebaseC = bcast(ebaseC, 0); ebaseW = bcast(ebaseW, 0); lbase = bcast(lbase, 0);
            SPHASE(3);
            mvpNext = mvp_of(hdrN.objectId);              // (the next cluster's matrix: on its way while this one's records are stored)
            if (kindA == K_CLIP) {
                const uint32_t k = cbase + (uint32_t)__popcll(cmA & lt);
                if (k < e.clipTriCap) { ClipTri ct; ct.objectId = hdr.objectId; ct.meshletId = hdr.meshletId; ct.slot = hdr.slot; ct.tri = lane; e.clipTris[k] = ct; }
                else atomicOr(&e.counters->overflow, 2u);
            }
            if (kindB == K_CLIP) {
                const uint32_t k = cbase + (uint32_t)__popcll(cmA) + (uint32_t)__popcll(cmB & lt);
                if (k < e.clipTriCap) { ClipTri ct; ct.objectId = hdr.objectId; ct.meshletId = hdr.meshletId; ct.slot = hdr.slot; ct.tri = lane + 64u; e.clipTris[k] = ct; }
                else atomicOr(&e.counters->overflow, 2u);
            }
            // giX names the record in a bin: compact index, or CHORD_REC_WIDE | wide index
            uint32_t giA = 0, giB = 0;
            bool okA = false, okB = false;
            if (cpA) {
                const uint32_t li = ebaseC + (uint32_t)__popcll(ecA & lt);
                if (li < e.triCapC) { giA = listShard * e.triCapC + li; write_record_c(&e.trisC[giA], tsA, dA); okA = true; }
                else atomicOr(&e.counters->overflow, 1u);
            } else if (kindA == K_EMIT) {
                const uint32_t li = ebaseW + wSlots * (uint32_t)__popcll(ewA & lt);
                if (li + wSlots <= e.triCap) {
                    giA = listShard * e.triCap + li; write_record(&e.tris[giA], tsA, dA, twoSided, masked);
                    if (MASKED && masked) setup_emit_mask_ext(p, &e.tris[giA + 1u], triWord[0], uvA, lW, CHORD_MATFLAG_MATERIAL(hdr.matFlags), tsA.area);
                    giA |= (MASKED && masked) ? CHORD_REC_MASKED : CHORD_REC_WIDE; okA = true;
                } else atomicOr(&e.counters->overflow, 1u);
            }
            if (cpB) {
                const uint32_t li = ebaseC + (uint32_t)__popcll(ecA) + (uint32_t)__popcll(ecB & lt);
                if (li < e.triCapC) { giB = listShard * e.triCapC + li; write_record_c(&e.trisC[giB], tsB, dB); okB = true; }
                else atomicOr(&e.counters->overflow, 1u);
            } else if (kindB == K_EMIT) {
                const uint32_t li = ebaseW + wSlots * ((uint32_t)__popcll(ewA) + (uint32_t)__popcll(ewB & lt));
                if (li + wSlots <= e.triCap) {
                    giB = listShard * e.triCap + li; write_record(&e.tris[giB], tsB, dB, twoSided, masked);
                    if (MASKED && masked) setup_emit_mask_ext(p, &e.tris[giB + 1u], triWord[1], uvB, lW, CHORD_MATFLAG_MATERIAL(hdr.matFlags), tsB.area);
                    giB |= (MASKED && masked) ? CHORD_REC_MASKED : CHORD_REC_WIDE; okB = true;
                } else atomicOr(&e.counters->overflow, 1u);
            }
            // <= 2x2 tiles: straight into the bins; more: the large list
            if (emA | emB) wave_bin_commit(e, ticket, okA, giA, okB, giB);
            if (lgA && okA) {
                const uint32_t k = lbase + (uint32_t)__popcll(lmA & lt);
                if (k < e.largeCap) e.largeList[(size_t)listShard * e.largeCap + k] = giA & CHORD_REC_WIDE_INDEX; else atomicOr(&e.counters->overflow, 1u);
            }
            if (lgB && okB) {
                const uint32_t k = lbase + (uint32_t)__popcll(lmA) + (uint32_t)__popcll(lmB & lt);
                if (k < e.largeCap) e.largeList[(size_t)listShard * e.largeCap + k] = giB & CHORD_REC_WIDE_INDEX; else atomicOr(&e.counters->overflow, 1u);
            }
            }
        }
        // LDS of this wave is rewritten by the next cluster: order the reads above before those writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // rotate the pipeline; the header after next is fetched now and first used after the next vertex phase
        hdr = hdrN;
        hdrN = load_header(c + 2u * stride);
        t0 = nt0; t1 = nt1;
        pax = nax; pay = nay; paz = naz; pbx = nbx; pby = nby; pbz = nbz;
        SPHASE(4);
    }
    if (sprof && lane == 0u) {
        const uint32_t w = blockIdx.x * 4u + wave;
        if (w < CHORD_MAX_TILES * 8u / 5u) for (int i = 0; i < 5; i++) p.tilePhase[(size_t)w * 5u + i] = sph[i];
    }
#undef SPHASE
}

// ---- the BLOCKS body, round 3: the same clusters -> the same blocks, organised around registers --------------------------
// Round 2's BLOCKS instantiation of raster_setup_body wanted 159 VGPRs and ran at the kernel's 128 with 22 of them in scratch
// (92 bytes per lane, re-read per cluster: the 21 GB of WRITE_SIZE the round-2 profile of BASELINE config 5 could not account
// for -- tools/microbench/write_size_calib shows the block stores themselves are counted at face value).  What it kept alive
// across the block code was the whole set-up of BOTH triangles of a lane (snapped vertices, depths, areas, bounds: 2 x 19
// registers) next to the software pipeline of the next cluster.  This body keeps, per triangle, a 2-bit kind and its pixel
// bounds (two packed registers) between the classification and the resolve, and sets a triangle up again from the wave's LDS
// copy of the vertices right before it is scan-converted, one triangle at a time.  The price is the snapping and the area of
// an emitted triangle twice (~45 VALU instructions per triangle of ~600); the division moves, it is not repeated.
// Wave-uniform records (draw command, meshlet header, object matrix) come through the scalar cache (constant address space:
// s_load instead of a vector load + v_readfirstlane per dword), and a block leaves the wave as 16-byte stores of consecutive
// word pairs (header | word 0, word 1 | word 2, ...): half the store instructions, whole 16-byte granules.
struct SetupHeader { uint32_t objectId, meshletId, slot, V, T, dataOffset, vertexBase, matFlags; };

// What the block kernel needs only when a cluster's blocks are written (a few lanes, once per cluster).  Kept in scalar
// registers across the whole loop these 22 dwords push the kernel past its 102 SGPRs, and every scalar the compiler parks in a
// VGPR lane costs a VALU instruction each way (~300 v_readlane / v_writelane in the loop body: 15 % of its VALU work).  They
// are re-read from the kernel-argument segment (scalar cache) right where they are used instead; the pointer goes through an
// empty asm so that the loads cannot be hoisted back out of the loop.
struct BlockEmitParams {
    uint32_t* tileCount; uint32_t* tileBins; uint32_t binCap;
    uint32_t* binPool; uint32_t binPoolChunks; uint32_t* binPoolCount;
    unsigned long long* binChunkTab; uint32_t binStamp; uint32_t binMaxChunks;
    unsigned long long* blockPool; uint32_t blockCap;
    DeviceCounters* counters;
};
__device__ __forceinline__ BlockEmitParams load_block_emit_params()
{
    const RasterParams* q = kernel_args();
    BlockEmitParams e;
    e.tileCount = scalar_load(&q->tileCount); e.tileBins = scalar_load(&q->tileBins); e.binCap = scalar_load(&q->binCap);
    e.binPool = scalar_load(&q->binPool); e.binPoolChunks = scalar_load(&q->binPoolChunks); e.binPoolCount = scalar_load(&q->binPoolCount);
    e.binChunkTab = scalar_load(&q->binChunkTab); e.binStamp = scalar_load(&q->binStamp); e.binMaxChunks = scalar_load(&q->binMaxChunks);
    e.blockPool = scalar_load(&q->blockPool); e.blockCap = scalar_load(&q->blockCap);
    e.counters = scalar_load(&q->counters);
    return e;
}

#define WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// classification of one triangle of the cluster in LDS: kind, and for K_EMIT its pixel bounds (x0 | x1 << 16, y0 | y1 << 16)
// and whether it is narrow (fits_compact).  The same culls and the same integers as cluster_records / raster_setup_body.
__device__ __forceinline__ int classify_triangle(const RasterParams& p, uint32_t t, uint32_t T, uint32_t packedIdx, bool twoSided, bool allFast,
                                                 const float* lX, const float* lY, const float* lW, const float* lU, const float* lV, const float* lD,
                                                 const int32_t* lSX, const int32_t* lSY, uint32_t& boxX, uint32_t& boxY, bool& narrow)
{
    boxX = 0u; boxY = 0u; narrow = false;
    if (t >= T) return K_NONE;
    const uint32_t i0 = packedIdx & 0xFFu, i1 = (packedIdx >> 8) & 0xFFu, i2 = (packedIdx >> 16) & 0xFFu;
    const float w0 = lW[i0], w1 = lW[i1], w2 = lW[i2];
    bool culled = false;
    if (!twoSided) {                                                  // #0 mesh_raster.hlsl:143-149
        const float x0 = lX[i0], y0 = lY[i0], x1 = lX[i1], y1 = lY[i1], x2 = lX[i2], y2 = lY[i2];
        const float det = (x0 * (y1 * w2 - w1 * y2) - y0 * (x1 * w2 - w1 * x2)) + w0 * (x1 * y2 - y1 * x2);
        culled = det <= 0.0f;
    }
    culled = culled || (w0 <= 0.0f && w1 <= 0.0f && w2 <= 0.0f);     // #1 :152-155
    const float u0 = lU[i0], v0 = lV[i0], u1 = lU[i1], v1 = lV[i1], u2 = lU[i2], v2 = lV[i2];
    const float maxU = fmaxf(u0, fmaxf(u1, u2)), maxV = fmaxf(v0, fmaxf(v1, v2));
    const float minU = fminf(u0, fminf(u1, u2)), minV = fminf(v0, fminf(v1, v2));
    culled = culled || ((minU >= 1.0f || minV >= 1.0f) || (maxU <= 0.0f || maxV <= 0.0f));   // #2 :168-171
    culled = culled || (rintf(minU * p.W) == rintf(maxU * p.W) || rintf(minV * p.H) == rintf(maxV * p.H)); // #3 :174-179
    if (culled) return K_NONE;
    if (!allFast) { const float d0 = lD[i0], d1 = lD[i1], d2 = lD[i2]; if (d0 != d0 || d1 != d1 || d2 != d2) return K_CLIP; }
    // (the snapped 24.8 coordinates are per-vertex values: computed once per vertex in the vertex phase, not once per use --
    // a cluster's 128 triangles name its 81 vertices 384 times, here and again in the resolve)
    const int32_t X0 = lSX[i0], Y0 = lSY[i0], X1 = lSX[i1], Y1 = lSY[i1], X2 = lSX[i2], Y2 = lSY[i2];
    const int32_t minX = min(X0, min(X1, X2)), maxX = max(X0, max(X1, X2));
    const int32_t minY = min(Y0, min(Y1, Y2)), maxY = max(Y0, max(Y1, Y2));
    // a triangle wider than 64 px is nothing a block holds: the cluster goes to the record kernel whatever else is true of
    // it (emitted and not narrow; the record kernel culls it or sets it up) -- and what is left here has deltas below 2^15:
    // the area is two full-rate 24-bit multiplies (the general 64-bit form is four quarter-rate ones, evaluated for every
    // triangle when it sits behind a select)
    if (maxX - minX > (1 << 14) || maxY - minY > (1 << 14)) return K_EMIT;
    narrow = true;
    const int32_t area2 = __mul24(X1 - X0, Y2 - Y0) - __mul24(X2 - X0, Y1 - Y0);
    if (area2 == 0 || (!twoSided && area2 > 0)) return K_NONE;
    // (sharded frames: ownership is decided per window part, below -- a cluster that straddles two ranks' tiles is small)
    const int32_t px0 = max(0, (minX + 127) >> 8), py0 = max(0, (minY + 127) >> 8);
    const int32_t px1 = min(p.Wi - 1, (maxX - 128) >> 8), py1 = min(p.Hi - 1, (maxY - 128) >> 8);
    if (px1 < px0 || py1 < py0) return K_NONE;
    boxX = (uint32_t)px0 | ((uint32_t)px1 << 16); boxY = (uint32_t)py0 | ((uint32_t)py1 << 16);
    return K_EMIT;
}

// set-up of an emitted, narrow triangle again from LDS and its scan conversion into the wave's pixel window.  `entry` is a word
// of the cluster's compacted list of emitted triangles (vertex indices in the low 24 bits, triangle number above), boxX / boxY
// the pixel bounds its classification found: the emitted triangles of a cluster -- 59 of 128 on BASELINE config 5 -- are
// resolved one per lane in one pass (a second one only when there are more than 64), not in two passes of a lane's own two
// triangles with half the lanes idle in each.  The classification accepted exactly these integers, and only narrow triangles
// reach a block: deltas below 2^15, the area from two 24-bit multiplies, its float an int32 conversion.
__device__ __forceinline__ void resolve_entry(const RasterParams& p, uint32_t entry, uint32_t boxX, uint32_t boxY, uint32_t slot,
                                              const int32_t* lSX, const int32_t* lSY, const float* lD, unsigned long long* win,
                                              int32_t bx0, int32_t by0)
{
    const uint32_t i0 = entry & 0xFFu, i1 = (entry >> 8) & 0xFFu, i2 = (entry >> 16) & 0xFFu, t = entry >> 24;
    TriSetup ts;
    ts.X[0] = lSX[i0]; ts.Y[0] = lSY[i0];
    ts.X[1] = lSX[i1]; ts.Y[1] = lSY[i1];
    ts.X[2] = lSX[i2]; ts.Y[2] = lSY[i2];
    const int32_t dx1 = ts.X[1] - ts.X[0], dy1 = ts.Y[1] - ts.Y[0], dx2 = ts.X[2] - ts.X[0], dy2 = ts.Y[2] - ts.Y[0];
    const int32_t area2 = __mul24(dx1, dy2) - __mul24(dx2, dy1);
    const int32_t area = area2 < 0 ? -area2 : area2;
    ts.s = area2 < 0 ? -1 : 1;
    ts.area = (int64_t)area;
    ts.invA = 1.0f / (float)area;
    ts.px0 = (int32_t)(boxX & 0xFFFFu); ts.px1 = (int32_t)(boxX >> 16);
    ts.py0 = (int32_t)(boxY & 0xFFFFu); ts.py1 = (int32_t)(boxY >> 16);
    float d[3] = {lD[i0], lD[i1], lD[i2]};
    if (p.biasConst != 0.0f || p.biasSlope != 0.0f) { const float o = depth_bias(ts, d, p.biasConst, p.biasSlope); d[0] += o; d[1] += o; d[2] += o; }
    ts.payload = p.depthOnly ? 0u : encode_triangle_instance(t, slot);
    ts.d0 = d[0]; ts.e1 = d[1] - d[0]; ts.e2 = d[2] - d[0];
    // (two copies of the pixel loop rather than a clamp + select per pixel that main-view frames never need)
    if (p.depthClamp != 0u) tile_raster_narrow<WIN>(win, ts, bx0, by0, ts.px0, ts.py0, ts.px1, ts.py1, false, true);
    else tile_raster_narrow<WIN>(win, ts, bx0, by0, ts.px0, ts.py0, ts.px1, ts.py1, false, false);
}

// Hot tiles (BASELINE config 5, variant "hotspot": one screen tile receives 12 % of the frame's 10 M blocks): every bin slot
// is a returning atomic on the tile's counter line, and a line retires ~88 of them per microsecond -- 1.26 M slots of the
// hottest tile are 14 ms of a 17.5-ms kernel spent in one queue.  Once a tile's bin holds SLOT_HOT entries, a wave that draws
// from it takes SLOT_AHEAD slots with one atomic and serves its next clusters in that tile from the reserve (a per-wave,
// direct-mapped table in LDS); what is left of a reserve when the wave ends is filled with the "no entry" word the tile
// kernel skips.  Tiles below the threshold -- every tile of every other workload -- never enter the table.
#define SLOT_CACHE 32u
#define SLOT_HOT 65536u
#ifndef SLOT_AHEAD
#define SLOT_AHEAD 8u
#endif
// ... and four times as many once the bin is a quarter of a million entries long: a rank of a sharded frame that owns such a tile
// owns little else, every one of its waves draws from that one counter line, and the queue on the line is what the kernel
// waits for (BASELINE config 5's hotspot at 8 ranks, the rank with the 1.34 M-entry tile: 7.6 -> 6.6 ms per frame; drawing 32
// ahead everywhere costs the one-GPU frame 4 %: every wave leaves up to 31 unused slots in each of the ~40 hot tiles it touched)
#ifndef SLOT_AHEAD_VERY_HOT
#define SLOT_AHEAD_VERY_HOT 32u
#endif
#define SLOT_VERY_HOT 262144u
struct SlotCache { uint32_t key[SLOT_CACHE], next[SLOT_CACHE], end[SLOT_CACHE]; uint32_t pend; };   // key = tile + 1 (0: free), slots [next, end) in reserve; pend: lane 0's draw in flight

#ifndef BLOCKS_LDS_VERTS
#define BLOCKS_LDS_VERTS 128    // vertices per cluster the block kernel takes (larger clusters are left to the record kernel): 12 + 8 KB of LDS per workgroup
#endif
template <bool HOT>
__device__ __forceinline__ void raster_setup_blocks_body(const RasterParams& p, const uint32_t count, float (*sVert)[4][BLOCKS_LDS_VERTS], int32_t (*sSnap)[4][BLOCKS_LDS_VERTS],
                                                         unsigned long long (*sWin)[WIN * WIN], SlotCache* slotCaches)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    SlotCache& sc = slotCaches[HOT ? wave : 0u];
    if (HOT && lane < SLOT_CACHE) { sc.key[lane] = 0u; sc.next[lane] = 0u; sc.end[lane] = 0u; }
    if (HOT) {
        // the tiles that were hot in the last frame are hot from this launch's first cluster on: a tile used to become hot only once its
        // bin held SLOT_HOT entries IN THIS LAUNCH, i.e. every frame began with 65 536 single draws from one counter line (~0.75 ms
        // at the ~88 returning atomics per microsecond a line sustains) and 24 k draws of eight before the draws of 32 -- with every wave of
        // the rank that owns the tile queued on it.  An entry with an empty reserve at position p draws ahead on its first use, 32
        // slots when p >= SLOT_VERY_HOT; a tile that has cooled down since costs a few "no entry" words at the end of the launch.
        const uint32_t* __restrict__ hl = scalar_load(&kernel_args()->hotTiles);
        WAVE_LDS_SYNC();
        if (hl) {
            const uint32_t nHot = min(hl[0], (uint32_t)CHORD_HOT_TILES);
            if (lane < nHot) {
                const uint32_t e = hl[1u + lane], tile = e & 0xFFFu, ci = tile & (SLOT_CACHE - 1u);
                const uint32_t pos = (e >> 31) ? SLOT_VERY_HOT : 0u;
                sc.key[ci] = tile + 1u; sc.next[ci] = pos; sc.end[ci] = pos;           // (two hot tiles on one entry: either wins)
            }
        }
        WAVE_LDS_SYNC();
    }
    float* lX = sVert[0][wave]; float* lY = sVert[1][wave]; float* lW = sVert[2][wave];
    float* lU = sVert[3][wave]; float* lV = sVert[4][wave]; float* lD = sVert[5][wave];
    int32_t* lSX = sSnap[0][wave]; int32_t* lSY = sSnap[1][wave];
    unsigned long long* win = sWin[wave];
    const uint32_t listShard = (blockIdx.x * 4u + wave) % CHORD_LIST_SHARDS;

    // the scene pointers are re-read from the kernel-argument segment where they are used (kernel_args; keeping them in
    // registers -- the "hot parameters" variant of profiles/r03_block_kernel_variants.txt -- cost 3 %)
    auto kq = [&]() -> const RasterParams* { return kernel_args(); };
    auto header_at = [&](uint32_t i) -> SetupHeader {
        const uint32_t k = __builtin_amdgcn_readfirstlane(min(i, count - 1u));
        SetupHeader h;
        const RasterParams* q = kq();
        const uint32_t* __restrict__ cw = reinterpret_cast<const uint32_t*>(scalar_load(&q->cmds) + k);
        h.objectId = scalar_load(cw); h.meshletId = scalar_load(cw + 1); h.slot = scalar_load(cw + 2);
        const DMeshlet* __restrict__ mm = &scalar_load(&q->meshlets)[h.meshletId];
        const uint32_t vt = scalar_load(&mm->vertexTriangleCount);
        h.V = vt & 0xFFu; h.T = (vt >> 8) & 0xFFu;
        h.dataOffset = scalar_load(&mm->dataOffset);
        h.vertexBase = scalar_load(&mm->vertexBase);
        h.matFlags = scalar_load(&scalar_load(&q->objStatic)[h.objectId].matFlags);
        if (CHORD_MATFLAG_ALPHA(h.matFlags) >= CHORD_ALPHA_BLEND) h.T = 0u;   // blended: in no bucket of renderMesh (mesh_raster.cpp:224)
        return h;
    };
    auto mvp_of = [&](uint32_t objectId) -> Mat4 {
        Mat4 m;
        const float* __restrict__ mv = scalar_load(&kq()->objFrame)[objectId].mvp;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) m.r[r][cc] = scalar_load(mv + r * 4 + cc);
        return m;
    };
    const uint32_t stride = gridDim.x * 4u;
    uint32_t c = blockIdx.x * 4u + wave;
    if (c >= count) return;
    SetupHeader hdr = header_at(c);
    SetupHeader hdrN = header_at(c + stride);
    uint32_t t0 = 0, t1 = 0;
    float pax, pay, paz, pbx, pby, pbz;
    {
        const uint32_t* __restrict__ md = scalar_load(&kq()->meshletData);
        const float* __restrict__ ps = scalar_load(&kq()->positions);
        const uint32_t ia = md[hdr.dataOffset + min(lane, max(hdr.V, 1u) - 1u)] + hdr.vertexBase;
        const uint32_t ib = md[hdr.dataOffset + min(lane + 64u, max(hdr.V, 1u) - 1u)] + hdr.vertexBase;
        if (lane < hdr.T) t0 = md[hdr.dataOffset + hdr.V + lane];
        if (lane + 64u < hdr.T) t1 = md[hdr.dataOffset + hdr.V + 64u + lane];
        const float* __restrict__ pa = ps + (size_t)(ia * 3u);      // (32-bit: chordvis_upload_scene refuses scenes whose vertex index x 3 would not fit)
        const float* __restrict__ pb = ps + (size_t)(ib * 3u);
        pax = pa[0]; pay = pa[1]; paz = pa[2]; pbx = pb[0]; pby = pb[1]; pbz = pb[2];
    }
    const bool sprof = RASTER_PROFILE && (p.debug & DBG_SETUP_CLOCKS) != 0;
    unsigned long long sph[5] = {0, 0, 0, 0, 0}, stp = sprof ? wall_clock64() : 0ull;
#define SPHASE(i) do { if (sprof) { const unsigned long long tn = wall_clock64(); sph[i] += tn - stp; stp = tn; } } while (0)
    // (the per-cluster opaque lane index of raster_setup_body was measured here too: 78 -> 76 VGPRs, but 928 -> 946 VALU wave-instructions
    // per cluster -- this kernel is bound by VALU issue, and what was hoisted out of its loop was arithmetic it now repeats; not kept)
    for (; c < count; c += stride) {
        const uint32_t V = hdr.V, T = hdr.T, dataOffset = hdr.dataOffset, vertexBase = hdr.vertexBase;
        const bool twoSided = (hdr.matFlags & CHORD_MATFLAG_TWO_SIDED) != 0u || p.depthOnly != 0u;
        const bool masked = CHORD_MATFLAG_ALPHA(hdr.matFlags) == CHORD_ALPHA_MASK;
        if (sprof) { volatile uint32_t sink = V + T; (void)sink; }
        SPHASE(0);
        const bool tooBig = V > (uint32_t)BLOCKS_LDS_VERTS;                  // (wave-uniform) more vertices than the wave's LDS arrays hold
        (void)dataOffset; (void)vertexBase;
        // ---- vertex phase (mesh_raster.hlsl:84-105), as raster_setup_body -------------------------------------------------
        bool notFast = false;
        if (!tooBig) {
            // (fetched here, not an iteration ahead: sixteen more scalar registers alive across the whole cluster cost more
            // lane spills than the other resident waves cover of this one scalar-cache round trip)
            const Mat4 mvp = mvp_of(hdr.objectId);
            auto vertex = [&](uint32_t i, float x, float y, float z) {
                const f4 h = mul_mv(mvp, x, y, z, 1.0f);
                const float aw = fabsf(h.w);
                lX[i] = h.x; lY[i] = h.y; lW[i] = h.w;
                const float u = h.x / aw * 0.5f + 0.5f, v = h.y / aw * -0.5f + 0.5f;
                lU[i] = u; lV[i] = v;
                // snapped 24.8 coordinates of the vertex (mesh_raster setup: the values every triangle on it uses; a vertex outside
                // the guard band may overflow the conversion -- its triangles take the clipper and never read them)
                lSX[i] = (int32_t)rintf((u * p.W) * 256.0f); lSY[i] = (int32_t)rintf((v * p.H) * 256.0f);
                const bool fast = p.depthClamp ? in_fast_volume_xy(h) : in_fast_volume(h);
                lD[i] = fast ? h.z / h.w : __builtin_nanf("");
                notFast = notFast || !fast;
            };
            if (lane < V) vertex(lane, pax, pay, paz);
            if (lane + 64u < V) vertex(lane + 64u, pbx, pby, pbz);
#if BLOCKS_LDS_VERTS > 128
            for (uint32_t i = lane + 128u; i < V; i += 64u) {
                const float* __restrict__ pp = scalar_load(&kq()->positions) + (size_t)(scalar_load(&kq()->meshletData)[dataOffset + i] + vertexBase) * 3;
                vertex(i, pp[0], pp[1], pp[2]);
            }
#endif
        }
        const bool allFast = __ballot(notFast) == 0ull;
        WAVE_LDS_SYNC();
        SPHASE(1);
        // next cluster: vertex indices and triangle words
        const uint32_t* __restrict__ md = scalar_load(&kq()->meshletData);
        const uint32_t nia = md[hdrN.dataOffset + min(lane, max(hdrN.V, 1u) - 1u)] + hdrN.vertexBase;
        const uint32_t nib = md[hdrN.dataOffset + min(lane + 64u, max(hdrN.V, 1u) - 1u)] + hdrN.vertexBase;
        const uint32_t nt0 = lane < hdrN.T ? md[hdrN.dataOffset + hdrN.V + lane] : 0u;
        const uint32_t nt1 = lane + 64u < hdrN.T ? md[hdrN.dataOffset + hdrN.V + 64u + lane] : 0u;

        // ---- classification: kind + pixel bounds per triangle, nothing else survives it ----------------------------------
        uint32_t bxA, byA, bxB, byB;
        bool nwA, nwB;
        int kindA = classify_triangle(p, lane, tooBig ? 0u : T, t0, twoSided, allFast, lX, lY, lW, lU, lV, lD, lSX, lSY, bxA, byA, nwA);
        int kindB = classify_triangle(p, lane + 64u, tooBig ? 0u : T, t1, twoSided, allFast, lX, lY, lW, lU, lV, lD, lSX, lSY, bxB, byB, nwB);
        if (ABL(p, DBG_NO_BIN)) { kindA = K_NONE; kindB = K_NONE; }
        SPHASE(2);
        // next cluster: its positions (the indices have arrived behind the classification)
        const float* __restrict__ ps = scalar_load(&kq()->positions);
        const float* __restrict__ npa = ps + (size_t)(nia * 3u);
        const float* __restrict__ npb = ps + (size_t)(nib * 3u);
        const float nax = npa[0], nay = npa[1], naz = npa[2], nbx = npb[0], nby = npb[1], nbz = npb[2];

        const bool eA = kindA == K_EMIT, eB = kindB == K_EMIT;
        const unsigned long long emA = __ballot(eA), emB = __ballot(eB);
        const unsigned long long bad = __ballot(kindA == K_CLIP || kindB == K_CLIP || (eA && !nwA) || (eB && !nwB)) | (tooBig && T != 0u ? 1ull : 0ull);
        bool blocksDone = false;
        if (!masked && (emA | emB) != 0ull && bad == 0ull) {
            int32_t bx0 = min(eA ? (int32_t)(bxA & 0xFFFFu) : 0x7FFF, eB ? (int32_t)(bxB & 0xFFFFu) : 0x7FFF);
            int32_t by0 = min(eA ? (int32_t)(byA & 0xFFFFu) : 0x7FFF, eB ? (int32_t)(byB & 0xFFFFu) : 0x7FFF);
            int32_t bx1 = max(eA ? (int32_t)(bxA >> 16) : -1, eB ? (int32_t)(bxB >> 16) : -1);
            int32_t by1 = max(eA ? (int32_t)(byA >> 16) : -1, eB ? (int32_t)(byB >> 16) : -1);
            if (__ballot(bx1 - bx0 >= WIN || by1 - by0 >= WIN) == 0ull) {            // (no single lane is already too wide)
                wave_min2_max2(bx0, by0, bx1, by1);
                const int32_t bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
                const uint32_t nE = (uint32_t)(__popcll(emA) + __popcll(emB));
                if (bw <= WIN && bh <= WIN && (uint32_t)(bw * bh + 8) * 8u <= nE * 36u) {
                    blocksDone = true;
                    // the window's parts per tile (<= 2 x 2): lane r < 4 owns the part in tile (r & 1 ? tx1 : tx0, r & 2 ? ty1 : ty0)
                    const int32_t tx0 = bx0 >> TILE_SHIFT, tx1 = bx1 >> TILE_SHIFT, ty0 = by0 >> TILE_SHIFT, ty1 = by1 >> TILE_SHIFT;
                    const uint32_t r = lane & 3u;
                    const bool sx = (r & 1u) != 0u, sy = (r & 2u) != 0u;
                    const int32_t rx0 = sx ? tx1 << TILE_SHIFT : bx0, rx1 = (sx || tx1 == tx0) ? bx1 : (tx0 << TILE_SHIFT) + TILE - 1;
                    const int32_t ry0 = sy ? ty1 << TILE_SHIFT : by0, ry1 = (sy || ty1 == ty0) ? by1 : (ty0 << TILE_SHIFT) + TILE - 1;
                    const uint32_t rw = (uint32_t)(rx1 - rx0 + 1), rh = (uint32_t)(ry1 - ry0 + 1);
                    const bool has = lane < 4u && !(sx && tx1 == tx0) && !(sy && ty1 == ty0) && owns_tile(p.shard, sx ? tx1 : tx0, sy ? ty1 : ty0);
                    const uint32_t hasMask = (uint32_t)__ballot(has) & 15u;
                    const uint32_t gran = has ? (rw * rh + 2u) >> 1 : 0u;              // header + w x h words, in 16-byte granules
                    const uint32_t g0 = bcast(gran, 0), g1 = bcast(gran, 1), g2 = bcast(gran, 2), g3 = bcast(gran, 3);
                    const uint32_t before = (r > 0u ? g0 : 0u) + (r > 1u ? g1 : 0u) + (r > 2u ? g2 : 0u), G = g0 + g1 + g2 + g3;
                    const uint32_t tile = __umul24((uint32_t)(sy ? ty1 : ty0), p.tilesX) + (uint32_t)(sx ? tx1 : tx0);
                    // one round trip: pool space (lane 0) and, per touched tile, ONE 64-bit add on the tile's counter pair
                    // (low word: bin slot, high word: the tile's block count) ...
                    uint32_t gbase = 0, slot = 0;
                    const BlockEmitParams e = load_block_emit_params();
                    // (lane 0's part -- the tile of the window's first pixel -- may be served from this wave's reserve on a hot tile)
                    // (what lane 0 drew -- 0: from the reserve, else the number of slots -- waits in LDS, not in a register, for the
                    // resolve to finish)
                    {
                        uint32_t ahead = 1u;
                        if (HOT) {
                            const uint32_t ci = tile & (SLOT_CACHE - 1u);
                            if (lane == 0u && has && sc.key[ci] == tile + 1u) {
                                const uint32_t nx = sc.next[ci];
                                if (nx < sc.end[ci]) { slot = nx; sc.next[ci] = nx + 1u; ahead = 0u; }
                                else ahead = nx >= SLOT_VERY_HOT ? SLOT_AHEAD_VERY_HOT : SLOT_AHEAD;   // a hot tile whose reserve is used up: draw ahead again
                            }
                            if (lane == 0u) sc.pend = ahead;
                        }
                        if (lane == 0u && G) gbase = atomicAdd(&e.counters->blockGranules[listShard * CHORD_SHARD_STRIDE], G);
                        if (has && ahead) slot = (uint32_t)atomicAdd(reinterpret_cast<unsigned long long*>(&e.tileCount[(size_t)tile * TC_STRIDE]), (unsigned long long)ahead * 0x100000001ull);
                    }
                    // ... and the cluster is resolved while they are in flight.  The emitted triangles are compacted first (the
                    // clip-space arrays x, y, w are dead behind the classification: the list and the two bound words take their place)
                    uint32_t* lList = reinterpret_cast<uint32_t*>(lX);
                    uint32_t* lBoxX = reinterpret_cast<uint32_t*>(lY);
                    uint32_t* lBoxY = reinterpret_cast<uint32_t*>(lW);
                    const uint32_t nA = (uint32_t)__popcll(emA);
                    WAVE_LDS_SYNC();                                             // (every lane's classification has read them)
                    if (eA) { const uint32_t k = mbcnt64(emA); lList[k] = (t0 & 0xFFFFFFu) | lane << 24; lBoxX[k] = bxA; lBoxY[k] = byA; }
                    if (eB) { const uint32_t k = nA + mbcnt64(emB); lList[k] = (t1 & 0xFFFFFFu) | (lane + 64u) << 24; lBoxX[k] = bxB; lBoxY[k] = byB; }
#pragma unroll
                    for (int k = 0; k < WIN * WIN / 64; k++) win[lane + 64u * k] = 0ull;
                    WAVE_LDS_SYNC();
                    // (a sharded frame's cluster none of whose window parts is this rank's: the conservative cluster test let it through)
                    const bool anyPart = hasMask != 0u;
                    if (anyPart) {
                        if (lane < nE) resolve_entry(p, lList[lane], lBoxX[lane], lBoxY[lane], hdr.slot, lSX, lSY, lD, win, bx0, by0);
                        if (nE > 64u) {                                          // (wave-uniform; one triangle's set-up alive at a time)
                            __builtin_amdgcn_sched_barrier(0);
                            if (lane + 64u < nE) resolve_entry(p, lList[lane + 64u], lBoxX[lane + 64u], lBoxY[lane + 64u], hdr.slot, lSX, lSY, lD, win, bx0, by0);
                        }
                    }
                    WAVE_LDS_SYNC();
                    SPHASE(3);
                    gbase = bcast(gbase, 0);
                    const bool fits = gbase + G <= e.blockCap;
                    if (!fits && lane == 0u) atomicOr(&e.counters->overflow, 1u);
                    const uint32_t off = listShard * e.blockCap + gbase + before;              // granule offset of lane r's block
                    uint32_t drew = 1u;
                    if (HOT) {
                        if (lane == 0u) drew = sc.pend;
                        if (lane == 0u && has && drew) {
                            const uint32_t ci = tile & (SLOT_CACHE - 1u);
                            if (drew > 1u) {
                                sc.next[ci] = slot + 1u; sc.end[ci] = slot + drew;
                                for (uint32_t k = 1u; k < drew; k++) bin_alloc(e, tile, slot + k);   // (every drawn slot, before the first store)
                            } else if (slot >= scalar_load(&kq()->slotHot) && (sc.key[ci] == 0u || sc.next[ci] >= sc.end[ci])) {
                                sc.key[ci] = tile + 1u; sc.next[ci] = 0u; sc.end[ci] = 0u;           // hot from now on (an entry with a live reserve is never replaced)
                            }
                        }
                    }
                    if (has && drew) bin_alloc(e, tile, slot);
                    if (has && fits) bin_put(e, tile, slot, CHORD_REC_BLOCK | off);
                    if (fits) {
                        for (uint32_t q = 0; q < 4u; q++) {                      // (wave-uniform: the parts that exist, one in 4 of 5 clusters)
                            if (!((hasMask >> q) & 1u)) continue;
                            const uint32_t qoff = bcast(off, (int)q), qw = bcast(rw, (int)q), qh = bcast(rh, (int)q);
                            const uint32_t qx = (uint32_t)bcast(rx0 - bx0, (int)q), qy = (uint32_t)bcast(ry0 - by0, (int)q);
                            const uint32_t qlx = (uint32_t)bcast(rx0 & (TILE - 1), (int)q), qly = (uint32_t)bcast(ry0 & (TILE - 1), (int)q);
                            const uint32_t n = qw * qh, gq = (n + 2u) >> 1;
                            // ceil(65536 / w), w <= 16: exact from the 1-ulp reciprocal (the quotient is an integer only for powers of two, where
                            // the reciprocal is exact, and otherwise at least 1/16 away from one)
                            const uint32_t rcp = (uint32_t)ceilf(65536.0f * __builtin_amdgcn_rcpf((float)qw));
                            const unsigned long long header = (unsigned long long)(qlx | qly << 6 | (qw - 1u) << 12 | (qh - 1u) << 16) | ((unsigned long long)rcp << 32);
                            ulonglong2* dst = reinterpret_cast<ulonglong2*>(e.blockPool + (size_t)qoff * 2u);
                            for (uint32_t g = lane; g < gq; g += 64u) {
                                // granule g = words 2g - 1, 2g of the block (word -1: the header)
                                const uint32_t j1 = 2u * g, j0 = g == 0u ? 0u : j1 - 1u;
                                // (24-bit multiplies, full rate: j < 512, rcp <= 65536, row < 16, w <= 16)
                                const uint32_t row0 = __umul24(j0, rcp) >> 16, row1 = __umul24(j1, rcp) >> 16;   // exact for j < 256, w <= 16
                                const unsigned long long w0 = g == 0u ? header : win[(qy + row0) * WIN + qx + (j0 - __umul24(row0, qw))];
                                const unsigned long long w1 = j1 < n ? win[(qy + row1) * WIN + qx + (j1 - __umul24(row1, qw))] : 0ull;
                                dst[g] = make_ulonglong2(w0, w1);
                            }
                        }
                    }
                }
            }
        }
        if (!blocksDone && ((emA | emB) != 0ull || bad != 0ull)) {
            // not a block (large, clipped, masked, wide window, too few triangles for its window): the cluster goes to the
            // record kernel that follows this one (raster_setup_kernel reads the leftover list of a dense launch)
            if (lane == 0u) {
                const uint32_t k = atomicAdd(scalar_load(&kq()->leftCount), 1u);
                ChordDrawCmd cmd; cmd.objectId = hdr.objectId; cmd.meshletId = hdr.meshletId; cmd.slot = hdr.slot;
                scalar_load(&kq()->leftCmds)[k] = cmd;                                      // (capacity = the input list's: never full)
            }
            SPHASE(3);
        }
        // LDS of this wave is rewritten by the next cluster: order the reads above before those writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        hdr = hdrN;
        hdrN = header_at(c + 2u * stride);
        t0 = nt0; t1 = nt1;
        pax = nax; pay = nay; paz = naz; pbx = nbx; pby = nby; pbz = nbz;
        SPHASE(4);
    }
    // what is left of the reserves: "no entry" words (the tile kernel skips them; the slots are counted in the bin's length)
    if (HOT) WAVE_LDS_SYNC();
    if (HOT && lane < SLOT_CACHE && sc.key[lane] != 0u && sc.next[lane] < sc.end[lane]) {
        const BlockEmitParams e = load_block_emit_params();
        for (uint32_t k = sc.next[lane]; k < sc.end[lane]; k++) bin_put(e, sc.key[lane] - 1u, k, 0xFFFFFFFFu);
    }
    if (sprof && lane == 0u) {
        const uint32_t w = blockIdx.x * 4u + wave;
        if (w < CHORD_MAX_TILES * 8u / 5u) for (int i = 0; i < 5; i++) p.tilePhase[(size_t)w * 5u + i] = sph[i];
    }
#undef SPHASE
}

// Which clusters a launch of the record kernel sets up.  Pixel blocks pay when clusters are small, and then there are many of
// them: a launch is DENSE when its clusters have fewer than 16 pixels of this rank's screen each on average (BASELINE config
// 5: one pixel per cluster; config 4: 32; config 3: 2 000).  A dense launch runs raster_setup_blocks_kernel first, which turns
// every cluster it can into pixel blocks and leaves the others in the leftover list; the record kernel then sets up the
// leftover list instead of the input list.  Either way the image is the same.  (Both kernels evaluate this from the same
// device-side count; the host only knows whether a list COULD be dense, from its capacity, and skips the block kernel
// when it cannot.)
__device__ __forceinline__ bool launch_is_dense(const RasterParams& p, uint32_t count)
{
    const unsigned long long pixels = (unsigned long long)p.Wi * (unsigned long long)p.Hi / (p.shard.ranks > 1u ? p.shard.ranks : 1u);
    return p.blockCap != 0u && p.leftCmds != nullptr && (p.blockForce != 0u || (unsigned long long)count * 16ull >= pixels);
}

#ifndef SETUP_MIN_WAVES
#define SETUP_MIN_WAVES 5       // waves per SIMD the register allocation of the record kernel aims at (94 VGPRs since the area form is chosen per wave; 4 -> 5: config 4 0.420 -> 0.411 ms)
#endif
#ifndef BLOCKS_MIN_WAVES
#define BLOCKS_MIN_WAVES 6      // ... of the block kernel (80 VGPRs, 20 KB of LDS per 256 threads, 106 SGPRs: 6 workgroups per CU)
#endif
// MASKED: the scene has alpha-tested materials (their clusters emit 48-byte records with a texture-coordinate extension);
// scenes without any -- every benchmark configuration -- run the instantiation that knows nothing of them.
#ifndef SETUP_MASKED_WAVES
#define SETUP_MASKED_WAVES 4
#endif
template <bool MASKED>
__global__ __launch_bounds__(256, MASKED ? SETUP_MASKED_WAVES : SETUP_MIN_WAVES) void raster_setup_kernel(RasterParams p)   // (masked: twelve more live registers, see uvA)
{
    __shared__ float sVert[6][4][LDS_VERTS];                   // x, y, w, u, v, depth of a wave's cluster (24 KB)
    // the wave's first command is asked for TOGETHER with the count (the list has room for the index whatever the count turns out to be)
    const uint32_t c0 = __builtin_amdgcn_readfirstlane(min(blockIdx.x * 4u + (threadIdx.x >> 6), p.cmdCap - 1u));
    const uint32_t* __restrict__ cw = reinterpret_cast<const uint32_t*>(p.cmds + c0);
    const uint32_t f0 = scalar_load(cw), f1 = scalar_load(cw + 1), f2 = scalar_load(cw + 2);
    uint32_t count = *p.count;
    const ChordDrawCmd* cmds = p.cmds;
    bool firstValid = true;
    if (launch_is_dense(p, count)) { count = *p.leftCount; cmds = p.leftCmds; firstValid = false; }   // (what the block kernel left over: another list)
    raster_setup_body<MASKED>(p, cmds, count, sVert, f0, f1, f2, firstValid);
}

// HOT: the variant that draws bin slots ahead on hot tiles (above).  It costs the plain kernel's loop 3 % (registers: the loop
// is at its SGPR limit), so the host launches it only when the previous frames' longest bin says a hot tile exists
// (RasterParams::binHint, written by the tile order kernel) -- a choice of speed, both variants fill the same bins.
template <bool HOT>
__global__ __launch_bounds__(256, BLOCKS_MIN_WAVES) void raster_setup_blocks_kernel(RasterParams p)
{
    __shared__ float sVert[6][4][BLOCKS_LDS_VERTS];            // x, y, w, u, v, depth of a wave's cluster (12 KB)
    __shared__ int32_t sSnap[2][4][BLOCKS_LDS_VERTS];          // ... and its snapped 24.8 screen coordinates (4 KB)
    __shared__ unsigned long long sWin[4][WIN * WIN];          // a small cluster's pixel window (8 KB)
    __shared__ SlotCache sSlots[HOT ? 4 : 1];                  // bin slots drawn ahead on hot tiles (1.5 KB)
    const uint32_t count = *p.count;
    if (!launch_is_dense(p, count)) return;
    raster_setup_blocks_body<HOT>(p, count, sVert, sSnap, sWin, sSlots);
}

// One lane bins one record into every tile its clamped bbox may touch (conservative edge test at the tile
// corners).  Only the clipper uses it; the hot paths go through wave_bin_small2 / raster_bin_large_kernel.
__device__ void bin_record_tiles(const RasterParams& p, const TriSetup& ts, uint32_t gi)
{
    const int ea[3] = {1, 2, 0}, eb[3] = {2, 0, 1};
    const int32_t tx0 = ts.px0 >> TILE_SHIFT, tx1 = ts.px1 >> TILE_SHIFT;
    const int32_t ty0 = ts.py0 >> TILE_SHIFT, ty1 = ts.py1 >> TILE_SHIFT;
    for (int32_t ty = ty0; ty <= ty1; ty++)
        for (int32_t tx = tx0; tx <= tx1; tx++) {
            const int32_t rx0 = max(ts.px0, tx << TILE_SHIFT), rx1 = min(ts.px1, (tx << TILE_SHIFT) + TILE - 1);
            const int32_t ry0 = max(ts.py0, ty << TILE_SHIFT), ry1 = min(ts.py1, (ty << TILE_SHIFT) + TILE - 1);
            bool hit = owns_tile(p.shard, tx, ty);
#pragma unroll
            for (int i = 0; i < 3; i++) {                         // (unrolled, no early exit: the vertex arrays stay in registers)
                const int64_t dxe = (int64_t)(ts.X[eb[i]] - ts.X[ea[i]]), dye = (int64_t)(ts.Y[eb[i]] - ts.Y[ea[i]]);
                const int64_t a = -(int64_t)ts.s * dye, b = (int64_t)ts.s * dxe;
                const int64_t bias = (a > 0 || (a == 0 && b > 0)) ? 0 : -1;
                const int64_t cx = (int64_t)(a > 0 ? rx1 : rx0) * 256 + 128, cy = (int64_t)(b > 0 ? ry1 : ry0) * 256 + 128;
                hit = hit && !((int64_t)ts.s * (dxe * (cy - ts.Y[ea[i]]) - dye * (cx - ts.X[ea[i]])) + bias < 0);
            }
            if (!hit) continue;
            const uint32_t tile = (uint32_t)ty * p.tilesX + (uint32_t)tx;
            const uint32_t slot = atomicAdd(&p.tileCount[(size_t)tile * TC_STRIDE], 1u);
            bin_store(p, tile, slot, gi);
            if ((gi & 0xE0000000u) == CHORD_REC_MASKED) p.tileCount[(size_t)tile * TC_STRIDE + TC_MASKED] = 1u;
        }
}

// ---- clipper kernel (rare path) ---------------------------------------------------------------
__device__ __forceinline__ float clip_dist(const f4& v, int k)
{
    switch (k) {
    case 0: return v.w - v.z;
    case 1: return v.z;
    case 2: return GUARD_BAND * v.w + v.x;
    case 3: return GUARD_BAND * v.w - v.x;
    case 4: return GUARD_BAND * v.w + v.y;
    default: return GUARD_BAND * v.w - v.y;
    }
}

__device__ __forceinline__ f4 clip_intersect(const f4& in, const f4& out, float din, float dout)
{
    const float t = din / (din - dout);
    f4 r;
    r.x = in.x + (out.x - in.x) * t;
    r.y = in.y + (out.y - in.y) * t;
    r.z = in.z + (out.z - in.z) * t;
    r.w = in.w + (out.w - in.w) * t;
    return r;
}

// The polygon being clipped (Sutherland-Hodgman ping-pong, <= 3 + 6 vertices) is indexed dynamically, which in registers means
// scratch memory (round 2: 592 bytes per lane).  It lives in LDS instead, slot-major ([buffer][vertex][thread]: consecutive
// threads in consecutive banks); one wave per clip block works, which bounds the arrays to 38 KB.
#define CLIP_MAXV 10
#define CLIP_THREADS 64u
// LDS polygon storage of STRIDE threads (thread tl): two ping-pong buffers of clip-space vertices + texture coordinates, and
// the projected vertices of the result
template <uint32_t STRIDE>
struct ClipLds {
    float4* poly; float* pu; float* pv; int32_t* px; int32_t* py; float* pd; uint32_t tl;
    __device__ __forceinline__ float4& P(int bf, int i) const { return poly[((uint32_t)bf * CLIP_MAXV + (uint32_t)i) * STRIDE + tl]; }
    __device__ __forceinline__ float& U(int bf, int i) const { return pu[((uint32_t)bf * CLIP_MAXV + (uint32_t)i) * STRIDE + tl]; }
    __device__ __forceinline__ float& V(int bf, int i) const { return pv[((uint32_t)bf * CLIP_MAXV + (uint32_t)i) * STRIDE + tl]; }
    __device__ __forceinline__ int32_t& X(int i) const { return px[(uint32_t)i * STRIDE + tl]; }
    __device__ __forceinline__ int32_t& Y(int i) const { return py[(uint32_t)i * STRIDE + tl]; }
    __device__ __forceinline__ float& D(int i) const { return pd[(uint32_t)i * STRIDE + tl]; }
};

// One triangle of a cluster through the homogeneous clipper: its vertices are transformed again from the position stream (the
// clip-space z is not kept by the setup kernels), clipped against near / far (unless depth-clamped) and the guard band, and
// projected + snapped into L.X / L.Y / L.D.  Returns the vertex count of the resulting polygon (a fan around vertex 0), 0 when
// nothing is left; `cur` is the buffer that holds its clip-space vertices and texture coordinates.
template <uint32_t STRIDE>
__device__ __forceinline__ int clip_triangle(const RasterParams& p, const ClipLds<STRIDE>& L, const DMeshlet& m, uint32_t V, uint32_t tri,
                                             const Mat4& mvp, bool masked, int& cur)
{
    const uint32_t packedIdx = p.meshletData[m.dataOffset + V + tri];
    for (int i = 0; i < 3; i++) {
        const uint32_t li = (packedIdx >> (8 * i)) & 0xFFu;
        const uint32_t vi = p.meshletData[m.dataOffset + li] + m.vertexBase;
        const float* pos = p.positions + (size_t)vi * 3;
        const f4 h = mul_mv(mvp, pos[0], pos[1], pos[2], 1.0f);
        L.P(0, i) = make_float4(h.x, h.y, h.z, h.w);
        float u = 0.0f, v = 0.0f;
        if (masked && p.texcoords) { u = p.texcoords[(size_t)vi * 2]; v = p.texcoords[(size_t)vi * 2 + 1]; }
        L.U(0, i) = u; L.V(0, i) = v;
    }
    int np = 3;
    cur = 0;
    for (int pl = p.depthClamp ? 2 : 0; pl < 6 && np >= 3; pl++) {
        int m2 = 0;
        for (int i = 0; i < np; i++) {
            const int j = (i + 1) % np;
            const float4 Pv = L.P(cur, i), Qv = L.P(cur, j);
            const f4 P = {Pv.x, Pv.y, Pv.z, Pv.w}, Q = {Qv.x, Qv.y, Qv.z, Qv.w};
            const float pui = L.U(cur, i), pvi = L.V(cur, i), puj = L.U(cur, j), pvj = L.V(cur, j);
            const float dp = clip_dist(P, pl), dq = clip_dist(Q, pl);
            const bool pin = dp >= 0.0f, qin = dq >= 0.0f;
            if (pin && m2 < CLIP_MAXV) { L.U(cur ^ 1, m2) = pui; L.V(cur ^ 1, m2) = pvi; L.P(cur ^ 1, m2) = Pv; m2++; }
            if (pin && !qin && m2 < CLIP_MAXV) {
                const float t = dp / (dp - dq);
                L.U(cur ^ 1, m2) = pui + (puj - pui) * t; L.V(cur ^ 1, m2) = pvi + (pvj - pvi) * t;
                const f4 x = clip_intersect(P, Q, dp, dq);
                L.P(cur ^ 1, m2) = make_float4(x.x, x.y, x.z, x.w); m2++;
            } else if (!pin && qin && m2 < CLIP_MAXV) {
                const float t = dq / (dq - dp);
                L.U(cur ^ 1, m2) = puj + (pui - puj) * t; L.V(cur ^ 1, m2) = pvj + (pvi - pvj) * t;
                const f4 x = clip_intersect(Q, P, dq, dp);
                L.P(cur ^ 1, m2) = make_float4(x.x, x.y, x.z, x.w); m2++;
            }
        }
        np = m2; cur ^= 1;
    }
    if (np < 3) return 0;
    for (int i = 0; i < np; i++) {
        const float4 h = L.P(cur, i);
        if (!(h.w > 0.0f)) return 0;
        const float u = h.x / fabsf(h.w) * 0.5f + 0.5f;
        const float v = h.y / fabsf(h.w) * -0.5f + 0.5f;
        L.X(i) = (int32_t)rintf((u * p.W) * 256.0f);
        L.Y(i) = (int32_t)rintf((v * p.H) * 256.0f);
        L.D(i) = h.z / h.w;
    }
    return np;
}

__device__ void raster_clip_part(const RasterParams& p, uint32_t block, uint32_t blocks)
{
    __shared__ float4 sPoly[2 * CLIP_MAXV * CLIP_THREADS];
    __shared__ float sPU[2 * CLIP_MAXV * CLIP_THREADS], sPV[2 * CLIP_MAXV * CLIP_THREADS];     // texture coordinates ride along (masked materials only)
    __shared__ int32_t sPX[CLIP_MAXV * CLIP_THREADS], sPY[CLIP_MAXV * CLIP_THREADS];
    __shared__ float sPD[CLIP_MAXV * CLIP_THREADS];
    if (threadIdx.x >= CLIP_THREADS) return;
    const ClipLds<CLIP_THREADS> L = {sPoly, sPU, sPV, sPX, sPY, sPD, threadIdx.x};
    const uint32_t n = min(p.counters->clipTriCount[p.pass], p.clipTriCap);
    const uint32_t listShard = block % CHORD_LIST_SHARDS;
    for (uint32_t k = block * CLIP_THREADS + threadIdx.x; k < n; k += blocks * CLIP_THREADS) {
        const ClipTri ct = p.clipTris[k];
        ChordDrawCmd cmd; cmd.objectId = ct.objectId; cmd.meshletId = ct.meshletId; cmd.slot = ct.slot;
        const DMeshlet& m = p.meshlets[cmd.meshletId];
        const uint32_t V = m.vertexTriangleCount & 0xFFu;
        const uint32_t matFlags = p.objStatic[cmd.objectId].matFlags;
        const bool twoSided = (matFlags & CHORD_MATFLAG_TWO_SIDED) != 0u || p.depthOnly != 0u;
        const bool masked = CHORD_MATFLAG_ALPHA(matFlags) == CHORD_ALPHA_MASK;
        const float* mv = p.objFrame[cmd.objectId].mvp;
        Mat4 mvp;
        for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) mvp.r[r][cc] = mv[r * 4 + cc];
        int cur;
        const int np = clip_triangle(p, L, m, V, ct.tri, mvp, masked, cur);
        if (np < 3) continue;
        const uint32_t payload = p.depthOnly ? 0u : encode_triangle_instance(ct.tri, cmd.slot);
        const uint32_t slots = masked ? 1u + CHORD_MASK_EXT_SLOTS : 1u;
        for (int i = 1; i + 1 < np; i++) {
            TriSetup ts;
            ts.X[0] = L.X(0); ts.X[1] = L.X(i); ts.X[2] = L.X(i + 1);
            ts.Y[0] = L.Y(0); ts.Y[1] = L.Y(i); ts.Y[2] = L.Y(i + 1);
            float d[3] = {L.D(0), L.D(i), L.D(i + 1)};
            ts.payload = payload;
            if (!tri_setup(ts, twoSided, p.Wi, p.Hi) || !owns_rect(p.shard, ts.px0, ts.py0, ts.px1, ts.py1)) continue;
            if (p.biasConst != 0.0f || p.biasSlope != 0.0f) { const float o = depth_bias(ts, d, p.biasConst, p.biasSlope); d[0] += o; d[1] += o; d[2] += o; }
            const uint32_t li = atomicAdd(&p.counters->triCount[listShard * CHORD_SHARD_STRIDE], slots);
            if (li + slots > p.triCap) { atomicOr(&p.counters->overflow, 1u); continue; }
            const uint32_t gi = listShard * p.triCap + li;
            write_record(&p.tris[gi], ts, d, twoSided, masked);
            if (masked) {
                const uint32_t material = CHORD_MATFLAG_MATERIAL(matFlags);
                const float u3[3] = {L.U(cur, 0), L.U(cur, i), L.U(cur, i + 1)}, v3[3] = {L.V(cur, 0), L.V(cur, i), L.V(cur, i + 1)};
                const float w3[3] = {L.P(cur, 0).w, L.P(cur, i).w, L.P(cur, i + 1).w};
                write_mask_ext(&p.tris[gi + 1u], &p.materials[material], material, ts.area, u3, v3, w3);
            }
            // clipped pieces are rare: binned right here, one (scattered) atomic per tile they may touch
            bin_record_tiles(p, ts, gi | (masked ? CHORD_REC_MASKED : CHORD_REC_WIDE));   // clipped pieces take the 48-byte form
        }
    }
}

// ---- large triangles: one wave per record, one lane per candidate tile -------------------------
// One launch, two independent roles: the first CLIP_BLOCKS blocks run the clipper, the rest bin the large records.
#define CLIP_BLOCKS 64u
__device__ void raster_bin_large_part(const RasterParams& p, uint32_t block, uint32_t blocks)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // the list is cut into CHORD_LIST_SHARDS sub-lists (one reservation counter each: a single word sustains only
    // ~88 returning atomics per microsecond); a flat index is mapped to (shard, entry) through their prefix sums
    __shared__ uint32_t sStart[CHORD_LIST_SHARDS + 1];
    if (threadIdx.x < 64u) {
        const uint32_t cnt = min(p.counters->largeCount[p.pass][threadIdx.x * CHORD_SHARD_STRIDE], p.largeCap);
        uint32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t nb = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= (uint32_t)d) incl += nb; }
        sStart[threadIdx.x + 1u] = incl;
        if (threadIdx.x == 0u) sStart[0] = 0u;
    }
    __syncthreads();
    const uint32_t n = sStart[CHORD_LIST_SHARDS];
    for (uint32_t k = block * 4u + wave; k < n; k += blocks * 4u) {
        uint32_t sh = 0;
#pragma unroll
        for (uint32_t st = 32u; st > 0u; st >>= 1) if (sStart[sh + st] <= k) sh += st;
        const uint32_t gi = __builtin_amdgcn_readfirstlane(p.largeList[(size_t)sh * p.largeCap + (k - sStart[sh])]);
        const TriRec* __restrict__ r = &p.tris[gi];
        TriSetup ts;
#pragma unroll
        for (int i = 0; i < 3; i++) { ts.X[i] = r->X[i]; ts.Y[i] = r->Y[i]; }
        ts.payload = 0;
        if (!tri_setup(ts, (r->twoSided & 1u) != 0, p.Wi, p.Hi)) continue;
        const bool maskedRec = (r->twoSided & 4u) != 0u;                 // (wave-uniform: one record per wave)
        const int ea[3] = {1, 2, 0}, eb[3] = {2, 0, 1};
        int64_t a[3], b[3], bias[3], dxe[3], dye[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            dxe[i] = (int64_t)(ts.X[eb[i]] - ts.X[ea[i]]); dye[i] = (int64_t)(ts.Y[eb[i]] - ts.Y[ea[i]]);
            a[i] = -(int64_t)ts.s * dye[i]; b[i] = (int64_t)ts.s * dxe[i];
            bias[i] = (a[i] > 0 || (a[i] == 0 && b[i] > 0)) ? 0 : -1;
        }
        const int32_t tx0 = ts.px0 >> TILE_SHIFT, tx1 = ts.px1 >> TILE_SHIFT;
        const int32_t ty0 = ts.py0 >> TILE_SHIFT, ty1 = ts.py1 >> TILE_SHIFT;
        const uint32_t tw = (uint32_t)(tx1 - tx0 + 1), nt = tw * (uint32_t)(ty1 - ty0 + 1);
        for (uint32_t tb = 0; tb < nt; tb += 64u) {
            const uint32_t t = tb + lane;
            if (t >= nt) continue;
            const int32_t tx = tx0 + (int32_t)(t % tw), ty = ty0 + (int32_t)(t / tw);
            // pixel rectangle of this tile clipped to the triangle's bbox; conservative edge test at its corners
            const int32_t rx0 = max(ts.px0, tx << TILE_SHIFT), rx1 = min(ts.px1, (tx << TILE_SHIFT) + TILE - 1);
            const int32_t ry0 = max(ts.py0, ty << TILE_SHIFT), ry1 = min(ts.py1, (ty << TILE_SHIFT) + TILE - 1);
            bool hit = owns_tile(p.shard, tx, ty);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const int64_t cx = (int64_t)(a[i] > 0 ? rx1 : rx0) * 256 + 128, cy = (int64_t)(b[i] > 0 ? ry1 : ry0) * 256 + 128;
                const int64_t E = (int64_t)ts.s * (dxe[i] * (cy - ts.Y[ea[i]]) - dye[i] * (cx - ts.X[ea[i]]));
                hit = hit && !(E + bias[i] < 0);
            }
            if (hit) {
                const uint32_t tile = (uint32_t)ty * p.tilesX + (uint32_t)tx;
                const uint32_t slot = atomicAdd(&p.tileCount[(size_t)tile * TC_STRIDE], 1u);   // distinct tiles per lane
                bin_store(p, tile, slot, gi | (maskedRec ? CHORD_REC_MASKED : CHORD_REC_WIDE));   // (the large list holds 48-byte records)
                if (maskedRec) p.tileCount[(size_t)tile * TC_STRIDE + TC_MASKED] = 1u;
            }
        }
    }
}

__global__ __launch_bounds__(256) void raster_clip_and_bin_large_kernel(RasterParams p)
{
    if (blockIdx.x < CLIP_BLOCKS) raster_clip_part(p, blockIdx.x, CLIP_BLOCKS);
    else raster_bin_large_part(p, blockIdx.x - CLIP_BLOCKS, gridDim.x - CLIP_BLOCKS);
}

// ---- tile schedule: heaviest work first ---------------------------------------------------------
// The tile kernel's duration is its slowest work item plus whatever is still queued behind it.  A tile
// whose bin is long is cut into slices of TILE_SLICE entries that different workgroups scan-convert
// concurrently (the last one to finish merges them, see raster_tile_kernel), and items are dispatched
// in descending order of their entry count (longest-processing-time first): one block lists the slices
// of split tiles first and bucket-sorts the other tiles by floor(log2(count)).  Tiles without entries
// come last on the first pass of a frame (they still have to be written: that is the clear) and are
// dropped otherwise.  Item = tile | slice << 12 | (slices - 1) << 22.
#define TILE_SLICE_SHIFT CHORD_TILE_SLICE_SHIFT
#define TILE_SLICE (1u << TILE_SLICE_SHIFT)
#ifndef TILE_SPLIT_MIN
#define TILE_SPLIT_MIN 6144u       // bins up to this many entries stay whole
#endif
#ifndef TILE_ORDER_KEEP
#define TILE_ORDER_KEEP 1          // 0: the schedule kernel runs in every pass whatever chordvis_set_tile_schedule_keep says (A/B builds)
#endif
#ifndef TILE_MAKE_NEXT
#define TILE_MAKE_NEXT 1            // 0: the tile kernel's workgroup 0 does not make the next frame's schedule (compile experiments only: launch_raster still relies on it)
#endif
#ifndef TILE_DIRECT
#define TILE_DIRECT 1              // 0: later passes of a frame keep their schedule kernel (A/B builds)
#endif
#ifndef TILE_DIRECT_MAX_CLUSTERS
#define TILE_DIRECT_MAX_CLUSTERS 1024u   // a later pass with more clusters than this keeps its schedule (config 4's second pass, 30 k clusters in every tile of the screen: +15 % per frame without one)
#endif
#ifndef TILE_SLICE_MIN
#define TILE_SLICE_MIN 1024u       // the shortest slice of a pass that has fewer tiles than the device has slots
#endif
template <uint32_t NT>
__device__ __forceinline__ void tile_order_part(const RasterParams& p, uint2* __restrict__ order)
{
    __shared__ uint32_t hist[20], base[20], cursor[20], splitItems, splitCursor, longest, hotCount, hotList[CHORD_HOT_TILES], entriesAll, tilesBusy;
    const uint32_t tiles = p.tilesX * p.tilesY;
    if (threadIdx.x < 20u) { hist[threadIdx.x] = 0; cursor[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { splitItems = 0; splitCursor = 0; longest = 0; hotCount = 0; entriesAll = 0; tilesBusy = 0; }
    __syncthreads();
    // A tile's word: bin entries (clamped to the capacity) | bit 31: the bin holds pixel blocks (the tile kernel's block pass; it used to
    // ask the counter line itself, a dependent round trip per tile in front of its first bin fetch) | bit 30: the bin holds alpha-tested
    // triangles -- on the first pass of a frame the masked pass (raster_masked_tile_kernel) has written the tile already and the tile
    // kernel starts from those words instead of from zero.  ~0: not a work item of this rank (sharded frames: another rank's tiles --
    // their bins are empty, and the clear pass must not touch them).
    // (Every thread asks for the lines of all its tiles at once and keeps ONE word per tile -- bucket, slices and position are worked out
    // again where they are needed: as a workgroup of the tile kernel this part is out of line and its registers are its own, but three
    // passes of dependent line fetches made it the longest chain of a short launch.)
    auto tile_word = [&](uint32_t t) -> uint32_t {
        if (!owns_tile(p.shard, (int32_t)(t % p.tilesX), (int32_t)(t / p.tilesX))) return 0xFFFFFFFFu;
        const uint32_t* __restrict__ line = &p.tileCount[(size_t)t * TC_STRIDE];
        const uint32_t n = line[0], nb = line[1], nm = line[3];
        return min(n, bin_capacity(p)) | (nb ? 0x80000000u : 0u) | (nm ? 0x40000000u : 0u);
    };
    constexpr uint32_t PER_THREAD = CHORD_MAX_TILES / NT;
    uint32_t mine[PER_THREAD];
#pragma unroll
    for (uint32_t k = 0; k < PER_THREAD; k++) { const uint32_t t = threadIdx.x + k * NT; mine[k] = t < tiles ? tile_word(t) : 0xFFFFFFFFu; }
    uint32_t sumMine = 0, tilesMine = 0;
    if (p.tileSlots) {
        // A pass with fewer non-empty tiles than the device holds tile workgroups (a rank of an 8-rank frame owns 255 tiles of a 4K
        // target, 256 CUs hold 512 workgroups; so does a 1080p target on one GPU) leaves slots idle while every tile is one
        // workgroup's serial work: its bins are cut finer -- into about as many equal shares as there are slots, never shorter than
        // TILE_SLICE_MIN entries (a slice pays for a tile of LDS zeroed and its touched words merged through the slab).
        // The image does not depend on the cut (64-bit max).
        // (a DPP reduction per wave, then one LDS atomic per wave: handed the 1 024 atomics, the compiler's atomic optimizer walks the
        // lanes of every wave in a scalar loop -- measured +6 us on a 4-us kernel)
#pragma unroll
        for (uint32_t k = 0; k < PER_THREAD; k++) {
            const uint32_t w = mine[k];
            if (w != 0xFFFFFFFFu) { const uint32_t c = w & 0x3FFFFFFFu; sumMine += c; tilesMine += c ? 1u : 0u; }
        }
        wave_sum2(sumMine, tilesMine);
        if ((threadIdx.x & 63u) == 0u) { atomicAdd(&entriesAll, sumMine); atomicAdd(&tilesBusy, tilesMine); }
        __syncthreads();
    }
    uint32_t splitMin = p.tileSplitMin, sliceLen = p.tileSliceLen;
    if (p.tileSlots && tilesBusy < p.tileSlots) {
        const uint32_t share = (entriesAll / p.tileSlots + 511u) & ~511u;   // (whole batches of the tile kernel)
        sliceLen = min(p.tileSliceLen, max(TILE_SLICE_MIN, share));
        splitMin = min(p.tileSplitMin, sliceLen + sliceLen / 2u);
    }
    // bucket of a tile: 18 = cut into slices, 4 = 2^11.., 16 = one entry, 17 = empty
    auto slices_of = [&](uint32_t c) -> uint32_t { return (c > splitMin && !ABL(p, DBG_NO_SPLIT)) ? min((c + sliceLen - 1u) / sliceLen, CHORD_TILE_MAX_SLICES) : 0u; };
#pragma unroll
    for (uint32_t k = 0; k < PER_THREAD; k++) {
        const uint32_t t = threadIdx.x + k * NT, w = mine[k];
        if (w == 0xFFFFFFFFu) continue;
        const uint32_t c = w & 0x3FFFFFFFu;
        // a bin this long is a hot tile: the next frame's block kernel draws its slots ahead from the first cluster on (hotTiles)
        if (c >= p.slotHot && p.hotTiles) { const uint32_t h = atomicAdd(&hotCount, 1u); if (h < CHORD_HOT_TILES) hotList[h] = t | (c >= SLOT_VERY_HOT ? 0x80000000u : 0u); }
        const uint32_t sl = slices_of(c);
        if (sl) {
            atomicAdd(&splitItems, sl);
            if (c > TILE_SPLIT_MIN) atomicMax(&longest, c);   // (the host's hint keeps its meaning: bins that a full launch would keep whole report 0)
        } else atomicAdd(&hist[c ? 16u - (31u - (uint32_t)__clz(c)) : 17u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = splitItems;
        for (int b = 0; b < 18; b++) { base[b] = acc; acc += hist[b]; }
        order[0] = make_uint2((p.clearTiles || p.orderAll) ? acc : acc - hist[17], 0u);
        if (p.binHint) *p.binHint = longest;                      // (bins short enough to stay whole report 0)
        if (p.heavyHint) p.heavyHint[(splitItems != 0u || *p.count > TILE_DIRECT_MAX_CLUSTERS) ? 0 : 2] = p.binStamp;   // (launch_raster TILE_DIRECT: heavy / light)
        if (p.countHint) *p.countHint = *p.count;
        if (p.hotTiles) p.hotTiles[0] = min(hotCount, (uint32_t)CHORD_HOT_TILES);
    }
    if (p.hotTiles && threadIdx.x < min(hotCount, (uint32_t)CHORD_HOT_TILES)) p.hotTiles[1u + threadIdx.x] = hotList[threadIdx.x];
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < PER_THREAD; k++) {
        const uint32_t t = threadIdx.x + k * NT, w = mine[k];
        if (w == 0xFFFFFFFFu) continue;
        const uint32_t c = w & 0x3FFFFFFFu, sl = slices_of(c);
        if (sl) {
            const uint32_t pos = atomicAdd(&splitCursor, sl);       // (the slices of a tile side by side, the split tiles in any order: all of them start the pass)
            for (uint32_t j = 0; j < sl; j++) order[1u + pos + j] = make_uint2(t | (j << 12) | ((sl - 1u) << 22), w);
        } else {
            const uint32_t b = c ? 16u - (31u - (uint32_t)__clz(c)) : 17u;
            order[1u + base[b] + atomicAdd(&cursor[b], 1u)] = make_uint2(t, w);   // the count rides along: one round trip less per tile
        }
    }
}

// The same part as ONE workgroup of the tile kernel (raster_tile_kernel, workgroup 0 of a launch with tileOrderNext): out of line,
// and reading the launch's arguments from the kernel-argument segment itself -- inlined, its scalars cost the tile kernel twelve
// spill slots (48 bytes of scratch in a kernel that has none).  (The KERNEL asks for the segment's address and hands it over: in a
// function that is not a kernel the compiler folds __builtin_amdgcn_kernarg_segment_ptr() to null -- every load of this part faulted.)
__device__ __noinline__ void tile_order_next_part(const RasterParams* q)
{
    tile_order_part<512u>(*q, q->tileOrderNext);
}

// (Ordering in the last workgroup of the binning launch was measured twice: with that launch's 1 088 workgroups the ticket
// alone costs 11 us; with the launch cut to 128 workgroups the merged kernel still takes 1.5 us longer per pass than these
// two -- the orderer must read the 2 040 counters with agent-scope loads, past its L2, after a ticket round trip.)
__global__ __launch_bounds__(1024) void raster_tile_order_kernel(RasterParams p)
{
    tile_order_part<1024u>(p, p.tileOrder);
}

// (Tried in round 2: clipper + large-record binning + this schedule in ONE launch of 128 workgroups, the last one --
// found by a ticket -- ordering the tiles with 256 threads: 11.5 us against 5 + 5 us for the two launches.  A kernel
// boundary costs 2.5 us on this GPU (tools/microbench/launch_floor); the rest of each 5 us is the dependent-load chain
// of a kernel with next to nothing to do, which merging does not remove.)

// ---- per-tile resolve kernel ------------------------------------------------------------------

struct WideEdges {
    int64_t a[3], b[3], bias[3], dx[3], dy[3];
    int32_t Xa[3], Ya[3];
    int32_t s;
};

__device__ __forceinline__ void wide_edges(const TriSetup& ts, WideEdges& w)
{
    const int ea[3] = {1, 2, 0}, eb[3] = {2, 0, 1};
    w.s = ts.s;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        w.dx[i] = (int64_t)(ts.X[eb[i]] - ts.X[ea[i]]);
        w.dy[i] = (int64_t)(ts.Y[eb[i]] - ts.Y[ea[i]]);
        w.Xa[i] = ts.X[ea[i]]; w.Ya[i] = ts.Y[ea[i]];
        w.a[i] = -(int64_t)ts.s * w.dy[i];
        w.b[i] = (int64_t)ts.s * w.dx[i];
        w.bias[i] = (w.a[i] > 0 || (w.a[i] == 0 && w.b[i] > 0)) ? 0 : -1;
    }
}

__device__ __forceinline__ int64_t edge_at(const WideEdges& w, int i, int32_t px, int32_t py)
{
    const int64_t cx = (int64_t)px * 256 + 128, cy = (int64_t)py * 256 + 128;
    return (int64_t)w.s * (w.dx[i] * (cy - w.Ya[i]) - w.dy[i] * (cx - w.Xa[i]));
}

__device__ __forceinline__ void lds_write(unsigned long long* tile, int32_t lx, int32_t ly, float z, uint32_t payload)
{
    const unsigned long long packed = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)payload;
    atomicMax(&tile[ly * TPITCH + lx], packed);            // ds_max_u64
}

// one lane scans its own (tile-clipped) tiny bbox with 32-bit edge functions.  ONE flattened loop over the
// bbox pixels: with nested row/column loops a wave pays max(rows) x max(cols) over its lanes (a 1x16 and a
// 16x1 box in the same wave = 256 trips); flattened it pays max(area) <= the tile's tiny-area threshold.
template <int PITCH>
__device__ __forceinline__ void tile_raster_narrow(unsigned long long* tile, const TriSetup& ts, int32_t ox, int32_t oy,
                                                   int32_t x0, int32_t y0, int32_t x1, int32_t y1, bool noPixels, const bool clampZ)
{
    // (32-bit multiplies are quarter rate on this GPU; every product here has factors below 2^23 -- vertices at most 64 px
    // apart, a pixel centre inside their bbox -- and takes the full-rate 24-bit form; +-s is a select, not a multiply)
    const bool neg = ts.s < 0;
    const int32_t dx0 = ts.X[2] - ts.X[1], dy0 = ts.Y[2] - ts.Y[1];
    const int32_t dx1 = ts.X[0] - ts.X[2], dy1 = ts.Y[0] - ts.Y[2];
    const int32_t dx2 = ts.X[1] - ts.X[0], dy2 = ts.Y[1] - ts.Y[0];
    const int32_t a0 = neg ? dy0 : -dy0, b0 = neg ? -dx0 : dx0;
    const int32_t a1 = neg ? dy1 : -dy1, b1 = neg ? -dx1 : dx1;
    const int32_t a2 = neg ? dy2 : -dy2, b2 = neg ? -dx2 : dx2;
    const int32_t bias0 = (a0 > 0 || (a0 == 0 && b0 > 0)) ? 0 : -1;
    const int32_t bias1 = (a1 > 0 || (a1 == 0 && b1 > 0)) ? 0 : -1;
    const int32_t bias2 = (a2 > 0 || (a2 == 0 && b2 > 0)) ? 0 : -1;
    const int32_t cx0 = x0 * 256 + 128, cy0 = y0 * 256 + 128;
    // s * (dx * (cy - Y) - dy * (cx - X)) = b * (cy - Y) + a * (cx - X)
    int32_t r0 = __mul24(b0, cy0 - ts.Y[1]) + __mul24(a0, cx0 - ts.X[1]) + bias0;   // bias folded in: inside <=> all >= 0
    int32_t r1 = __mul24(b1, cy0 - ts.Y[2]) + __mul24(a1, cx0 - ts.X[2]) + bias1;
    int32_t r2 = __mul24(b2, cy0 - ts.Y[0]) + __mul24(a2, cx0 - ts.X[0]) + bias2;
    int32_t E0 = r0, E1 = r1, E2 = r2;
    const int32_t w = x1 - x0 + 1, count = __mul24(w, y1 - y0 + 1);
    // step to the next pixel / from the last pixel of a row to the first of the next; the body is branch-free:
    // a pixel outside the triangle merges 0, which ds_max ignores
    const int32_t sx0 = a0 * 256, sx1 = a1 * 256, sx2 = a2 * 256;
    const int32_t sw0 = b0 * 256 - __mul24(w - 1, sx0), sw1 = b1 * 256 - __mul24(w - 1, sx1), sw2 = b2 * 256 - __mul24(w - 1, sx2);
    int32_t col = 0;
    unsigned long long* px = tile + (y0 - oy) * PITCH + (x0 - ox);
    const unsigned long long payload = (unsigned long long)ts.payload;
    for (int32_t i = 0; i < count; i++) {
        const bool inside = (E0 | E1 | E2) >= 0 && !noPixels;
        const float l1 = (float)(E1 - bias1) * ts.invA, l2 = (float)(E2 - bias2) * ts.invA;
        float z = (ts.d0 + l1 * ts.e1) + l2 * ts.e2;
        if (clampZ) z = fminf(fmaxf(z, 0.0f), 1.0f);
        atomicMax(px, inside ? (((unsigned long long)__float_as_uint(z) << 32) | payload) : 0ull);   // ds_max_u64
        col++;
        const bool wrap = col == w;
        E0 += wrap ? sw0 : sx0; E1 += wrap ? sw1 : sx1; E2 += wrap ? sw2 : sx2;
        px += wrap ? PITCH - w + 1 : 1;
        col = wrap ? 0 : col;
    }
}

// ---- row units ------------------------------------------------------------------------------
// Triangles are not equal: a tile may hold thousands of 1-pixel triangles or a few that cover it
// completely.  Each batch of 256 bin entries is therefore cut into (triangle, pixel row) units; a
// block-wide prefix sum over the row counts hands every thread one row at a time, so a lane's loop
// length is one row span (<= 64 pixels) instead of one bbox area (<= 4096).  Tiny triangles (clipped
// bbox <= TINY_AREA / TINY_AREA_DENSE pixels) are scanned directly by the thread that set them up.
//
// Edge functions in the row loop are incremental (no multiplies).  Three exact representations:
//   kind 0  int32   vertices at most 64 px apart (|E| < 2^30)
//   kind 1  double  |coordinates| < 2^25 sub-pixels: every product and sum below 2^53, so fp64 is exact
//                   and v_cvt_f32_f64 IS the canonical (float)(double)E
//   kind 2  int64   anything else (guard-band monsters)
// The threshold follows the tile: where a bin is long (>= TINY_DENSE_MIN entries: distant instances, a triangle or two per
// pixel) nearly everything is small and a lane's own loop of up to 48 pixels beats a unit list that is mostly one- and two-row
// triangles; where it is short the triangles are larger and more varied, and 8 keeps the lanes of a wave alike.  Measured
// (one threshold 8 / 16 / 32: config 3 0.1882 / 0.1897 / 0.1932 ms, config 4 0.443 / 0.429 / 0.419; 8 | 48 at 1 024 entries:
// 0.1875 and 0.4194).
#ifndef TINY_AREA
#define TINY_AREA 8
#endif
#ifndef TINY_AREA_DENSE
#define TINY_AREA_DENSE 48
#endif
#ifndef TINY_DENSE_MIN
#define TINY_DENSE_MIN 1024u
#endif

struct UnitParams {           // one batch entry, as the row loop wants it
    int32_t X[3], Y[3];
    float d0, e1, e2, invA;
    uint32_t payload;
    uint32_t box;             // x0 | y0 << 8 | x1 << 16 | y1 << 24, tile-local
    int32_t skind;            // s in bit 0 (1 = negative), kind << 1
};
#define TB 512                  // threads per tile workgroup = entries per batch
static_assert(TB == 512, "tile_order_part rounds a pass's slice length to batches of 512");
#define UNIT_CAP 4096           // units per round of a batch

template <typename E_t>
__device__ __forceinline__ void scan_row(unsigned long long* __restrict__ tileRow, const UnitParams& u, int32_t ox, int32_t py,
                                         int32_t lx0, int32_t lx1, bool noPixels, const bool clampZ)
{
    const E_t s = (u.skind & 1) ? (E_t)-1 : (E_t)1;
    const E_t dx0 = (E_t)(u.X[2] - u.X[1]), dy0 = (E_t)(u.Y[2] - u.Y[1]);
    const E_t dx1 = (E_t)(u.X[0] - u.X[2]), dy1 = (E_t)(u.Y[0] - u.Y[2]);
    const E_t dx2 = (E_t)(u.X[1] - u.X[0]), dy2 = (E_t)(u.Y[1] - u.Y[0]);
    const E_t a0 = -s * dy0, b0 = s * dx0, a1 = -s * dy1, b1 = s * dx1, a2 = -s * dy2, b2 = s * dx2;
    const E_t bias0 = (a0 > 0 || (a0 == 0 && b0 > 0)) ? (E_t)0 : (E_t)-1;
    const E_t bias1 = (a1 > 0 || (a1 == 0 && b1 > 0)) ? (E_t)0 : (E_t)-1;
    const E_t bias2 = (a2 > 0 || (a2 == 0 && b2 > 0)) ? (E_t)0 : (E_t)-1;
    const E_t cx = (E_t)(ox + lx0) * (E_t)256 + (E_t)128, cy = (E_t)py * (E_t)256 + (E_t)128;
    E_t E0 = s * (dx0 * (cy - (E_t)u.Y[1]) - dy0 * (cx - (E_t)u.X[1])) + bias0;     // bias folded in: inside <=> all >= 0
    E_t E1 = s * (dx1 * (cy - (E_t)u.Y[2]) - dy1 * (cx - (E_t)u.X[2])) + bias1;
    E_t E2 = s * (dx2 * (cy - (E_t)u.Y[0]) - dy2 * (cx - (E_t)u.X[0])) + bias2;
    const E_t st0 = a0 * (E_t)256, st1 = a1 * (E_t)256, st2 = a2 * (E_t)256;
    // Span of the row in steps k from lx0: E_i + k * st_i >= 0 for all i.  The crossings are estimated in fp32
    // (v_rcp_f32; error far below the +-1 step of slack taken on both sides) and only bound the
    // loop; coverage itself is decided by the exact edge values inside it, so the loop has no data-dependent
    // branch: a pixel outside merges the value 0, which ds_max ignores.
    float klo = 0.0f, khi = (float)(lx1 - lx0);
    {
        const float e[3] = {(float)E0, (float)E1, (float)E2}, t[3] = {(float)st0, (float)st1, (float)st2};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float q = fminf(fmaxf(-e[i] * __builtin_amdgcn_rcpf(t[i]), -4.0f), 4096.0f);   // NaN (0 * inf) -> -4
            if (t[i] > 0.0f) klo = fmaxf(klo, floorf(q) - 1.0f);
            else if (t[i] < 0.0f) khi = fminf(khi, floorf(q) + 1.0f);
            else if (e[i] < 0.0f) khi = -1.0f;                                                  // constant and outside
        }
    }
    const int32_t k0 = (int32_t)klo, k1 = (int32_t)khi;
    E0 += (E_t)k0 * st0; E1 += (E_t)k0 * st1; E2 += (E_t)k0 * st2;
    unsigned long long* px = tileRow + lx0 + k0;
    const unsigned long long payload = noPixels ? 0ull : (unsigned long long)u.payload;
    for (int32_t k = k0; k <= k1; k++) {
        const bool inside = std::is_floating_point<E_t>::value ? (E0 >= (E_t)0 && E1 >= (E_t)0 && E2 >= (E_t)0)
                                                              : (((int64_t)E0 | (int64_t)E1 | (int64_t)E2) >= 0);
        // canonical l_i = float(E_i) * invA with E_i the unbiased integer
        const float l1 = (float)(double)(E1 - bias1) * u.invA, l2 = (float)(double)(E2 - bias2) * u.invA;
        float z = (u.d0 + l1 * u.e1) + l2 * u.e2;
        if (clampZ) z = fminf(fmaxf(z, 0.0f), 1.0f);
        const unsigned long long packed = ((unsigned long long)__float_as_uint(z) << 32) | payload;
        atomicMax(px, inside && !noPixels ? packed : 0ull);      // ds_max_u64
        E0 += st0; E1 += st1; E2 += st2; px++;
    }
}

// exclusive scan of one value per thread over the TB-thread block
__device__ __forceinline__ uint32_t block_scan_tb(uint32_t v, uint32_t* waveSums, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t nb = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += nb;
    }
    if (lane == 63u) waveSums[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < TB / 64u; w++) { const uint32_t sw = waveSums[w]; if (w < wave) base += sw; tot += sw; }
    *total = tot;
    return base + incl - v;
}


// ---- batch entries with pre-computed edge constants ---------------------------------------------------------------
// The thread that fetches a bin entry also reduces it to what a pixel row needs, ONCE: for triangles whose vertices are
// at most 64 px apart (nearly all) the three edge functions as  E_i(lx, ly) = C_i + ((a_i * lx + b_i * ly) << 8)  with
// C_i the (top-left-biased) value at the tile's origin pixel and a_i, b_i the 16-bit pixel steps / 256 -- a row unit
// then costs two multiply-adds per edge instead of re-deriving deltas, orientation, biases and the 24.8 products
// (about 50 of the ~135 instructions a unit spent before its first pixel).  All values are exact 32-bit integers:
// |a|, |b| <= 2^14 and a pixel of the tile is at most 2^15 + 128 sub-pixels from a vertex, so |E| < 2^31.
// Rows wider than SEG pixels are cut into SEG-pixel segments, one unit each: a wave's row loops are at most SEG trips
// long whatever mix of triangles the tile holds (62 % of the lanes were active in round 1), and a tile covered by a few
// huge triangles becomes hundreds of units instead of 64.
// Wide triangles (kind 1: fp64 edges, kind 2: int64) keep their vertices and derive the edges per unit as before.
#ifndef SEG_SHIFT
#define SEG_SHIFT 6             // 64: one unit per row (16 / 32 were measured 3 % / 1 % slower on config 3: more units, same trips)
#endif
#define SEG (1 << SEG_SHIFT)
// Units of masked (alpha-tested) triangles: MASKED_ROWS pixel rows x segments of MASKED_SEG pixels, MASKED_PIXELS_PER_TRIP pixels
// per trip of the row loop.  Measured on street_4k_masked (profiles/r04_masked_variants.txt; tile kernel per frame): one row x 64
// px x 1 pixel per trip 418 us; segments of 32 / 16 / 8 px 419 / 421 / 498 (and no different once a unit's set-up is one round
// trip); two pixels per trip (taps of both in flight) 488: the masked instantiation sits at 128 VGPRs with scratch, more live taps
// move spills into the loop; 16 rows per unit 1 012.  By removal (-DEXP_MASKED): of the pass's ~250 us the bilinear taps are 80,
// the remainders of non-power-of-two sizes were 45 (period_mod recovered 30), the two divisions 5.
#ifndef MASKED_ROWS
#define MASKED_ROWS 1
#endif
#ifndef MASKED_SEG_SHIFT
#define MASKED_SEG_SHIFT 6
#endif
#define MASKED_SEG (1 << MASKED_SEG_SHIFT)
#ifndef MASKED_PIXELS_PER_TRIP
#define MASKED_PIXELS_PER_TRIP 1
#endif
#define ENTRY_WORDS 13            // word 12: record index of a masked triangle (its extension follows it)
// entries per batch of the tile kernel's triangle pipeline (the first TILE_BATCH threads fetch and set up one bin entry each) and units
// per round of its unit list: 512 / 4096 = 26 + 16 KB of LDS beside the 33-KB tile = two workgroups per CU; 256 / 1536 = 13 + 6 KB =
// three, if the registers allow it (TILE_MIN_BLOCKS 6: 80 VGPRs) -- a build switch, measured in profiles/r05_tile_kernel_experiments.txt
#ifndef TILE_BATCH
#define TILE_BATCH TB
#endif
#ifndef TILE_UNIT_CAP
#define TILE_UNIT_CAP UNIT_CAP
#endif
struct EntrySoA { uint32_t w[ENTRY_WORDS][TILE_BATCH]; };
#define EF_KIND_SHIFT 24          // box word: x0 | y0 << 6 | x1 << 12 | y1 << 18 | kind << 24 | bias1 << 26 | bias2 << 27 | sneg << 28

template <typename E_t>
__device__ __forceinline__ int32_t scan_span(unsigned long long* __restrict__ tileRow, E_t E0, E_t E1, E_t E2, E_t st0, E_t st1, E_t st2,
                                          E_t bias1, E_t bias2, float d0, float e1, float e2, float invA, uint32_t payloadIn,
                                          int32_t lx0, int32_t lx1, bool noPixels, const bool clampZ = false)
{
    // Span of the row in steps k from lx0 (see scan_row): fp32 estimates only BOUND the loop, coverage is exact inside.
    float klo = 0.0f, khi = (float)(lx1 - lx0);
    {
        const float e[3] = {(float)E0, (float)E1, (float)E2}, t[3] = {(float)st0, (float)st1, (float)st2};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float q = fminf(fmaxf(-e[i] * __builtin_amdgcn_rcpf(t[i]), -4.0f), 4096.0f);   // NaN (0 * inf) -> -4
            if (t[i] > 0.0f) klo = fmaxf(klo, floorf(q) - 1.0f);
            else if (t[i] < 0.0f) khi = fminf(khi, floorf(q) + 1.0f);
            else if (e[i] < 0.0f) khi = -1.0f;                                                  // constant and outside
        }
    }
    const int32_t k0 = (int32_t)klo, k1 = (int32_t)khi;
    E0 += (E_t)k0 * st0; E1 += (E_t)k0 * st1; E2 += (E_t)k0 * st2;
    unsigned long long* px = tileRow + lx0 + k0;
    const unsigned long long payload = noPixels ? 0ull : (unsigned long long)payloadIn;
    for (int32_t k = k0; k <= k1; k++) {
        const bool inside = std::is_floating_point<E_t>::value ? (E0 >= (E_t)0 && E1 >= (E_t)0 && E2 >= (E_t)0)
                                                              : (((int64_t)E0 | (int64_t)E1 | (int64_t)E2) >= 0);
        const float l1 = (float)(double)(E1 - bias1) * invA, l2 = (float)(double)(E2 - bias2) * invA;
        float z = (d0 + l1 * e1) + l2 * e2;
        if (clampZ) z = fminf(fmaxf(z, 0.0f), 1.0f);
        const unsigned long long packed = ((unsigned long long)__float_as_uint(z) << 32) | payload;
        atomicMax(px, inside && !noPixels ? packed : 0ull);      // ds_max_u64
        E0 += st0; E1 += st1; E2 += st2; px++;
    }
    return max(k1 - k0 + 1, 0);
}

// int32 specialisation of the conversion (float)(double)E: exact for |E| < 2^31 either way, one instruction instead of two
__device__ __forceinline__ int32_t scan_span_i32(unsigned long long* __restrict__ tileRow, int32_t E0, int32_t E1, int32_t E2,
                                              int32_t st0, int32_t st1, int32_t st2, int32_t bias1, int32_t bias2,
                                              float d0, float e1, float e2, float invA, uint32_t payloadIn,
                                              int32_t lx0, int32_t lx1, bool noPixels, const bool clampZ)
{
    float klo = 0.0f, khi = (float)(lx1 - lx0);
    {
        const float e[3] = {(float)E0, (float)E1, (float)E2}, t[3] = {(float)st0, (float)st1, (float)st2};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float q = fminf(fmaxf(-e[i] * __builtin_amdgcn_rcpf(t[i]), -4.0f), 4096.0f);
            if (t[i] > 0.0f) klo = fmaxf(klo, floorf(q) - 1.0f);
            else if (t[i] < 0.0f) khi = fminf(khi, floorf(q) + 1.0f);
            else if (e[i] < 0.0f) khi = -1.0f;
        }
    }
    const int32_t k0 = (int32_t)klo, k1 = (int32_t)khi;
    E0 += __mul24(k0, st0); E1 += __mul24(k0, st1); E2 += __mul24(k0, st2);     // |st| <= 2^22, 0 <= k0 < 2^12: 24-bit operands
    // (Round 3: trimming the ~3 slack pixels of the fp32 bounds with exact edge tests before the loop -- 1.5 of a row's 5.6
    // two-pixel trips -- and predicating the merges instead of merging 0 were both measured: tile kernel 107 -> 109 / 108 us
    // on config 3, 197 -> 200 / 197 on config 4.  The LDS atomic pipe is ~55 % busy in pass 0 but it is not what the kernel
    // waits for; profiles/r03_tile_kernel_experiments.txt.)
    int32_t U1 = E1 - bias1, U2 = E2 - bias2;                    // the unbiased values the canonical depth uses
    unsigned long long* px = tileRow + lx0 + k0;
    const unsigned long long payload = noPixels ? 0ull : (unsigned long long)payloadIn;
    const int32_t kFirst = k0;
    // two pixels per trip.  The second may lie one past the bound: coverage is decided by the exact edge values, and
    // the word it would touch is at most the row's padding word (lx1 <= 63), a pixel right of the screen or a pixel of
    // the next segment of the same row (which merges the same value) -- never another row.
    for (int32_t k = k0; k <= k1; k += 2) {
        const bool insideA = (E0 | E1 | E2) >= 0;
        const bool insideB = ((E0 + st0) | (E1 + st1) | (E2 + st2)) >= 0;
        const float l1a = (float)U1 * invA, l2a = (float)U2 * invA;   // (float)(double)E == (float)E: one rounding of an exact integer
        const float l1b = (float)(U1 + st1) * invA, l2b = (float)(U2 + st2) * invA;
        float za = (d0 + l1a * e1) + l2a * e2;
        float zb = (d0 + l1b * e1) + l2b * e2;
        if (clampZ) { za = fminf(fmaxf(za, 0.0f), 1.0f); zb = fminf(fmaxf(zb, 0.0f), 1.0f); }
        atomicMax(px, insideA && !noPixels ? (((unsigned long long)__float_as_uint(za) << 32) | payload) : 0ull);      // ds_max_u64
        atomicMax(px + 1, insideB && !noPixels ? (((unsigned long long)__float_as_uint(zb) << 32) | payload) : 0ull);
        E0 += 2 * st0; E1 += 2 * st1; E2 += 2 * st2; U1 += 2 * st1; U2 += 2 * st2; px += 2;
    }
    return max(k1 - kFirst + 1, 0);
}

// entry record of a triangle that needs row units; returns its unit count
__device__ __forceinline__ uint32_t entry_store(EntrySoA& en, uint32_t t, const TriSetup& ts, bool narrow,
                                                int32_t ox, int32_t oy, int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                                                bool masked = false, uint32_t recIndex = 0u)
{
    const bool maskedNarrow = masked && narrow;                   // (bit 29 of the box word: masked_row may use 32-bit edge functions)
    narrow = narrow && !masked;                                   // masked triangles keep their vertices (kind 3)
    uint32_t box = (uint32_t)(x0 - ox) | ((uint32_t)(y0 - oy) << 6) | ((uint32_t)(x1 - ox) << 12) | ((uint32_t)(y1 - oy) << 18);
    if (narrow) {
        const int32_t s = ts.s;
        const int32_t dx0 = ts.X[2] - ts.X[1], dy0 = ts.Y[2] - ts.Y[1];
        const int32_t dx1 = ts.X[0] - ts.X[2], dy1 = ts.Y[0] - ts.Y[2];
        const int32_t dx2 = ts.X[1] - ts.X[0], dy2 = ts.Y[1] - ts.Y[0];
        const int32_t a0 = -s * dy0, b0 = s * dx0, a1 = -s * dy1, b1 = s * dx1, a2 = -s * dy2, b2 = s * dx2;
        const int32_t bias0 = (a0 > 0 || (a0 == 0 && b0 > 0)) ? 0 : -1;
        const int32_t bias1 = (a1 > 0 || (a1 == 0 && b1 > 0)) ? 0 : -1;
        const int32_t bias2 = (a2 > 0 || (a2 == 0 && b2 > 0)) ? 0 : -1;
        const int32_t cx = ox * 256 + 128, cy = oy * 256 + 128;
        en.w[0][t] = ((uint32_t)a0 & 0xFFFFu) | ((uint32_t)b0 << 16);
        en.w[1][t] = ((uint32_t)a1 & 0xFFFFu) | ((uint32_t)b1 << 16);
        en.w[2][t] = ((uint32_t)a2 & 0xFFFFu) | ((uint32_t)b2 << 16);
        // |deltas| <= 2^14 and the tile origin is at most 2^15 + 128 sub-pixels from a vertex: 24-bit operands, |E| < 2^31
        en.w[3][t] = (uint32_t)(s * (__mul24(dx0, cy - ts.Y[1]) - __mul24(dy0, cx - ts.X[1])) + bias0);
        en.w[4][t] = (uint32_t)(s * (__mul24(dx1, cy - ts.Y[2]) - __mul24(dy1, cx - ts.X[2])) + bias1);
        en.w[5][t] = (uint32_t)(s * (__mul24(dx2, cy - ts.Y[0]) - __mul24(dy2, cx - ts.X[0])) + bias2);
        box |= (bias1 ? 1u << 26 : 0u) | (bias2 ? 1u << 27 : 0u);
    } else {
        int32_t mag = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) mag = max(mag, max(abs(ts.X[i]), abs(ts.Y[i])));
        en.w[0][t] = (uint32_t)ts.X[0]; en.w[1][t] = (uint32_t)ts.X[1]; en.w[2][t] = (uint32_t)ts.X[2];
        en.w[3][t] = (uint32_t)ts.Y[0]; en.w[4][t] = (uint32_t)ts.Y[1]; en.w[5][t] = (uint32_t)ts.Y[2];
        box |= (masked ? 3u : (mag < (1 << 25) ? 1u : 2u)) << EF_KIND_SHIFT;
        box |= ts.s < 0 ? 1u << 28 : 0u;
        box |= maskedNarrow ? 1u << 29 : 0u;
        if (masked) en.w[12][t] = recIndex;
    }
    en.w[6][t] = __float_as_uint(ts.d0); en.w[7][t] = __float_as_uint(ts.e1); en.w[8][t] = __float_as_uint(ts.e2);
    en.w[9][t] = __float_as_uint(ts.invA); en.w[10][t] = ts.payload; en.w[11][t] = box;
    // units: one per (row, segment); a masked triangle's are groups of MASKED_ROWS rows (masked_rows)
    if (masked) return ((uint32_t)(y1 - y0) / MASKED_ROWS + 1u) * ((uint32_t)((x1 - x0) >> MASKED_SEG_SHIFT) + 1u);
    return (uint32_t)(y1 - y0 + 1) * ((uint32_t)((x1 - x0) >> SEG_SHIFT) + 1u);
}

// A pixel row of a masked triangle: exact int64 edges, canonical depth, and per covered pixel the perspective-correct
// texture coordinates, one alpha fetch and the clip() of mesh_raster.hlsl:198-204.
// E_t: int32_t for triangles whose vertices are at most 64 px apart (|E| < 2^31, as in scan_span_i32; (float)E is then the
// canonical (float)(double)E), int64_t for the rest.  The span of the row is bounded first (fp32 estimates, one pixel of slack:
// scan_span), so the loop runs over the covered pixels, not over the bbox row.
// (A unit of a masked triangle is a GROUP of up to MASKED_ROWS pixel rows, not one: what a unit pays before its first pixel --
// the triangle's extension record, its material, the texture level, the edge constants: three dependent memory round trips --
// is then paid once per small triangle instead of once per row.)
template <typename E_t>
__device__ __forceinline__ void masked_rows(const RasterParams& p, unsigned long long* __restrict__ tileRow0, const UnitParams& u, uint32_t recIndex,
                                         int32_t ox, int32_t py0, int32_t nrows, int32_t lx0, int32_t lx1, bool noPixels, const bool clampZ)
{
    // (60 bytes of the extension, one round trip: nothing here depends on another fetch)
    const TriRecMaskExt ext = *reinterpret_cast<const TriRecMaskExt*>(&p.tris[recIndex + 1u]);
    struct { float alphaFactor, alphaCutOff; } m = {ext.alphaFactor, ext.alphaCutOff};
    const AlphaLevel tex = alpha_level(p.texAlpha, ext);
    const E_t sgn = (u.skind & 1) ? (E_t)-1 : (E_t)1;
    const E_t dx0 = (E_t)(u.X[2] - u.X[1]), dy0 = (E_t)(u.Y[2] - u.Y[1]);
    const E_t dx1 = (E_t)(u.X[0] - u.X[2]), dy1 = (E_t)(u.Y[0] - u.Y[2]);
    const E_t dx2 = (E_t)(u.X[1] - u.X[0]), dy2 = (E_t)(u.Y[1] - u.Y[0]);
    const E_t a0 = -sgn * dy0, b0 = sgn * dx0, a1 = -sgn * dy1, b1 = sgn * dx1, a2 = -sgn * dy2, b2 = sgn * dx2;
    const E_t bias0 = (a0 > 0 || (a0 == 0 && b0 > 0)) ? (E_t)0 : (E_t)-1;
    const E_t bias1 = (a1 > 0 || (a1 == 0 && b1 > 0)) ? (E_t)0 : (E_t)-1;
    const E_t bias2 = (a2 > 0 || (a2 == 0 && b2 > 0)) ? (E_t)0 : (E_t)-1;
    const E_t cx = (E_t)(ox + lx0) * (E_t)256 + (E_t)128, cy = (E_t)py0 * (E_t)256 + (E_t)128;
    E_t R0 = sgn * (dx0 * (cy - (E_t)u.Y[1]) - dy0 * (cx - (E_t)u.X[1])) + bias0;       // bias folded in: inside <=> all >= 0
    E_t R1 = sgn * (dx1 * (cy - (E_t)u.Y[2]) - dy1 * (cx - (E_t)u.X[2])) + bias1;
    E_t R2 = sgn * (dx2 * (cy - (E_t)u.Y[0]) - dy2 * (cx - (E_t)u.X[0])) + bias2;
    const E_t st0 = a0 * (E_t)256, st1 = a1 * (E_t)256, st2 = a2 * (E_t)256;
    const E_t sy0 = b0 * (E_t)256, sy1 = b1 * (E_t)256, sy2 = b2 * (E_t)256;           // one row down
    for (int32_t r = 0; r < nrows; r++, R0 += sy0, R1 += sy1, R2 += sy2) {
    E_t E0 = R0, E1 = R1, E2 = R2;
    unsigned long long* __restrict__ tileRow = tileRow0 + r * TPITCH;
    float klo = 0.0f, khi = (float)(lx1 - lx0);
    {
        const float e[3] = {(float)E0, (float)E1, (float)E2}, t[3] = {(float)st0, (float)st1, (float)st2};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float q = fminf(fmaxf(-e[i] * __builtin_amdgcn_rcpf(t[i]), -4.0f), 4096.0f);   // NaN (0 * inf) -> -4
            if (t[i] > 0.0f) klo = fmaxf(klo, floorf(q) - 1.0f);
            else if (t[i] < 0.0f) khi = fminf(khi, floorf(q) + 1.0f);
            else if (e[i] < 0.0f) khi = -1.0f;
        }
    }
    const int32_t k0 = (int32_t)klo, k1 = (int32_t)khi;
    E0 += (E_t)k0 * st0; E1 += (E_t)k0 * st1; E2 += (E_t)k0 * st2;
    // the packed word of the pixel with these (biased) edge values, 0 when it is outside or its alpha fails the cut-off.  No branch:
    // a pixel outside the triangle evaluates its texture coordinates anyway (whatever they are, texel_floor / wrap_index give an
    // index inside the level), so that the taps of both pixels of a trip are in flight together -- the loop is bound by the latency
    // of its dependent byte loads, one round trip per covered pixel before.
    auto pixel = [&](E_t e0, E_t e1, E_t e2) -> unsigned long long {
        const bool inside = (e0 | e1 | e2) >= 0;
        const float l1 = std::is_same<E_t, int32_t>::value ? (float)(e1 - bias1) * u.invA : (float)(double)(e1 - bias1) * u.invA;
        const float l2 = std::is_same<E_t, int32_t>::value ? (float)(e2 - bias2) * u.invA : (float)(double)(e2 - bias2) * u.invA;
        const float l0 = (1.0f - l1) - l2;
        const float den = (l0 * ext.iw[0] + l1 * ext.iw[1]) + l2 * ext.iw[2];
        float tu = (l0 * ext.uw[0] + l1 * ext.uw[1]) + l2 * ext.uw[2];
        float tv = (l0 * ext.vw[0] + l1 * ext.vw[1]) + l2 * ext.vw[2];
        if (EXP_MASKED & 8) {
            float r = __builtin_amdgcn_rcpf(den);
            r = __builtin_fmaf(__builtin_fmaf(-den, r, 1.0f), r, r);
            float q = tu * r; q = __builtin_fmaf(__builtin_fmaf(-den, q, tu), r, q); tu = __builtin_fmaf(__builtin_fmaf(-den, q, tu), r, q);
            q = tv * r; q = __builtin_fmaf(__builtin_fmaf(-den, q, tv), r, q); tv = __builtin_fmaf(__builtin_fmaf(-den, q, tv), r, q);
        } else { tu = tu / den; tv = tv / den; }
        const float alpha = sample_alpha(tex, tu, tv);
        const bool keep = inside && !(alpha * m.alphaFactor - m.alphaCutOff < 0.0f) && !noPixels;      // clip()
        float z = (u.d0 + l1 * u.e1) + l2 * u.e2;
        if (clampZ) z = fminf(fmaxf(z, 0.0f), 1.0f);
        return keep ? (((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)u.payload) : 0ull;
    };
    unsigned long long* px = tileRow + lx0 + k0;
#if MASKED_PIXELS_PER_TRIP == 2
    for (int32_t k = k0; k <= k1; k += 2, E0 += 2 * st0, E1 += 2 * st1, E2 += 2 * st2, px += 2) {
        const unsigned long long va = pixel(E0, E1, E2);
        const unsigned long long vb = k + 1 <= k1 ? pixel(E0 + st0, E1 + st1, E2 + st2) : 0ull;
        atomicMax(px, va);                                                        // ds_max_u64 (0 changes nothing)
        atomicMax(px + 1, vb);                                                    // (at most the row's padding word when k + 1 > lx1)
    }
#else
    for (int32_t k = k0; k <= k1; k++, E0 += st0, E1 += st1, E2 += st2, px++) {
        if ((E0 | E1 | E2) < 0) continue;
        atomicMax(px, pixel(E0, E1, E2));
    }
#endif
    }
}

// one unit = (entry e, its j-th (row, segment))
template <bool MASKED, bool DEPTH>
__device__ __forceinline__ int32_t entry_unit(const RasterParams& p, const EntrySoA& en, unsigned long long* tile, uint32_t e, uint32_t row, uint32_t seg,
                                           int32_t ox, int32_t oy, bool noPixels)
{
    const uint32_t box = en.w[11][e];
    const int32_t bx0 = (int32_t)(box & 63u), bx1 = (int32_t)((box >> 12) & 63u);
    const int32_t ly = (int32_t)row;
    const int32_t lx0 = bx0 + (int32_t)(seg << SEG_SHIFT), lx1 = min(bx1, lx0 + SEG - 1);
    unsigned long long* tileRow = tile + ly * TPITCH;
    const float d0 = __uint_as_float(en.w[6][e]), e1 = __uint_as_float(en.w[7][e]), e2 = __uint_as_float(en.w[8][e]);
    const float invA = __uint_as_float(en.w[9][e]);
    const uint32_t payload = en.w[10][e];
    const uint32_t kind = (box >> EF_KIND_SHIFT) & 3u;
    if (kind == 0u) {
        const uint32_t p0 = en.w[0][e], p1 = en.w[1][e], p2 = en.w[2][e];
        const int32_t a0 = (int32_t)(int16_t)(p0 & 0xFFFFu), b0 = (int32_t)p0 >> 16;
        const int32_t a1 = (int32_t)(int16_t)(p1 & 0xFFFFu), b1 = (int32_t)p1 >> 16;
        const int32_t a2 = (int32_t)(int16_t)(p2 & 0xFFFFu), b2 = (int32_t)p2 >> 16;
        // (16-bit steps times 6-bit pixel offsets: full-rate 24-bit multiplies)
        const int32_t E0 = (int32_t)en.w[3][e] + (__mul24(a0, lx0) + __mul24(b0, ly)) * 256;
        const int32_t E1 = (int32_t)en.w[4][e] + (__mul24(a1, lx0) + __mul24(b1, ly)) * 256;
        const int32_t E2 = (int32_t)en.w[5][e] + (__mul24(a2, lx0) + __mul24(b2, ly)) * 256;
        return scan_span_i32(tileRow, E0, E1, E2, a0 * 256, a1 * 256, a2 * 256, (box >> 26) & 1u ? -1 : 0, (box >> 27) & 1u ? -1 : 0,
                             d0, e1, e2, invA, payload, lx0, lx1, noPixels, DEPTH);
    } else {
        UnitParams u;
        u.X[0] = (int32_t)en.w[0][e]; u.X[1] = (int32_t)en.w[1][e]; u.X[2] = (int32_t)en.w[2][e];
        u.Y[0] = (int32_t)en.w[3][e]; u.Y[1] = (int32_t)en.w[4][e]; u.Y[2] = (int32_t)en.w[5][e];
        u.d0 = d0; u.e1 = e1; u.e2 = e2; u.invA = invA; u.payload = payload; u.box = 0;
        u.skind = (int32_t)((box >> 28) & 1u) | (int32_t)(kind << 1);
        if (kind == 1u)      scan_row<double>(tileRow, u, ox, oy + ly, lx0, lx1, noPixels, DEPTH);
        else if (kind == 2u) scan_row<int64_t>(tileRow, u, ox, oy + ly, lx0, lx1, noPixels, DEPTH);
        // (kind 3, a masked triangle: entry_unit_masked, in a pass of its own over the round's units)
        return lx1 - lx0 + 1;
    }
}

// The units of alpha-tested triangles (kind 3) are listed and scanned apart from the others (the block-wide scan counts both
// kinds): the texture fetch of masked_row (level offsets, wrap modes, four taps, two divisions per pixel) inlined into the loop
// above put every masked instantiation of the tile kernel 37-54 VGPRs past its 128 -- 144-224 bytes of scratch per lane that
// every unit of every triangle paid for, masked or not -- and in a list of their own the masked units fill whole waves instead
// of a lane here and there.  A batch without masked entries skips the pass.
template <bool DEPTH>
__device__ __forceinline__ void entry_unit_masked(const RasterParams& p, const EntrySoA& en, unsigned long long* tile, uint32_t e, uint32_t row, uint32_t seg,
                                                  int32_t ox, int32_t oy, bool noPixels)
{
    const uint32_t box = en.w[11][e];
    const int32_t bx0 = (int32_t)(box & 63u), bx1 = (int32_t)((box >> 12) & 63u), by1 = (int32_t)((box >> 18) & 63u);
    const int32_t ly = (int32_t)row, nrows = MASKED_ROWS == 1 ? 1 : min(MASKED_ROWS, by1 - ly + 1);   // (row: the first row of the group)
    const int32_t lx0 = bx0 + (int32_t)(seg << MASKED_SEG_SHIFT), lx1 = min(bx1, lx0 + MASKED_SEG - 1);
    UnitParams u;
    u.X[0] = (int32_t)en.w[0][e]; u.X[1] = (int32_t)en.w[1][e]; u.X[2] = (int32_t)en.w[2][e];
    u.Y[0] = (int32_t)en.w[3][e]; u.Y[1] = (int32_t)en.w[4][e]; u.Y[2] = (int32_t)en.w[5][e];
    u.d0 = __uint_as_float(en.w[6][e]); u.e1 = __uint_as_float(en.w[7][e]); u.e2 = __uint_as_float(en.w[8][e]);
    u.invA = __uint_as_float(en.w[9][e]); u.payload = en.w[10][e]; u.box = 0;
    u.skind = (int32_t)((box >> 28) & 1u) | (3 << 1);
    if ((box >> 29) & 1u) masked_rows<int32_t>(p, tile + ly * TPITCH, u, en.w[12][e], ox, oy + ly, nrows, lx0, lx1, noPixels, DEPTH);   // vertices at most 64 px apart
    else                  masked_rows<int64_t>(p, tile + ly * TPITCH, u, en.w[12][e], ox, oy + ly, nrows, lx0, lx1, noPixels, DEPTH);
}

// Tile-out of a finished tile fused with its HZB reduction (single-GPU frames): every word goes to the visibility
// buffer once (16-byte coalesced stores) and, from the same LDS reads, the tile is reduced to the HZB texels it owns
// -- mips 0..5 = 32x32 ... 1x1 -- exactly as hzb_mip0_kernel + hzb_mips_kernel would from memory (edge-clamped
// source, binary16 RNE, +1 ulp on the max chain at mip 5, hzb.hlsl:67-71), plus the tile's valid-depth range.
// Wave w owns pixel rows 8w..8w+7: lanes 0-31 read two pixels of an even row, lanes 32-63 of the odd row below, so a
// mip-0 texel is one cross-half shuffle away, mips 1-2 are shuffles + the running rows of the wave, and only mips
// 3-5 (8x8 values per tile) cross waves through LDS: one barrier per tile.
// INTERIOR: every HZB texel of the tile, at every level, lies inside the chain's valid extent (all tiles but those at the
// right / bottom screen edge): no per-texel bounds arithmetic.
// What the tile-out needs of the kernel's arguments, read again from the kernel-argument segment (scalar cache) when a tile is
// finished: held in scalar registers across the scan conversion, these ~30 dwords are spilled to VGPR lanes and back.
struct TileOutParams {
    ChordHZBDesc hzbDesc;                                   // (only the fields the reduction uses are loaded)
    uint16_t* hzbMinA; uint16_t* hzbMinB; uint16_t* hzbMaxB; uint32_t* tileRange;
    uint16_t* hzbExA; uint16_t* hzbExB;
    unsigned long long* vis; float* depthOut; int32_t Wi; uint32_t tilesX; uint32_t debug; uint32_t clearTiles;
};
__device__ __forceinline__ TileOutParams load_tile_out_params()
{
    const RasterParams* q = kernel_args();
    TileOutParams t;
    t.hzbDesc.srcWidth = scalar_load(&q->hzbDesc.srcWidth); t.hzbDesc.srcHeight = scalar_load(&q->hzbDesc.srcHeight);
    t.hzbDesc.width = scalar_load(&q->hzbDesc.width); t.hzbDesc.height = scalar_load(&q->hzbDesc.height);
    t.hzbDesc.mipCount = scalar_load(&q->hzbDesc.mipCount);
#pragma unroll
    for (int l = 0; l < 6; l++) t.hzbDesc.mipOffset[l] = scalar_load(&q->hzbDesc.mipOffset[l]);
    t.hzbMinA = scalar_load(&q->hzbMinA); t.hzbMinB = scalar_load(&q->hzbMinB); t.hzbMaxB = scalar_load(&q->hzbMaxB);
    t.hzbExA = scalar_load(&q->hzbExA); t.hzbExB = scalar_load(&q->hzbExB);
    t.tileRange = scalar_load(&q->tileRange); t.vis = scalar_load(&q->vis); t.depthOut = scalar_load(&q->depthOut);
    t.Wi = scalar_load(&q->Wi); t.tilesX = scalar_load(&q->tilesX); t.debug = scalar_load(&q->debug); t.clearTiles = scalar_load(&q->clearTiles);
    return t;
}

// SH: a tile of a sharded frame -- the words go to the tile's slot of the rank's chunk (tile-linear), the HZB texels, the valid
// range and the tile's bin length to its slots of the exchange buffers.
template <bool INTERIOR, bool SH>
__device__ __forceinline__ void tile_out_and_hzb_body(const TileOutParams& p, const unsigned long long* tile, float* sM2, uint32_t* sRange,
                                                      uint32_t tileId, uint32_t slotId, uint32_t binLength, int32_t ox, int32_t oy, int32_t tw, int32_t th)
{
    static_assert(TILE == 64 && TB == 512, "wave w <-> pixel rows 8w..8w+7");
    const ChordHZBDesc& d = p.hzbDesc;
    const uint32_t tX = tileId % p.tilesX, tY = tileId / p.tilesX;
    // (the thread index goes through an empty asm: everything the reduction derives from it -- a dozen texel offsets per lane -- is
    // loop-invariant over the kernel's work items, and hoisted out of that loop it sat in registers through the whole scan conversion:
    // 8 of them in scratch in the sharded instantiation)
    uint32_t tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const uint32_t lane = tid & 63u, wave = tid >> 6, half = lane >> 5, l = lane & 31u;
    uint16_t* exA = SH && p.hzbExA ? p.hzbExA + (size_t)slotId * CHORD_HZB_SLOT_HALVES : nullptr;
    uint16_t* exB = SH ? p.hzbExB + (size_t)slotId * CHORD_HZB_FINAL_SLOT_HALVES : nullptr;
    auto vw = [&](uint32_t lv) { return min(max(1u, d.width >> lv), (((d.srcWidth - 1u) >> 1) >> lv) + 1u); };
    auto vh = [&](uint32_t lv) { return min(max(1u, d.height >> lv), (((d.srcHeight - 1u) >> 1) >> lv) + 1u); };
    // texel (lx, ly) of the tile's (32 >> lv)^2 texels at level lv
    auto put = [&](uint32_t lv, uint32_t lx, uint32_t ly, float mn, float mx) {
        const uint32_t side = 32u >> lv, gx = tX * side + lx, gy = tY * side + ly;
        if (INTERIOR || (lv < d.mipCount && gx < vw(lv) && gy < vh(lv))) {
            const uint16_t hmn = f32_to_f16(mn);
            uint16_t hmx = f32_to_f16(mx);
            if (lv == 5u) hmx = (uint16_t)(hmx + 1u);                       // storeHZBMip5
            if (SH) {
                const uint32_t o = hzb_slot_level_offset(lv) + ly * side + lx;
                if (exA) exA[o] = hmn;
                exB[o] = hmn;
                exB[CHORD_HZB_FINAL_MAX_OFFSET + o] = hmx;
            } else {
                const size_t o = d.mipOffset[lv] + (size_t)gy * max(1u, d.width >> lv) + gx;
                if (p.hzbMinA) p.hzbMinA[o] = hmn;
                p.hzbMinB[o] = hmn;
                p.hzbMaxB[o] = hmx;
            }
        }
    };
    // (word index of the tile's first word; 32 bits: a 4096 x 4096 target has 2^24 words, a sharded one at most 1.25 x that)
    const uint32_t visBase = SH ? slotId << (2 * TILE_SHIFT) : (uint32_t)oy * (uint32_t)p.Wi + (uint32_t)ox;
    const uint32_t visPitch = SH ? (uint32_t)TILE : (uint32_t)p.Wi;
    // A lane owns 2x2 pixel quads: column pair l, row pairs q = 2 half + rp (rp = 0, 1) of the wave's four.  A mip-0 texel
    // is lane-local, a mip-1 texel is the lane's two iterations and its x neighbour (DPP), a mip-2 texel adds the x
    // neighbour two over and the other half of the wave (the one cross-half exchange per lane); 8-byte LDS reads of
    // consecutive words per half-wave, 512-byte coalesced visibility stores per row.
    uint32_t rmin = 0xFFFFFFFFu, rmax = 0u;
    float m1n = 0.0f, m1x = 0.0f;
#pragma unroll
    for (uint32_t rp = 0; rp < 2u; rp++) {
        const uint32_t q = 2u * half + rp;
        const int32_t row = (int32_t)(8u * wave + 2u * q), x2 = (int32_t)(2u * l);
        const int32_t r0 = min(row, th - 1), r1 = min(row + 1, th - 1), xa = min(x2, tw - 1), xb = min(x2 + 1, tw - 1);
        const unsigned long long v00 = tile[r0 * TPITCH + xa], v01 = tile[r0 * TPITCH + xb];
        const unsigned long long v10 = tile[r1 * TPITCH + xa], v11 = tile[r1 * TPITCH + xb];
        if (!SH && p.depthOut) {
            // depth-only pass (renderMeshDepth): what leaves the tile is the D32 image -- the high halves of the words, 8 bytes
            // per lane and row (no 64-bit image, no extract pass afterwards)
            if (INTERIOR || x2 < tw) {
                float* dst = p.depthOut + (size_t)(oy + row) * (size_t)p.Wi + ox + x2;
                const float d00 = __uint_as_float((uint32_t)(v00 >> 32)), d01 = __uint_as_float((uint32_t)(v01 >> 32));
                const float d10 = __uint_as_float((uint32_t)(v10 >> 32)), d11 = __uint_as_float((uint32_t)(v11 >> 32));
                if (INTERIOR || row < th) { if (INTERIOR || x2 + 1 < tw) *reinterpret_cast<float2*>(dst) = make_float2(d00, d01); else dst[0] = d00; }
                if (INTERIOR || row + 1 < th) { if (INTERIOR || x2 + 1 < tw) *reinterpret_cast<float2*>(dst + p.Wi) = make_float2(d10, d11); else dst[p.Wi] = d10; }
            }
        } else
        if (!ABL(p, DBG_NO_VIS_STORE) && (INTERIOR || x2 < tw)) {
            unsigned long long* dst = p.vis + visBase + (size_t)row * visPitch + x2;
            if (INTERIOR || row < th) { if (INTERIOR || x2 + 1 < tw) *reinterpret_cast<ulonglong2*>(dst) = make_ulonglong2(v00, v01); else dst[0] = v00; }
            if (INTERIOR || row + 1 < th) { if (INTERIOR || x2 + 1 < tw) *reinterpret_cast<ulonglong2*>(dst + visPitch) = make_ulonglong2(v10, v11); else dst[visPitch] = v10; }
        }
        const float dq[4] = {__uint_as_float((uint32_t)(v00 >> 32)), __uint_as_float((uint32_t)(v01 >> 32)),
                             __uint_as_float((uint32_t)(v10 >> 32)), __uint_as_float((uint32_t)(v11 >> 32))};
#pragma unroll
        for (int k = 0; k < 4; k++) {                                       // valid range (hzb.hlsl:163-176), branch-free
            const uint32_t bits = __float_as_uint(dq[k]);
            const bool drawn = dq[k] > 0.0f;
            rmax = max(rmax, drawn ? bits : 0u);
            rmin = min(rmin, (drawn && dq[k] < 1.0f) ? bits : 0xFFFFFFFFu);
        }
        const float mn = fminf(fminf(dq[0], dq[1]), fminf(dq[2], dq[3])), mx = fmaxf(fmaxf(dq[0], dq[1]), fmaxf(dq[2], dq[3]));
        put(0u, l, 4u * wave + q, mn, mx);                                  // mip 0 texel (l, 4 wave + q)
        if (rp == 0u) { m1n = mn; m1x = mx; } else { m1n = fminf(m1n, mn); m1x = fmaxf(m1x, mx); }
    }
    m1n = fminf(m1n, __shfl_xor(m1n, 1, 64)); m1x = fmaxf(m1x, __shfl_xor(m1x, 1, 64));   // mip 1 texel (l/2, 2 wave + half)
    if ((l & 1u) == 0u) put(1u, l >> 1, 2u * wave + half, m1n, m1x);
    float m2n = fminf(m1n, __shfl_xor(m1n, 2, 64)), m2x = fmaxf(m1x, __shfl_xor(m1x, 2, 64));
    m2n = fminf(m2n, __shfl_xor(m2n, 32, 64)); m2x = fmaxf(m2x, __shfl_xor(m2x, 32, 64));   // mip 2 texel (l/4, wave)
    if (half == 0u && (l & 3u) == 0u) {
        put(2u, l >> 2, wave, m2n, m2x);
        sM2[wave * 8u + (l >> 2)] = m2n; sM2[64u + wave * 8u + (l >> 2)] = m2x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        rmin = min(rmin, (uint32_t)__shfl_down(rmin, off, 64));
        rmax = max(rmax, (uint32_t)__shfl_down(rmax, off, 64));
    }
    if (lane == 0u) { sRange[wave * 2u] = rmin; sRange[wave * 2u + 1u] = rmax; }
    __syncthreads();
    if (wave == 0u) {
        const uint32_t x = lane & 7u, y = lane >> 3;                        // mip 2 texel (x, y) of the tile
        float n3 = sM2[lane], x3 = sM2[64u + lane];
        n3 = fminf(n3, __shfl_xor(n3, 1, 64)); x3 = fmaxf(x3, __shfl_xor(x3, 1, 64));
        n3 = fminf(n3, __shfl_xor(n3, 8, 64)); x3 = fmaxf(x3, __shfl_xor(x3, 8, 64));
        if (!(x & 1u) && !(y & 1u)) put(3u, x >> 1, y >> 1, n3, x3);
        n3 = fminf(n3, __shfl_xor(n3, 2, 64)); x3 = fmaxf(x3, __shfl_xor(x3, 2, 64));
        n3 = fminf(n3, __shfl_xor(n3, 16, 64)); x3 = fmaxf(x3, __shfl_xor(x3, 16, 64));
        if (!(x & 3u) && !(y & 3u)) put(4u, x >> 2, y >> 2, n3, x3);
        n3 = fminf(n3, __shfl_xor(n3, 4, 64)); x3 = fmaxf(x3, __shfl_xor(x3, 4, 64));
        n3 = fminf(n3, __shfl_xor(n3, 32, 64)); x3 = fmaxf(x3, __shfl_xor(x3, 32, 64));
        if (lane == 0u) {
            put(5u, 0u, 0u, n3, x3);
            uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
            for (uint32_t w = 0; w < TB / 64u; w++) { lo = min(lo, sRange[2u * w]); hi = max(hi, sRange[2u * w + 1u]); }
            if (SH) {
                // the slot's tail: valid range of the tile, and its bin entries of the frame so far (what the tile map is balanced by)
                uint32_t* tail = reinterpret_cast<uint32_t*>(exB + CHORD_HZB_FINAL_RANGE_OFFSET);
                const uint32_t before = p.clearTiles ? 0u : tail[2];
                tail[0] = lo; tail[1] = hi; tail[2] = before + binLength; tail[3] = 0u;
            } else {
                p.tileRange[2u * tileId] = lo; p.tileRange[2u * tileId + 1u] = hi;   // reduced over tiles by hzb_tail_kernel
            }
        }
    }
}

template <bool SH>
__device__ __forceinline__ void tile_out_and_hzb(const unsigned long long* tile, float* sM2, uint32_t* sRange,
                                                 uint32_t tileId, uint32_t slotId, uint32_t binLength, int32_t ox, int32_t oy, int32_t tw, int32_t th)
{
    const TileOutParams p = load_tile_out_params();
    const ChordHZBDesc& d = p.hzbDesc;
    const uint32_t tX = tileId % p.tilesX, tY = tileId / p.tilesX;
    const uint32_t vw0 = min(max(1u, d.width), ((d.srcWidth - 1u) >> 1) + 1u), vh0 = min(max(1u, d.height), ((d.srcHeight - 1u) >> 1) + 1u);
    // (valid extents shrink by floor per level, so a tile inside level 0's is inside every level's up to 5)
    const bool interior = tw == TILE && th == TILE && d.mipCount > 5u && (tX + 1u) * 32u <= vw0 && (tY + 1u) * 32u <= vh0;
    if (interior) tile_out_and_hzb_body<true, SH>(p, tile, sM2, sRange, tileId, slotId, binLength, ox, oy, tw, th);
    else tile_out_and_hzb_body<false, SH>(p, tile, sM2, sRange, tileId, slotId, binLength, ox, oy, tw, th);
}

// Split tiles (see raster_tile_kernel): merges this slice's LDS tile into the tile's accumulation slab and draws a
// ticket; the last slice to arrive takes the merged words back into LDS and returns true.  Kept out of line so
// that its registers do not count against the scan-conversion loops (134 vs 109 VGPRs inlined).
__device__ __noinline__ bool merge_slices(unsigned long long* tile, unsigned long long* slab, uint32_t* ticket, uint32_t slices,
                                          uint32_t* sTicket, uint32_t* overflow)
{
    // all of a thread's atomics are in flight together (one round trip, not TILE*TILE/TB of them)
    constexpr uint32_t PER = TILE * TILE / TB;
    unsigned long long got[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t i = threadIdx.x + k * TB;
        const unsigned long long v = tile[(i >> TILE_SHIFT) * TPITCH + (i & (TILE - 1))];
        got[k] = v != 0ull ? atomicMax(slab + i, v) : 0ull;       // the returned value makes the wave wait for the atomic
    }
    unsigned long long seen = 0ull;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) seen |= got[k];
    if (seen == 0xFFFFFFFFFFFFFFFFull) *overflow = 8u;            // (never true: keeps the returns live)
    __syncthreads();
    if (threadIdx.x == 0) *sTicket = atomicAdd(ticket, 1u);
    __syncthreads();
    if (*sTicket != slices - 1u) return false;                    // (no thread touches LDS `tile` past this point)
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) got[k] = atomicExch(slab + threadIdx.x + k * TB, 0ull);
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t i = threadIdx.x + k * TB;
        tile[(i >> TILE_SHIFT) * TPITCH + (i & (TILE - 1))] = got[k];
    }
    __syncthreads();                                              // the caller's next pass over `tile` maps words to threads differently
    return true;
}

// DEPTH: a depth-only pass with depth clamp (shadow views): the interpolated depth is clamped to [0, 1]
// Pixel blocks of small clusters (CHORD_REC_BLOCK): the wave merges the blocks its lanes hold as bin entries, two at a time,
// every lane a word: coalesced 8-byte loads and one ds_max_u64 per non-empty word.  (header: see device_layer.h)
__device__ __forceinline__ void merge_block_word(unsigned long long* tile, unsigned long long v, uint32_t i, uint32_t hdr, uint32_t rcp)
{
    const uint32_t w = ((hdr >> 12) & 15u) + 1u;
    const uint32_t row = (i * rcp) >> 16, col = i - row * w;              // exact for i < 256, w <= 16
    const uint32_t lx = (hdr & 63u) + col, ly = ((hdr >> 6) & 63u) + row;
    if (v != 0ull && lx < (uint32_t)TILE && ly < (uint32_t)TILE) atomicMax(&tile[ly * TPITCH + lx], v);
}

__device__ __forceinline__ void merge_blocks(unsigned long long* tile, const unsigned long long* __restrict__ pool, unsigned long long todo,
                                             uint32_t name, uint32_t hdrLo, uint32_t hdrHi, uint32_t lane)
{
    // four blocks per step, a 16-byte granule per lane and block: granule g holds the block's words 2g - 1 and 2g (word -1 is
    // the header), so the first 127 words of each of the four are in flight together (a block of sub-pixel geometry has
    // 64..121) -- ~4 KB of loads outstanding per wave instead of one block's worth per memory round trip, in 16-byte
    // requests (8-byte loads run at 0.54-0.70x their rate, MI355X_MICROARCH.md).  Two groups of four alternate (A, B): the
    // loads of the next group are issued before the current one is merged, so a step's memory round trip hides behind the
    // previous step's LDS merges instead of following them -- what a workgroup that is alone on its CU needs (a rank of an
    // 8-rank frame owns 255 tiles: one workgroup per CU, nothing else to switch to).
#define MB_PICK(H, R, N, SRC)                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                                       \
        const bool any = todo != 0ull;                                                                                    \
        const int l = any ? __ffsll((long long)todo) - 1 : 0;                                                             \
        todo &= todo - 1ull;                                    /* (0 stays 0) */                                         \
        H[j] = bcast(hdrLo, l); R[j] = bcast(hdrHi, l);                                                                   \
        SRC[j] = reinterpret_cast<const ulonglong2*>(pool + (size_t)(bcast(name, l) & CHORD_REC_INDEX_MASK) * 2u);        \
        N[j] = any ? (((H[j] >> 12) & 15u) + 1u) * (((H[j] >> 16) & 15u) + 1u) : 0u;                                      \
    }
    // granule g exists iff 2g - 1 < n
#define MB_LOAD(A, N, SRC)                                                                                                \
    _Pragma("unroll") for (int j = 0; j < 4; j++) A[j] = 2u * lane <= N[j] && N[j] != 0u ? SRC[j][lane] : make_ulonglong2(0ull, 0ull);
#define MB_MERGE(A, H, R, N, SRC)                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                                       \
        if (lane != 0u) merge_block_word(tile, A[j].x, 2u * lane - 1u, H[j], R[j]);                                       \
        if (2u * lane < N[j]) merge_block_word(tile, A[j].y, 2u * lane, H[j], R[j]);                                      \
    }                                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; j++)                                                                         \
        for (uint32_t g = lane + 64u; 2u * g <= N[j]; g += 64u) {                                                         \
            const ulonglong2 v = SRC[j][g];                                                                               \
            merge_block_word(tile, v.x, 2u * g - 1u, H[j], R[j]);                                                         \
            if (2u * g < N[j]) merge_block_word(tile, v.y, 2u * g, H[j], R[j]);                                           \
        }
    if (!todo) return;
#ifndef MB_GROUPS
#define MB_GROUPS 2
#endif
#if MB_GROUPS == 3
    // three groups of four in flight (a build switch, measured in round 5 once the tile kernel had the registers: a group's loads
    // covered by the merges of the two groups before it -- the tile kernel of a rank of the 8-rank config-5 frame 0.267 -> 0.291 ms,
    // one GPU on subpixel_64m 136 -> 147 us: the third group's 20 scalars are lane spills inside the merge loop; not the default).
    // An exhausted group picks nothing (n = 0), loads nothing and merges nothing, so the rotation needs no bookkeeping.
    uint32_t hA[4], rA[4], nA[4], hB[4], rB[4], nB[4], hC[4], rC[4], nC[4];
    const ulonglong2* srcA[4];
    const ulonglong2* srcB[4];
    const ulonglong2* srcC[4];
    ulonglong2 aA[4], aB[4], aC[4];
    MB_PICK(hA, rA, nA, srcA) MB_LOAD(aA, nA, srcA)
    MB_PICK(hB, rB, nB, srcB) MB_LOAD(aB, nB, srcB)
    MB_PICK(hC, rC, nC, srcC) MB_LOAD(aC, nC, srcC)
    do {
        MB_MERGE(aA, hA, rA, nA, srcA) MB_PICK(hA, rA, nA, srcA) MB_LOAD(aA, nA, srcA)
        MB_MERGE(aB, hB, rB, nB, srcB) MB_PICK(hB, rB, nB, srcB) MB_LOAD(aB, nB, srcB)
        MB_MERGE(aC, hC, rC, nC, srcC) MB_PICK(hC, rC, nC, srcC) MB_LOAD(aC, nC, srcC)
    } while ((nA[0] | nB[0] | nC[0]) != 0u);
#else
    uint32_t hA[4], rA[4], nA[4], hB[4], rB[4], nB[4];
    const ulonglong2* srcA[4];
    const ulonglong2* srcB[4];
    ulonglong2 aA[4], aB[4];
    MB_PICK(hA, rA, nA, srcA)
    MB_LOAD(aA, nA, srcA)
    for (;;) {
        const bool moreB = todo != 0ull;
        if (moreB) { MB_PICK(hB, rB, nB, srcB) MB_LOAD(aB, nB, srcB) }
        MB_MERGE(aA, hA, rA, nA, srcA)
        if (!moreB) break;
        const bool moreA = todo != 0ull;
        if (moreA) { MB_PICK(hA, rA, nA, srcA) MB_LOAD(aA, nA, srcA) }
        MB_MERGE(aB, hB, rB, nB, srcB)
        if (!moreA) break;
    }
#endif
#undef MB_PICK
#undef MB_LOAD
#undef MB_MERGE
}

#ifndef TILE_DEEP_FETCH
#define TILE_DEEP_FETCH 0               // 1: opaque instantiations fetch records two batches ahead (two register sets, batch loop unrolled by two) -- measured in round 5, no gain (profiles/r05_tile_kernel_experiments.txt item 9); 0: the round-2 form
#endif
#ifndef TILE_MIN_BLOCKS
#define TILE_MIN_BLOCKS 4               // waves per SIMD the register allocation aims at (launch bounds: 4 -> 128 VGPRs, 6 -> 80)
#endif
template <bool SH, bool MASKED, bool DEPTH>
__global__ __launch_bounds__(TB, TILE_MIN_BLOCKS) void raster_tile_kernel(RasterParams p)
{
    __shared__ __align__(16) unsigned long long tile[TILE * TPITCH];   // 32.5 KB
    __shared__ EntrySoA prm;                                     // 26 KB: the batch's entries that need row units
    __shared__ uint32_t unitList[TILE_UNIT_CAP];                      // 16 KB: (entry | row << 9 | segment << 15) of the units of a round
    __shared__ uint32_t offs[16];                                // (scratch of the tile-out reduction)
    __shared__ uint32_t waveSums[2][TB / 64];
    __shared__ uint32_t chunkTab[64];                            // the overflow chunks this item's entries live in
    __shared__ uint32_t sTicket;
    if (ABL(p, DBG_TILE_EXIT) && !p.clearTiles) return;
    // (the item count and the block's first item are fetched together: one round trip, not two dependent ones; the
    // list has an entry for every tile, so slot 1 + blockIdx.x exists whether or not it is active)
    // (direct passes -- orderKept 2 -- have no list: item i is tile i, and the first thing a workgroup asks memory for is its tile's counter line)
    // Workgroup 0 of a launch with tileOrderNext makes the NEXT frame's schedule of this pass: the bin counts it orders are final before
    // this kernel starts, the schedule kernel's work is one workgroup's, and here it costs one of the device's 512 tile slots for the
    // first microseconds of the launch instead of a launch of its own between the binner and this kernel (launch_raster).
    // The kernel sits at its scalar-register limit, so where that skew of the workgroup index is needed decides how it is had:
    //  * in front of the loop -- the workgroup's first item, on every tile's chain of dependent fetches -- from the argument as the
    //    launch handed it over (dead once the loop is entered).  Re-read there it was a scalar load and a wait in front of every tile's
    //    first fetch, and of every untouched tile's workgroup of a direct pass: config 3 0.1597 -> 0.1585 ms, config 4 0.3886 -> 0.3871;
    //  * at the loop's end -- the stride, off every chain -- re-read from the kernel-argument segment (wgs): held across the body, skew
    //    or stride is one more scalar to keep through the scan conversion and around the call of merge_slices.  Measured both ways:
    //    48 bytes of scratch in a kernel that had none; and with a skew of 1 in EVERY launch (a constant: workgroup 0 of a launch without
    //    a schedule to make just ends) and the stride gridDim.x - 1 held instead, config 3 0.1595 -> 0.1607 ms.  (And the other way --
    //    item, item count and stride ALL worked out again at the loop's end, only the index carried: 0.1582 -> 0.1591.)
#define TILE_SKEW (scalar_load(&kernel_args()->tileOrderNext) != nullptr ? 1u : 0u)
#define wgs (gridDim.x - TILE_SKEW)
    const uint32_t tilesAll = p.tilesX * p.tilesY;
    const uint32_t wg0 = blockIdx.x - (p.tileOrderNext != nullptr ? 1u : 0u);
    if (p.tileOrderNext && blockIdx.x == 0u) {
#if TILE_MAKE_NEXT
        tile_order_next_part(kernel_args());
#endif
        return;
    }
    const bool direct = p.orderKept == 2u;
    // (the list is asked under `!direct`: asked for before anything branches -- so that the arguments it needs come in one scalar load
    // with the others -- a direct pass's workgroups, 1 604 of 2 040 of them with nothing else to do, waited for a fetch they drop:
    // config 3 0.1581 -> 0.1593 ms, while config 4, which has no direct pass, gained 0.5 %)
    uint2 firstItem = make_uint2(wg0, 0u);
    uint32_t active = tilesAll;
    if (!direct) { firstItem = p.tileOrder[1u + min(wg0, tilesAll - 1u)]; active = p.tileOrder[0].x; }
    // (bit 31 of the loop's index: not the workgroup's first item -- asked of gridDim.x, "first" was a scalar load from the dispatch
    // packet and a wait at the top of every item)
    for (uint32_t oiw = wg0; (oiw & 0x7FFFFFFFu) < active; oiw = ((oiw & 0x7FFFFFFFu) + wgs) | 0x80000000u) {
    const uint32_t oi = oiw & 0x7FFFFFFFu;
    // (the thread index of this work item goes through an empty asm: whatever the body derives from it is invariant over the
    // item loop, and hoisted out of it those values -- offsets, masks, lane roles -- sat in registers across the whole kernel)
    uint32_t tix = threadIdx.x;
    asm volatile("" : "+v"(tix));
    const uint2 itemCount = (oiw >> 31) == 0u ? firstItem : (direct ? make_uint2(oi, 0u) : p.tileOrder[1u + oi]);
    const uint32_t item = itemCount.x;
    const uint32_t tileId = item & 0xFFFu, slice = (item >> 12) & 0x3FFu, slices = (item >> 22) + 1u;
    // (kept order: the tile's counter line is read here, and -- the address needs the tile only -- a whole tile's first bin entries
    // beside it, so that the chain item -> bin entry -> record does not grow by the round trip for the length)
    const uint32_t* bin = p.tileBins + (size_t)tileId * p.binCap;
    const bool entryThread = TILE_BATCH == TB || tix < TILE_BATCH;   // (a batch is TILE_BATCH bin entries, one per thread of the first waves)
    uint32_t countWord = itemCount.y, wordSpec = 0xFFFFFFFFu;
    if (p.orderKept) {
        // (the bin words are asked for whatever the item is -- a slice's first entries are elsewhere and these are dropped --, and the
        // counter line with a SCALAR load (words 0 and 1: final since the binner ended; the kernel's own atomics touch word 2 only):
        // under `slices == 1`, and with the line in a vector register the allocator used again for the bin words' address, the compiler
        // put the fetch it was meant to travel beside behind the wait for the line -- a round trip in front of every tile)
        if (entryThread) wordSpec = bin[tix];
        const unsigned long long cnt = scalar_load(reinterpret_cast<const unsigned long long*>(&p.tileCount[(size_t)tileId * TC_STRIDE]));
        countWord = min((uint32_t)cnt, bin_capacity(p)) | ((uint32_t)(cnt >> 32) ? 0x80000000u : 0u);
    }
    const uint32_t nAll = countWord & 0x3FFFFFFFu;                // (already clamped to the bin capacity)
    if (ABL(p, DBG_SKIP_LIGHT) && nAll < 64u) continue;
    if (ABL(p, DBG_SKIP_HEAVY) && nAll >= 64u) continue;
    if (direct) {
        // a later pass of the frame touches a fraction of the tiles (config 3: 436 of 2 040): the others' workgroups end here, one round
        // trip after their start, and the touched ones have saved the trip to a work list -- what raster_tile_order_kernel cost the
        // pass was its place in the chain of dependent launches.  No slices, no order: a bin long enough to want them says so to the
        // host, and the pass gets its schedule back from the next frame on (a choice of speed: this workgroup does the whole bin).
        if (p.heavyHint && tix == 0u) {
            if (nAll > p.tileSplitMin) p.heavyHint[0] = p.binStamp;
            if (oi == 0u) p.heavyHint[*p.count > TILE_DIRECT_MAX_CLUSTERS ? 0 : 2] = p.binStamp;     // (the pass's cluster count: one workgroup reports it)
        }
        if (nAll == 0u) continue;
    }
    if (p.orderAll && !p.clearTiles && nAll == 0u) continue;     // (a later pass's schedule that lists every tile: the untouched ones end here)
    const bool hasBlocks = (countWord >> 31) != 0u;               // (the order kernel saw pixel blocks in the tile's bin)
    const bool preloaded = !CHORD_MASKED_FUSED && p.clearTiles && (countWord & 0x40000000u) != 0u;   // the masked pass wrote this tile (first pass of a frame)
    const int32_t tinyArea = nAll >= TINY_DENSE_MIN ? TINY_AREA_DENSE : TINY_AREA;
    // entries [lo, n) of the bin are this item's
    // (equal parts rounded up to whole batches: with the schedule's slice count -- ceil(entries / slice length), slice length
    // TILE_SLICE, or shorter in a pass with fewer tiles than the device has slots -- that is slices of the schedule's length)
    const uint32_t per = slices > 1u ? ((nAll + slices - 1u) / slices + TB - 1u) & ~(TB - 1u) : nAll;
    const uint32_t lo = slices > 1u ? min(nAll, slice * per) : 0u;
    const uint32_t n = ABL(p, DBG_NO_BATCH) ? lo : (slices > 1u ? min(nAll, lo + per) : nAll);
    const bool prof = RASTER_PROFILE && (p.debug & DBG_TILE_CLOCKS) != 0;
    const unsigned long long t0 = prof ? wall_clock64() : 0ull;
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tp = t0;
    uint32_t cUnits = 0, cUnitIters = 0, cTiny = 0, cTinyIters = 0;      // (profile only)
    // (profile build: a phase ends at a barrier of its own, and what a wave spends AT that barrier -- from its arrival to the release, i.e.
    // waiting for the workgroup's slowest wave -- is summed per phase: bw[])
    unsigned long long bw[6] = {0, 0, 0, 0, 0, 0};
#define PHASE(i) do { if (prof) { const unsigned long long ta = wall_clock64(); __syncthreads(); const unsigned long long tn = wall_clock64(); bw[i] += tn - ta; ph[i] += tn - tp; tp = tn; } } while (0)
    const int32_t ox = (int32_t)(tileId % p.tilesX) * TILE, oy = (int32_t)(tileId / p.tilesX) * TILE;
    const int32_t tw = min(TILE, p.Wi - ox), th = min(TILE, p.Hi - oy);
    const bool noPixels = ABL(p, DBG_NO_PIXELS);
    // where the tile's words live: row-major in the image, or (sharded frames: SH) tile-linear in the tile's slot of the rank's
    // chunk -- 64 rows of 64 words, the whole slot also for a tile cut by the screen edge.  (Only owned tiles are work items.)
    // (the slot is read again from the tile map where it is needed -- tile-in and tile-out, a scalar-cache hit -- rather than held
    // across the scan conversion: the kernel sits at its 128 VGPRs / 102 SGPRs)
    auto slot_of_tile = [&]() -> uint32_t { return SH ? __builtin_amdgcn_readfirstlane(scalar_load(&kernel_args()->shard.tileSlot)[tileId]) : 0u; };
    // word index of the tile's first word (32 bits: a 4096 x 4096 target has 2^24 words, a sharded one at most 1.25 x that) and its row pitch
    auto vis_base = [&]() -> uint32_t { return SH ? slot_of_tile() << (2 * TILE_SHIFT) : (uint32_t)oy * (uint32_t)p.Wi + (uint32_t)ox; };
    const uint32_t visPitch = SH ? (uint32_t)TILE : (uint32_t)p.Wi;

    // The item's first bin entries are requested BEFORE the tile-in, and the records they name before its barrier: the chain
    // item -> bin entry -> record is what a light tile waits for (profiles/r03_tile_profile_config3.txt: a third of a tile's time),
    // and the tile-in -- 33 KB of LDS stores, or the tile's words from memory -- depends on none of it.  (Bins that continue in
    // pool chunks name them through chunkTab, which is filled behind the barrier: they keep the old order.)
    const bool early = n <= p.binCap && !hasBlocks;               // (uniform)
    const uint32_t k0 = lo + tix;
    const uint32_t word0 = (early && entryThread && k0 < n) ? ((p.orderKept && slices == 1u) ? wordSpec : bin[k0]) : 0xFFFFFFFFu;

    // ---- tile in: zero (first pass: this is the clear; un-fused later passes merge by max at tile-out), or the
    //      current words when a later pass must leave the finished tile in LDS for the fused HZB reduction ----
    const bool rmw = (p.hzbFused && !p.clearTiles) || preloaded;
    // slices of a split tile start from zero; when the tile must leave this kernel complete (first pass, fused HZB)
    // they meet in memory and the last one to arrive merges them (below)
    const bool mergeSlices = slices > 1u && (p.clearTiles || rmw);
    if (!(rmw && !mergeSlices)) {
        // (the whole array incl. the padding words, 16 bytes per store)
        static_assert((TILE * TPITCH) % 2 == 0, "whole 16-byte words");
        for (uint32_t i = tix; i < TILE * TPITCH / 2; i += TB) reinterpret_cast<ulonglong2*>(tile)[i] = make_ulonglong2(0ull, 0ull);
    } else
    for (uint32_t i = tix, visBase = vis_base(); i < TILE * TILE / 2; i += TB) {
        const int32_t ly = (int32_t)(i >> (TILE_SHIFT - 1)), lx = (int32_t)(i & (TILE / 2 - 1)) * 2;
        ulonglong2 v = make_ulonglong2(0ull, 0ull);
        if (ly < th && lx < tw) {
            const unsigned long long* src = scalar_load(&kernel_args()->vis) + (size_t)(visBase + (uint32_t)ly * visPitch + (uint32_t)lx);
            if (lx + 1 < tw) v = *reinterpret_cast<const ulonglong2*>(src);
            else v.x = src[0];
        }
        tile[ly * TPITCH + lx] = v.x; tile[ly * TPITCH + lx + 1] = v.y;
    }
    // a record in flight: 32 or 48 raw bytes (which, says bit 31 of its name) in three named registers quads (an
    // array or a struct passed by reference ends up in scratch memory), expanded when its batch is processed
#define FETCH_REC(gi, a, b, c)                                                                       \
    do {                                                                                             \
        if ((gi) & CHORD_REC_WIDE) {                                                                 \
            const uint4* src_ = reinterpret_cast<const uint4*>(&p.tris[(gi) & CHORD_REC_WIDE_INDEX]); \
            a = src_[0]; b = src_[1]; c = src_[2];                                                   \
        } else {                                                                                     \
            const uint4* src_ = reinterpret_cast<const uint4*>(&p.trisC[(gi)]);                      \
            a = src_[0]; b = src_[1];                                                                \
        }                                                                                            \
    } while (0)
    uint32_t idxNext = 0xFFFFFFFFu, nameNext = 0xFFFFFFFFu;
    uint4 nq0 = make_uint4(0, 0, 0, 0), nq1 = nq0, nq2 = nq0;
    // (pixel blocks have a pass of their own below, alpha-tested triangles a kernel of their own: neither is a record of this pipeline)
    auto record_name = [](uint32_t w) -> uint32_t { return (w >= CHORD_REC_BLOCK || (!(MASKED && CHORD_MASKED_FUSED) && (w & 0xE0000000u) == CHORD_REC_MASKED)) ? 0xFFFFFFFFu : w; };
    if (early) {
        nameNext = record_name(word0);                                          // (binEntry of an entry of the fixed bin)
        if (nameNext != 0xFFFFFFFFu) FETCH_REC(nameNext, nq0, nq1, nq2);        // record of batch 0
        if (entryThread && k0 + TILE_BATCH < n) idxNext = record_name(bin[k0 + TILE_BATCH]);   // bin entry of batch 1
    }
    __syncthreads();
    PHASE(0);

    // ---- scan-convert the bin, 256 entries per batch -------------------------------------------
    // Software pipeline over the two dependent fetches of a batch (bin entry -> 48-byte record): the
    // record of batch b+1 and the bin entry of batch b+2 are in flight while batch b is scan-converted.
    // Overflow chunks the item's entries [lo, n) live in: a WINDOW of 64 chunk names in LDS (chunk c at chunkTab[c & 63], names of
    // chunks [chunk0, chunkEnd); an entry of another pass or a failed allocation reads as invalid and its entries are skipped).
    // A fresh schedule cuts a bin into slices of at most ~33 chunks, so the window holds a whole item; an item of a KEPT schedule
    // (orderKept) was cut for an earlier frame's bin and may now span any number of chunks -- a camera cut into a hotspot view --,
    // so the loops below slide the window along (window_advance): 32 chunks at a time once the loop has passed the window's middle.
    // Their look-ahead is at most three batches = 1.5 chunks, and names below the current batch's chunk are dead.
    uint32_t chunk0 = 0, chunkEnd = 0;
    const uint32_t chunksAll = n > p.binCap ? (n - p.binCap + CHORD_BIN_CHUNK - 1u) >> CHORD_BIN_CHUNK_SHIFT : 0u;   // chunks [0, chunksAll) hold entries below n
    auto window_load = [&](uint32_t from, uint32_t to) {
        const RasterParams* q = kernel_args();                                // (long bins only: not worth registers across the kernel)
        for (uint32_t j = from + tix; j < to; j += TB) {
            const unsigned long long e = scalar_load(&q->binChunkTab)[(size_t)tileId * scalar_load(&q->binMaxChunks) + j];
            chunkTab[j & 63u] = (uint32_t)(e >> 32) == scalar_load(&q->binStamp) ? (uint32_t)e : CHORD_BIN_CHUNK_INVALID;
        }
    };
    auto window_reset = [&]() {                                  // (callers: n > binCap; every thread of the workgroup)
        __syncthreads();                                          // (readers of an earlier window are done)
        chunk0 = lo > p.binCap ? (lo - p.binCap) >> CHORD_BIN_CHUNK_SHIFT : 0u;
        chunkEnd = min(chunksAll, chunk0 + 64u);
        window_load(chunk0, chunkEnd);
        __syncthreads();
    };
    auto window_advance = [&](uint32_t base) {                   // top of a loop iteration over entries [base, ...): uniform
        if (chunkEnd >= chunksAll || base < p.binCap + ((chunk0 + 32u) << CHORD_BIN_CHUNK_SHIFT)) return;
        __syncthreads();
        const uint32_t to = min(chunksAll, chunkEnd + 32u);
        window_load(chunkEnd, to);
        chunk0 += 32u; chunkEnd = to;
        __syncthreads();
    };
    if (n > p.binCap) window_reset();
    const uint32_t wideLimit = p.triCap * CHORD_LIST_SHARDS, compactLimit = p.triCapC * CHORD_LIST_SHARDS;
    auto binWord = [&](uint32_t k) -> uint32_t {               // bin entry k as stored, ~0u = none
        if (k < p.binCap) return bin[k];
        const uint32_t o = k - p.binCap, c = o >> CHORD_BIN_CHUNK_SHIFT;
        const uint32_t id = (c - chunk0) < (chunkEnd - chunk0) ? chunkTab[c & 63u] : CHORD_BIN_CHUNK_INVALID;
        if (id == CHORD_BIN_CHUNK_INVALID) return 0xFFFFFFFFu;
        return p.binPool[(size_t)id * CHORD_BIN_CHUNK + (o & (CHORD_BIN_CHUNK - 1u))];
    };
    // ---- pixel blocks of small clusters first: their own pass over the item's entries (nothing of the triangle
    //      pipeline below is live here; tiles without blocks -- word 2 of the tile's counter line -- skip it) ----------
    // (the block pool and its size are read from the kernel arguments here: frames without blocks -- all but the densest -- do
    // not hold them in registers)
    if (hasBlocks && !noPixels) {
        const uint32_t blockCap = scalar_load(&kernel_args()->blockCap);
        const unsigned long long* blockPool = scalar_load(&kernel_args()->blockPool);
        const uint32_t blockLimit = blockCap * CHORD_LIST_SHARDS;
        uint32_t giNext = lo + tix < n ? binWord(lo + tix) : 0xFFFFFFFFu;
        for (uint32_t base = lo; base < n; base += TB) {
            window_advance(base);
            const uint32_t gi = giNext;
            giNext = base + TB + tix < n ? binWord(base + TB + tix) : 0xFFFFFFFFu;
            // (never a block outside the pool: a slot drawn but not written after a reported overflow holds anything)
            const bool isBlock = gi != 0xFFFFFFFFu && gi >= CHORD_REC_BLOCK && (gi & CHORD_REC_INDEX_MASK) < blockLimit;
            uint2 hdr = make_uint2(0u, 0u);
            if (isBlock) hdr = *reinterpret_cast<const uint2*>(blockPool + (size_t)(gi & CHORD_REC_INDEX_MASK) * 2u);
            merge_blocks(tile, blockPool, __ballot(isBlock), gi, hdr.x, hdr.y, tix & 63u);
        }
        // (the triangle pass below walks the same entries from `lo` again: a window that was slid along goes back)
        if (n > p.binCap && chunk0 != (lo > p.binCap ? (lo - p.binCap) >> CHORD_BIN_CHUNK_SHIFT : 0u)) window_reset();
    }
    auto binEntry = [&](uint32_t k) -> uint32_t {              // record name of bin entry k, ~0u = none (or a pixel block)
        const uint32_t gi = record_name(binWord(k));           // (~0u stays ~0u)
        if (gi == 0xFFFFFFFFu || k < p.binCap) return gi;
        const bool okIdx = (gi & CHORD_REC_WIDE) ? (gi & CHORD_REC_WIDE_INDEX) < wideLimit : gi < compactLimit;
        return okIdx ? gi : 0xFFFFFFFFu;                       // (only after a reported overflow)
    };
    if (!early) {
        idxNext = entryThread && k0 < n ? binEntry(k0) : 0xFFFFFFFFu;            // bin entry of batch 0
        nameNext = idxNext;
        if (nameNext != 0xFFFFFFFFu) FETCH_REC(nameNext, nq0, nq1, nq2);         // record of batch 0
        idxNext = entryThread && k0 + TILE_BATCH < n ? binEntry(k0 + TILE_BATCH) : 0xFFFFFFFFu;   // bin entry of batch 1
    }
    // One batch in two parts, so that the caller can re-load the record registers in between (by value: a record handed over by
    // reference ends up in scratch memory).  batch_setup: the records q0..q2 named `name` of the batch's entries (one per entry
    // thread) are set up -- tiny triangles scanned at once, the others stored as entries; returns the thread's unit count.
    // batch_units: block-wide scan, unit lists, row units.
    auto batch_setup = [&](const uint4 q0, const uint4 q1, const uint4 q2, const uint32_t name) -> uint32_t {
        const bool have = name != 0xFFFFFFFFu && !ABL(p, DBG_NO_ENTRY);
        if (prof) { volatile uint32_t sink = q1.w; (void)sink; }
        PHASE(1);
        uint32_t rows = 0;
        if (have) {
            TriSetup ts;
            // (fields are picked out of the raw dwords by value: a pointer cast would put the record in scratch memory)
            if (name & CHORD_REC_WIDE) {
                TriRec w;
                w.X[0] = (int32_t)q0.x; w.X[1] = (int32_t)q0.y; w.X[2] = (int32_t)q0.z; w.Y[0] = (int32_t)q0.w;
                w.Y[1] = (int32_t)q1.x; w.Y[2] = (int32_t)q1.y;
                w.d[0] = __uint_as_float(q1.z); w.d[1] = __uint_as_float(q1.w); w.d[2] = __uint_as_float(q2.x);
                w.payload = q2.y; w.twoSided = q2.z; w.pad = q2.w;
                tri_setup_from_record(ts, w, p.Wi, p.Hi);
            } else {
                TriRecC cr;
                cr.X0 = (int32_t)q0.x; cr.Y0 = (int32_t)q0.y;
                cr.dX1 = (int16_t)(q0.z & 0xFFFFu); cr.dY1 = (int16_t)(q0.z >> 16);
                cr.dX2 = (int16_t)(q0.w & 0xFFFFu); cr.dY2 = (int16_t)(q0.w >> 16);
                cr.d[0] = __uint_as_float(q1.x); cr.d[1] = __uint_as_float(q1.y); cr.d[2] = __uint_as_float(q1.z);
                cr.payload = q1.w;
                tri_setup_from_compact(ts, cr, p.Wi, p.Hi);
            }
            {
                const int32_t x0 = max(ts.px0, ox), y0 = max(ts.py0, oy);
                const int32_t x1 = min(ts.px1, ox + tw - 1), y1 = min(ts.py1, oy + th - 1);
                if (x1 >= x0 && y1 >= y0) {
                    const bool narrow = narrow_extent(ts);
                    const bool maskedRec = MASKED && (name & CHORD_REC_WIDE) && (q2.z & 4u);   // a TriRecMaskExt follows the record
                    if (narrow && !maskedRec && (x1 - x0 + 1) * (y1 - y0 + 1) <= tinyArea) {
                        if (!ABL(p, DBG_NO_TINY)) tile_raster_narrow<TPITCH>(tile, ts, ox, oy, x0, y0, x1, y1, noPixels, DEPTH);
                        if (prof) { cTiny++; cTinyIters += (uint32_t)((x1 - x0 + 1) * (y1 - y0 + 1)); }
                    } else {
                        rows = entry_store(prm, tix, ts, narrow, ox, oy, x0, y0, x1, y1, maskedRec, name & CHORD_REC_WIDE_INDEX);   // (units, not rows)
                    }
                }
            }
        }
        return rows;
    };
    auto batch_units = [&](const uint32_t rows, const uint32_t batchNo) {
        PHASE(2);
        uint32_t total;
        // (the wave sums alternate between two buffers: a batch without units then needs no barrier but the scan's own)
        // (MASKED: one scan for both populations -- units of opaque entries in the low half, of masked entries (kind 3) in the high
        // half: a batch has at most 512 x 64 units)
        const bool mine3 = MASKED && rows && ((prm.w[11][tix] >> EF_KIND_SHIFT) & 3u) == 3u;
        const uint32_t rowsN = mine3 ? 0u : rows, rowsM = mine3 ? rows : 0u;
        const uint32_t offAll = block_scan_tb(rowsN | (rowsM << 16), waveSums[batchNo & 1u], &total);
        const uint32_t offN = offAll & 0xFFFFu, offM = offAll >> 16, totalM = MASKED ? total >> 16 : 0u;
        total &= 0xFFFFu;
        PHASE(3);
        // my units [off, off + n) of a population, cut to the round's window [r0, r0 + TILE_UNIT_CAP): one LDS word each
        auto list_units = [&](uint32_t off, uint32_t n, uint32_t r0) {
            if (!n) return;
            const uint32_t box = prm.w[11][tix];
            const uint32_t nseg = ((((box >> 12) & 63u) - (box & 63u)) >> SEG_SHIFT) + 1u, y0l = (box >> 6) & 63u;
            const uint32_t lo2 = max(off, r0), hi2 = min(off + n, r0 + TILE_UNIT_CAP);
            if (lo2 < hi2) {
                uint32_t j = lo2 - off;
                uint32_t row = y0l + (nseg == 1u ? j : nseg == 2u ? j >> 1 : nseg == 4u ? j >> 2 : j / 3u);
                uint32_t seg = nseg == 1u ? 0u : nseg == 2u ? (j & 1u) : nseg == 4u ? (j & 3u) : j % 3u;
                for (uint32_t u = lo2; u < hi2; u++) {
                    unitList[u - r0] = tix | (row << 9) | (seg << 15);
                    if (++seg == nseg) { seg = 0u; row++; }
                }
            }
        };
        // rounds of TILE_UNIT_CAP units: every entry thread lists its units of the round, then every thread takes units TB apart -- a
        // unit finds its entry with ONE read instead of a 9-step binary search
        for (uint32_t r0 = 0; r0 < total; r0 += TILE_UNIT_CAP) {
            list_units(offN, rowsN, r0);
            __syncthreads();
            const uint32_t nr = min(total - r0, (uint32_t)TILE_UNIT_CAP);
            for (uint32_t ui = tix; ui < nr; ui += TB) {
                const uint32_t d = unitList[ui];
                if (ABL(p, DBG_NO_UNITS)) continue;
                const int32_t trips = entry_unit<MASKED, DEPTH>(p, prm, tile, d & 511u, (d >> 9) & 63u, (d >> 15) & 3u, ox, oy, noPixels);
                if (prof) { cUnits++; cUnitIters += (uint32_t)trips; }
            }
            if (r0 + TILE_UNIT_CAP < total || totalM) __syncthreads(); // the list is rewritten by the next round
        }
        for (uint32_t r0 = 0; r0 < totalM; r0 += TILE_UNIT_CAP) {      // (MASKED only) the alpha-tested triangles' units
            if (rowsM) {
                const uint32_t box = prm.w[11][tix];
                const uint32_t nseg = ((((box >> 12) & 63u) - (box & 63u)) >> MASKED_SEG_SHIFT) + 1u, y0l = (box >> 6) & 63u;
                const uint32_t lo2 = max(offM, r0), hi2 = min(offM + rowsM, r0 + TILE_UNIT_CAP);
                if (lo2 < hi2) {
                    const uint32_t j = lo2 - offM, g = j / nseg;
                    uint32_t row = y0l + g * MASKED_ROWS, seg = j - g * nseg;
                    for (uint32_t u = lo2; u < hi2; u++) {
                        unitList[u - r0] = tix | (row << 9) | (seg << 15);
                        if (++seg == nseg) { seg = 0u; row += MASKED_ROWS; }
                    }
                }
            }
            __syncthreads();
            const uint32_t nr = min(totalM - r0, (uint32_t)TILE_UNIT_CAP);
            for (uint32_t ui = tix; ui < nr; ui += TB) {
                const uint32_t d = unitList[ui];
                entry_unit_masked<DEPTH>(p, prm, tile, d & 511u, (d >> 9) & 63u, (d >> 15) & 7u, ox, oy, noPixels);
            }
            if (r0 + TILE_UNIT_CAP < totalM) __syncthreads();
        }
        total |= totalM;
        if (total) __syncthreads();                               // prm / the unit list are rewritten by the next batch
        PHASE(4);
    };
    uint32_t batchNo = 0;
    if constexpr (TILE_DEEP_FETCH && !MASKED) {
        // (a build switch, off: measured and not kept)  Records TWO batches ahead (opaque instantiations: 103 VGPRs leave room for a
        // second set of record registers; the masked ones sit at 128).  The phase clocks of the profiling build put 44 % of the time
        // of config 4's dense tiles into the wait for a batch's records (profiles/r05_tile_profile_config4.txt) -- with two batches
        // of cover instead of one the product kernel takes exactly as long (config 4: tile 161.6 vs 161.3 us per frame; config 3
        // +1 %: 115 VGPRs and 30 % more code), so that wait is the profiling build's own barrier in front of every clock, not latency
        // the product kernel is exposed to.  Two register sets alternate, unrolled by two: a copy from set to set would make the
        // compiler wait for the younger load at the copy.  Set A holds the record of batch 0 (fetched above), set B takes batch
        // 1's, the bin entries run three batches ahead.
        uint32_t nameA = nameNext, nameB = idxNext;
        uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0, b2 = b0;
        if (nameB != 0xFFFFFFFFu) FETCH_REC(nameB, b0, b1, b2);                  // record of batch 1
        uint32_t idx = entryThread && k0 + 2u * TILE_BATCH < n ? binEntry(k0 + 2u * TILE_BATCH) : 0xFFFFFFFFu;   // bin entry of batch 2
        for (uint32_t base = lo; base < n;) {
            window_advance(base);
            uint32_t rows = batch_setup(nq0, nq1, nq2, nameA);
            nameA = idx;
            if (nameA != 0xFFFFFFFFu) FETCH_REC(nameA, nq0, nq1, nq2);          // record of batch b + 2 into the set just read
            idx = entryThread && base + tix + 3u * TILE_BATCH < n ? binEntry(base + tix + 3u * TILE_BATCH) : 0xFFFFFFFFu;
            batch_units(rows, batchNo);
            base += TILE_BATCH; batchNo++;
            if (base >= n) break;
            window_advance(base);
            rows = batch_setup(b0, b1, b2, nameB);
            nameB = idx;
            if (nameB != 0xFFFFFFFFu) FETCH_REC(nameB, b0, b1, b2);
            idx = entryThread && base + tix + 3u * TILE_BATCH < n ? binEntry(base + tix + 3u * TILE_BATCH) : 0xFFFFFFFFu;
            batch_units(rows, batchNo);
            base += TILE_BATCH; batchNo++;
        }
    } else
    for (uint32_t base = lo; base < n; base += TILE_BATCH, batchNo++) {
        window_advance(base);
        const uint32_t k = base + tix;
        const uint4 q0 = nq0, q1 = nq1, q2 = nq2;
        const uint32_t name = nameNext;
        nameNext = idxNext;
        if (nameNext != 0xFFFFFFFFu) FETCH_REC(nameNext, nq0, nq1, nq2);         // record of the next batch
        idxNext = entryThread && k + 2u * TILE_BATCH < n ? binEntry(k + 2u * TILE_BATCH) : 0xFFFFFFFFu;   // bin entry of the batch after
        batch_units(batch_setup(q0, q1, q2, name), batchNo);
    }
    __syncthreads();

    // ---- split tile: the slices meet in the tile's accumulation slab (device-scope atomic max of the pixels a
    //      slice touched); the last slice to arrive takes the merged words back -- with an atomic exchange that
    //      also restores the slab's invariant (all zero between uses).  Only atomics touch the slab, and a slice
    //      draws its ticket after every one of its atomics has returned, so no cache-wide release/acquire is
    //      needed (an agent-scope fence writes back the whole L2 of the XCD: measured 2x slower here). ----------
    if (mergeSlices) {
        const RasterParams* q = kernel_args();                                    // (split tiles only)
        if (!merge_slices(tile, scalar_load(&q->tileSlabs) + (size_t)tileId * (TILE * TILE), &scalar_load(&q->tileCount)[(size_t)tileId * TC_STRIDE + TC_TICKET],
                          slices, &sTicket, &scalar_load(&q->counters)->overflow)) continue;   // not the last slice: done
        if (rmw) {
            for (uint32_t i = tix, visBase = vis_base(); i < TILE * TILE / 2; i += TB) {
                const int32_t ly = (int32_t)(i >> (TILE_SHIFT - 1)), lx = (int32_t)(i & (TILE / 2 - 1)) * 2;
                if (ly >= th || lx >= tw) continue;
                const unsigned long long* src = p.vis + (size_t)(visBase + (uint32_t)ly * visPitch + (uint32_t)lx);
                tile[ly * TPITCH + lx] = max(tile[ly * TPITCH + lx], src[0]);
                if (lx + 1 < tw) tile[ly * TPITCH + lx + 1] = max(tile[ly * TPITCH + lx + 1], src[1]);
            }
        }
        __syncthreads();
    }

    // ---- tile out ------------------------------------------------------------------------------------
    if (ABL(p, DBG_NO_OUT)) {
    } else if (p.hzbFused && !ABL(p, DBG_NO_HZB)) {
        // single-GPU frame: the whole tile goes out (first pass: this is the clear; later passes loaded it), and its
        // HZB texels with it; the batch buffers are free now and hold the cross-wave part of the reduction
        tile_out_and_hzb<SH>(tile, reinterpret_cast<float*>(&prm.w[0][0]), offs, tileId, slot_of_tile(), nAll, ox, oy, tw, th);
    } else if (p.clearTiles || p.hzbFused) {
        // first pass of the frame: every word is written (16-byte coalesced stores); this is the clear
        for (uint32_t i = tix, visBase = vis_base(); i < TILE * TILE / 2; i += TB) {
            const int32_t ly = (int32_t)(i >> (TILE_SHIFT - 1)), lx = (int32_t)(i & (TILE / 2 - 1)) * 2;
            if (ly >= th || lx >= tw) continue;
            const ulonglong2 v = make_ulonglong2(tile[ly * TPITCH + lx], tile[ly * TPITCH + lx + 1]);
            unsigned long long* dst = p.vis + (size_t)(visBase + (uint32_t)ly * visPitch + (uint32_t)lx);
            if (lx + 1 < tw) *reinterpret_cast<ulonglong2*>(dst) = v;
            else dst[0] = v.x;
        }
    } else {
        // later passes: only the pixels this pass touched are merged, with a row-coalesced global atomicMax
        // (8 lanes per 64-byte line) — no read-modify-write of the whole tile
        for (uint32_t i = tix, visBase = vis_base(); i < TILE * TILE; i += TB) {
            const int32_t ly = (int32_t)(i >> TILE_SHIFT), lx = (int32_t)(i & (TILE - 1));
            const unsigned long long v = tile[ly * TPITCH + lx];
            if (v != 0ull && ly < th && lx < tw) atomicMax(p.vis + (size_t)(visBase + (uint32_t)ly * visPitch + (uint32_t)lx), v);
        }
    }
    PHASE(5);
    if (prof) {
        if (tix == 0) {
            p.tileClocks[tileId] = wall_clock64() - t0;
            for (int i = 0; i < 6; i++) p.tilePhase[(size_t)tileId * 8 + i] = ph[i];
            p.tilePhase[(size_t)tileId * 8 + 6] = 0ull; p.tilePhase[(size_t)tileId * 8 + 7] = 0ull;
        }
        __syncthreads();
        // work counters of the tile: units << 32 | row-loop trips, tiny triangles << 32 | their bbox pixels
        atomicAdd(&p.tilePhase[(size_t)tileId * 8 + 6], ((unsigned long long)cUnits << 32) | cUnitIters);
        atomicAdd(&p.tilePhase[(size_t)tileId * 8 + 7], ((unsigned long long)cTiny << 32) | cTinyIters);
        // the eight waves' time at the phase barriers, in the high halves of the phase words (ticks of 10 ns: a phase stays far below 2^32)
        if ((tix & 63u) == 0u) for (int i = 0; i < 6; i++) atomicAdd(&p.tilePhase[(size_t)tileId * 8 + i], bw[i] << 32);
    }
    __syncthreads();                                              // the LDS tile is reused by the next iteration
    }
}
#undef wgs
#undef TILE_SKEW

// ---- alpha-tested (masked) triangles: a pass of their own (mesh_raster.hlsl:34-38,107-112,198-204) ---------------------------------
// Until round 4 the masked row units ran inside raster_tile_kernel: every masked instantiation of it sat at 128 VGPRs with
// 144-240 bytes of scratch per lane (the texture fetch next to the whole opaque batch pipeline), the opaque triangles of a masked
// scene paid for the pass structure (no tiny-triangle path, an extension fetch per unit: 37 us per frame on street_4k_masked), and
// nothing that keeps a second pixel's taps in flight fitted.  Now a bin entry says by itself that it names an alpha-tested
// triangle (CHORD_REC_MASKED), the tile kernel skips such entries -- its only instantiations are the opaque ones --, and this
// kernel, launched between the binning and the tile schedule, scan-converts them: one workgroup per tile whose counter line
// carries the TC_MASKED flag, the tile in LDS from zero, the entries set up exactly as before (entry_store, masked_rows: the
// arithmetic is untouched), row units dealt out densely.  On the first pass of a frame the finished tile is WRITTEN (plain
// 16-byte stores) and the tile kernel, told by the tile schedule (bit 30 of the work item), starts from those words instead of
// from zero; on later passes the touched pixels are merged with device-scope atomicMax and the tile kernel reads the tile back
// as it does anyway.  The 64-bit max is order-independent, so which kernel merges a fragment first changes nothing.
// Price: a masked tile's 32 KB once out and once in on the first pass, and one more launch per pass of a scene with alpha-tested
// materials; scenes without them launch nothing of this.
#if !CHORD_MASKED_FUSED && TILE_BATCH != TB
#error "the separate masked pass sets up one entry per thread of its workgroup: TILE_BATCH must be TB"
#endif
#if TILE_BATCH == TB
template <bool SH, bool DEPTH>
__global__ __launch_bounds__(TB, 4) void raster_masked_tile_kernel(RasterParams p)
{
    __shared__ __align__(16) unsigned long long tile[TILE * TPITCH];   // 32.5 KB
    __shared__ EntrySoA prm;                                     // 26 KB
    __shared__ uint32_t unitList[UNIT_CAP];                      // 16 KB
    __shared__ uint32_t waveSums[2][TB / 64];
    __shared__ uint32_t chunkTab[64];
    // work items: the tile schedule's (heaviest bins first: the masked pass ends with light tiles, like the tile kernel); an item
    // without the masked flag -- or a further slice of a split tile -- is none of this pass's business
    // (one workgroup per tile of the target strides over the list: a frame with split tiles has more items than tiles)
    const uint32_t active = p.tileOrder[0].x;
    for (uint32_t oi = blockIdx.x; oi < active; oi += gridDim.x) {
    const uint2 item = p.tileOrder[1u + oi];
    if ((item.y & 0x40000000u) == 0u || ((item.x >> 12) & 0x3FFu) != 0u) continue;
    const uint32_t tileId = item.x & 0xFFFu;
    const uint32_t n = item.y & 0x3FFFFFFFu;                      // (clamped to the bin capacity by the schedule)
    const int32_t ox = (int32_t)(tileId % p.tilesX) * TILE, oy = (int32_t)(tileId / p.tilesX) * TILE;
    const int32_t tw = min(TILE, p.Wi - ox), th = min(TILE, p.Hi - oy);
    const bool noPixels = ABL(p, DBG_NO_PIXELS);
    for (uint32_t i = threadIdx.x; i < TILE * TPITCH / 2; i += TB) reinterpret_cast<ulonglong2*>(tile)[i] = make_ulonglong2(0ull, 0ull);
    const uint32_t* bin = p.tileBins + (size_t)tileId * p.binCap;
    const uint32_t wideLimit = min(p.triCap * CHORD_LIST_SHARDS, CHORD_REC_WIDE_INDEX);
    uint32_t batch = 0;
    // the bin in segments: its fixed part, then windows of 64 pool chunks (their names through chunkTab, as in the tile kernel)
    for (uint32_t segLo = 0; segLo < n;) {
        uint32_t segHi, chunk0 = 0;
        if (segLo < p.binCap) segHi = min(n, p.binCap);
        else {
            chunk0 = (segLo - p.binCap) >> CHORD_BIN_CHUNK_SHIFT;
            segHi = min(n, segLo + 64u * CHORD_BIN_CHUNK);
            __syncthreads();                                      // (the previous window's readers are done)
            for (uint32_t j = threadIdx.x; j < 64u; j += TB) {
                const unsigned long long e = chunk0 + j < p.binMaxChunks ? p.binChunkTab[(size_t)tileId * p.binMaxChunks + chunk0 + j] : 0ull;
                chunkTab[j] = (uint32_t)(e >> 32) == p.binStamp ? (uint32_t)e : CHORD_BIN_CHUNK_INVALID;
            }
        }
        __syncthreads();                                          // the zeroed tile / the chunk names are visible
        auto binWord = [&](uint32_t k) -> uint32_t {
            if (k < p.binCap) return bin[k];
            const uint32_t o = k - p.binCap, cj = (o >> CHORD_BIN_CHUNK_SHIFT) - chunk0;
            const uint32_t id = cj < 64u ? chunkTab[cj] : CHORD_BIN_CHUNK_INVALID;
            return id == CHORD_BIN_CHUNK_INVALID ? 0xFFFFFFFFu : p.binPool[(size_t)id * CHORD_BIN_CHUNK + (o & (CHORD_BIN_CHUNK - 1u))];
        };
        uint32_t wNext = segLo + threadIdx.x < segHi ? binWord(segLo + threadIdx.x) : 0xFFFFFFFFu;
        for (uint32_t base = segLo; base < segHi; base += TB, batch++) {
            const uint32_t w = wNext;
            wNext = base + TB + threadIdx.x < segHi ? binWord(base + TB + threadIdx.x) : 0xFFFFFFFFu;
            const uint32_t idx = w & CHORD_REC_WIDE_INDEX;
            uint32_t units = 0;
            if ((w & 0xE0000000u) == CHORD_REC_MASKED && idx < wideLimit) {
                const uint4* src = reinterpret_cast<const uint4*>(&p.tris[idx]);
                const uint4 q0 = src[0], q1 = src[1], q2 = src[2];
                TriRec r;
                r.X[0] = (int32_t)q0.x; r.X[1] = (int32_t)q0.y; r.X[2] = (int32_t)q0.z; r.Y[0] = (int32_t)q0.w;
                r.Y[1] = (int32_t)q1.x; r.Y[2] = (int32_t)q1.y;
                r.d[0] = __uint_as_float(q1.z); r.d[1] = __uint_as_float(q1.w); r.d[2] = __uint_as_float(q2.x);
                r.payload = q2.y; r.twoSided = q2.z; r.pad = q2.w;
                TriSetup ts;
                tri_setup_from_record(ts, r, p.Wi, p.Hi);
                const int32_t x0 = max(ts.px0, ox), y0 = max(ts.py0, oy);
                const int32_t x1 = min(ts.px1, ox + tw - 1), y1 = min(ts.py1, oy + th - 1);
                if (x1 >= x0 && y1 >= y0) units = entry_store(prm, threadIdx.x, ts, narrow_extent(ts), ox, oy, x0, y0, x1, y1, true, idx);
            }
            uint32_t total;
            const uint32_t off = block_scan_tb(units, waveSums[batch & 1u], &total);   // (its barrier also publishes the entries)
            for (uint32_t r0 = 0; r0 < total; r0 += UNIT_CAP) {
                if (units) {
                    const uint32_t box = prm.w[11][threadIdx.x];
                    const uint32_t nseg = ((((box >> 12) & 63u) - (box & 63u)) >> MASKED_SEG_SHIFT) + 1u, y0l = (box >> 6) & 63u;
                    const uint32_t lo2 = max(off, r0), hi2 = min(off + units, r0 + UNIT_CAP);
                    if (lo2 < hi2) {
                        const uint32_t j = lo2 - off, g = j / nseg;
                        uint32_t row = y0l + g * MASKED_ROWS, seg = j - g * nseg;
                        for (uint32_t u = lo2; u < hi2; u++) {
                            unitList[u - r0] = threadIdx.x | (row << 9) | (seg << 15);
                            if (++seg == nseg) { seg = 0u; row += MASKED_ROWS; }
                        }
                    }
                }
                __syncthreads();
                const uint32_t nr = min(total - r0, (uint32_t)UNIT_CAP);
                for (uint32_t ui = threadIdx.x; ui < nr; ui += TB) {
                    const uint32_t d = unitList[ui];
                    entry_unit_masked<DEPTH>(p, prm, tile, d & 511u, (d >> 9) & 63u, (d >> 15) & 7u, ox, oy, noPixels);
                }
                if (r0 + UNIT_CAP < total) __syncthreads();       // the list is rewritten by the next round
            }
            if (total) __syncthreads();                           // prm / the unit list are rewritten by the next batch
        }
        segLo = segHi;
    }
    __syncthreads();
    // ---- tile out: the whole tile on the first pass of a frame (the tile kernel starts from it), the touched pixels otherwise ----
    const uint32_t visBase = SH ? p.shard.tileSlot[tileId] << (2 * TILE_SHIFT) : (uint32_t)oy * (uint32_t)p.Wi + (uint32_t)ox;
    const uint32_t visPitch = SH ? (uint32_t)TILE : (uint32_t)p.Wi;
    if (p.clearTiles) {
        for (uint32_t i = threadIdx.x; i < TILE * TILE / 2; i += TB) {
            const int32_t ly = (int32_t)(i >> (TILE_SHIFT - 1)), lx = (int32_t)(i & (TILE / 2 - 1)) * 2;
            if (ly >= th || lx >= tw) continue;
            const ulonglong2 v = make_ulonglong2(tile[ly * TPITCH + lx], tile[ly * TPITCH + lx + 1]);
            unsigned long long* dst = p.vis + (size_t)(visBase + (uint32_t)ly * visPitch + (uint32_t)lx);
            if (lx + 1 < tw) *reinterpret_cast<ulonglong2*>(dst) = v;
            else dst[0] = v.x;
        }
    } else {
        for (uint32_t i = threadIdx.x; i < TILE * TILE; i += TB) {
            const int32_t ly = (int32_t)(i >> TILE_SHIFT), lx = (int32_t)(i & (TILE - 1));
            const unsigned long long v = tile[ly * TPITCH + lx];
            if (v != 0ull && ly < th && lx < tw) atomicMax(p.vis + (size_t)(visBase + (uint32_t)ly * visPitch + (uint32_t)lx), v);
        }
    }
    __syncthreads();                                              // the LDS tile is reused by the next item
    }
}
#endif

// ---- launcher ---------------------------------------------------------------------------------
hipError_t launch_raster(ChordCtx* c, const CmdList& in, bool clearTiles)
{
#define LR_HIP(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)
    RasterParams p;
    p.hzbFused = 0; p.hzbMinA = nullptr; p.hzbMinB = nullptr; p.hzbMaxB = nullptr; p.tileRange = c->dTileRange;
    p.hzbExA = nullptr; p.hzbExB = nullptr;
    p.hzbDesc = c->hzb[0].desc;
    if (c->fuseHzb && c->shard.ranks == 1) {
        p.hzbFused = 1;
        p.hzbMinA = (clearTiles && c->fuseHzbTemp) ? c->hzb[0].minTexels : nullptr;
        p.hzbMinB = c->hzb[c->fuseHzbSlot].minTexels; p.hzbMaxB = c->hzb[c->fuseHzbSlot].maxTexels;
    } else if (c->fuseHzb) {
        // sharded frame: the same reduction, into the owned tiles' slots of the exchange buffers (launch_hzb_untile after the all-gathers)
        p.hzbFused = 1;
        p.hzbExA = (clearTiles && c->fuseHzbTemp) ? c->dHzbExchange : nullptr;
        p.hzbExB = c->dHzbFinalExchange;
    }
    p.count = in.count; p.cmds = in.cmds; p.cmdCap = std::max(in.capacity, 1u);
    if (c->shard.ranks > 1) {
        // sharded frame: only the clusters that touch this rank's screen tiles reach the setup kernel.  The lists of a frame
        // are the rank's own already (the group cull writes the rank's share of list 0 beside the full list; the HZB culls
        // of a sharded frame start from it); a list of unknown origin goes through the rank filter first.
        bool mine = false;
        if (in.cmds == c->lists[0].cmds && c->mineValid) { p.count = c->dCounts + 4; p.cmds = c->dMineCmds; mine = true; }
        else if (in.cmds == c->dMineCmds && c->mineValid) mine = true;
        else for (int k = 1; k < 3; k++) if (in.cmds == c->lists[k].cmds && c->listMine[k]) mine = true;
        if (!mine) {
            if (!c->dRankCmds) LR_HIP(hipMalloc((void**)&c->dRankCmds, sizeof(ChordDrawCmd) * (size_t)c->cmdCapacity));
            CmdList filtered;
            filtered.count = c->dCounts + 5; filtered.cmds = c->dRankCmds; filtered.capacity = in.capacity;
            LR_HIP(hipMemsetAsync(filtered.count, 0, sizeof(uint32_t), c->stream));
            launch_rank_filter(c, in, filtered);
            p.count = filtered.count; p.cmds = filtered.cmds;
        }
    }
    p.objFrame = c->dObjFrame; p.objStatic = c->dObjStatic;
    p.meshlets = c->dMeshlets; p.meshletData = c->dMeshletData; p.positions = c->dPositions;
    p.materials = c->dMaterials; p.texAlpha = c->dTexAlpha; p.texcoords = c->dTexcoords;
    p.vis = (unsigned long long*)c->dVis;
    p.depthOut = (p.hzbFused && clearTiles) ? c->depthOutTarget : nullptr;     // (only the fused first pass writes a whole image)
    p.W = (float)c->width; p.H = (float)c->height; p.Wi = (int32_t)c->width; p.Hi = (int32_t)c->height;
    p.shard = c->shard;
    p.blockPool = c->dBlockPool; p.blockCap = (c->debugFlags & DBG_NO_BLOCKS) ? 0u : c->blockCap;
    p.blockForce = (c->debugFlags & DBG_FORCE_BLOCKS) ? 1u : 0u;
    p.tris = c->dTris; p.triCap = c->triCap / CHORD_LIST_SHARDS;
    p.trisC = c->dTrisC; p.triCapC = c->triCapC / CHORD_LIST_SHARDS;
    const uint32_t tiles = c->tilesX * c->tilesY;
    const uint32_t pass = c->rasterCalls & 1u;
    p.tileCount = c->dFrameState->tileCount + (size_t)pass * tiles * TC_STRIDE;
    p.tileBins = c->dTileBins + (size_t)pass * tiles * c->binCap; p.binCap = c->binCap;
    p.binPool = c->dBinPool + (size_t)pass * c->binPoolChunks * CHORD_BIN_CHUNK; p.binPoolChunks = c->binPoolChunks;
    p.binPoolCount = &c->dCounters->binPoolCount[pass];
    p.binChunkTab = c->dBinChunkTab + (size_t)pass * tiles * c->binMaxChunks; p.binMaxChunks = c->binMaxChunks;
    p.binStamp = ++c->rasterSerial;
    p.tilesX = c->tilesX; p.tilesY = c->tilesY;
    p.clipTris = c->dClipTris + (size_t)pass * (c->clipTriCap / 2); p.clipTriCap = c->clipTriCap / 2; p.pass = pass;
    p.largeList = c->dLargeList + (size_t)pass * (c->largeCap / 2); p.largeCap = c->largeCap / 2 / CHORD_LIST_SHARDS;   // per shard
    p.counters = c->dCounters;
    p.clearTiles = clearTiles ? 1u : 0u;
    p.depthOnly = c->depthOnly ? 1u : 0u; p.depthClamp = c->depthClamp ? 1u : 0u;
    p.biasConst = c->depthBiasConst; p.biasSlope = c->depthBiasSlope;
    p.debug = c->debugFlags;
    p.tileClocks = c->dTileClocks + (size_t)pass * CHORD_MAX_TILES;
    p.tileOrder = reinterpret_cast<uint2*>(c->dTileOrder); p.tileSlabs = c->dTileSlabs;
    p.tilePhase = c->dTileClocks + (size_t)2 * CHORD_MAX_TILES + (size_t)pass * CHORD_MAX_TILES * 8;

    // A frame zeroes every count once (begin_frame_clear); outside a frame, or from the third raster
    // call of a frame on, the pass slot is recycled here.
    if (!c->inFrame || c->rasterCalls >= 2) {
        LR_HIP(hipMemsetAsync(p.tileCount, 0, sizeof(uint32_t) * TC_STRIDE * tiles, c->stream));
        LR_HIP(hipMemsetAsync(&c->dCounters->clipTriCount[pass], 0, sizeof(uint32_t), c->stream));
        LR_HIP(hipMemsetAsync(c->dCounters->largeCount[pass], 0, sizeof(c->dCounters->largeCount[pass]), c->stream));
        LR_HIP(hipMemsetAsync(&c->dCounters->binPoolCount[pass], 0, sizeof(uint32_t), c->stream));
        if (!c->inFrame) { LR_HIP(hipMemsetAsync(c->dCounters->triCount, 0, sizeof(c->dCounters->triCount), c->stream));
                           LR_HIP(hipMemsetAsync(c->dCounters->triCountC, 0, sizeof(c->dCounters->triCountC), c->stream));
                           LR_HIP(hipMemsetAsync(c->dCounters->blockGranules, 0, sizeof(c->dCounters->blockGranules), c->stream)); }
    }

    // Pixel blocks: a list can only be dense (launch_is_dense) when its capacity allows one cluster per 16 pixels of the rank's
    // screen; then the block kernel runs ahead of the record kernel, which sets up what the block kernel left over.
    const uint64_t rankPixels = (uint64_t)c->width * c->height / (c->shard.ranks > 1 ? c->shard.ranks : 1u);
    const bool maybeDense = p.blockCap != 0u && (p.blockForce != 0u || (uint64_t)in.capacity * 16ull >= rankPixels);
    p.leftCount = nullptr; p.leftCmds = nullptr;
    p.binHint = nullptr; p.countHint = nullptr;
    p.slotHot = (c->debugFlags & DBG_FORCE_HOT) ? 64u : SLOT_HOT;
    {   // Sharded frames only: a rank of an 8-rank 4K frame owns 255 tiles (config 4: tile kernel 0.088 -> 0.053 ms per rank); the
        // one single-GPU case with fewer tiles than slots, a 1080p target, measured the same with and without
        // (profiles/r05_tile_kernel_experiments.txt, item 8).  (CHORDVIS_TILE_SLOTS: measurements only -- 0 keeps every bin up to
        // TILE_SPLIT_MIN entries whole, a number applies to every frame)
        static const int forced = [] { const char* e = getenv("CHORDVIS_TILE_SLOTS"); return e ? atoi(e) : -1; }();
        p.tileSlots = forced >= 0 ? (uint32_t)forced : c->shard.ranks > 1 ? (uint32_t)c->numCUs * (CHORD_TILE_SHIFT == 6 ? 2u : 6u) : 0u;
        // (CHORDVIS_TILE_SPLIT_MIN / CHORDVIS_TILE_SLICE: measurements only -- the image does not depend on the cut)
        static const int splitMin = [] { const char* e = getenv("CHORDVIS_TILE_SPLIT_MIN"); return e ? atoi(e) : -1; }();
        static const int sliceLen = [] { const char* e = getenv("CHORDVIS_TILE_SLICE"); return e ? atoi(e) : -1; }();
        p.tileSplitMin = splitMin > 0 ? (uint32_t)splitMin : TILE_SPLIT_MIN;
        p.tileSliceLen = sliceLen >= 512 ? ((uint32_t)sliceLen + 511u) & ~511u : TILE_SLICE;
    }
    p.hotTiles = c->dHotTiles ? c->dHotTiles + (size_t)pass * (1u + CHORD_HOT_TILES) : nullptr;
    // ... and whether it IS dense the device decides from the list's length (launch_is_dense).  A list that could be dense but was
    // nowhere near it in the last frame the GPU finished (BASELINE config 4: one cluster per 60 pixels, capacity for one per 30)
    // gets no block kernel at all -- the launch would find nothing to do and cost its 4.5 us, twice per frame.  Half the device's
    // threshold, so a list at the threshold keeps its block kernel whichever way the last frame fell; the record kernel then
    // sets up the input list itself (no leftover list: launch_is_dense is false without one), and either way renders the same image.
    bool blocksWorthLaunching = maybeDense;
    if (maybeDense && c->dBinHint) {
        p.countHint = c->dBinHint + 2 + pass;
        const uint32_t last = c->hBinHint[2 + pass];              // (0xFFFFFFFF: no frame yet)
        if (p.blockForce == 0u && last != 0xFFFFFFFFu && (uint64_t)last * 32ull < rankPixels) blocksWorthLaunching = false;
    }
    if (blocksWorthLaunching) {
        if (c->dBinHint) p.binHint = c->dBinHint + pass;          // (host-visible word per pass, allocated with the G-buffer)
        if (!c->dLeftCmds) LR_HIP(hipMalloc((void**)&c->dLeftCmds, sizeof(ChordDrawCmd) * (size_t)c->cmdCapacity));
        p.leftCount = c->dCounts + 6 + pass; p.leftCmds = c->dLeftCmds;
        if (!c->inFrame || c->rasterCalls >= 2) LR_HIP(hipMemsetAsync(p.leftCount, 0, sizeof(uint32_t), c->stream));
    }
    uint32_t blocks = (in.capacity + 3u) / 4u;
#ifndef SETUP_GRID_MULT
#define SETUP_GRID_MULT 1u
#endif
    const uint32_t maxBlocks = (uint32_t)c->numCUs * SETUP_MIN_WAVES * SETUP_GRID_MULT;    // what is resident at SETUP_MIN_WAVES waves per SIMD
    if (blocks > maxBlocks) blocks = maxBlocks;
    if (blocks < 1) blocks = 1;
    const bool sh = c->shard.ranks > 1;
    stamp(c, S_HZBCULL);      // closes whatever preceded the raster (HZB cull / list reset)
    if (blocksWorthLaunching) {
        const uint32_t bb = std::max(1u, std::min((in.capacity + 3u) / 4u, (uint32_t)c->numCUs * BLOCKS_MIN_WAVES));
        // the longest bin of this pass in the last frame the GPU finished: a hot tile then -> the variant that draws ahead now
        const bool hot = (c->hBinHint && c->hBinHint[pass] >= SLOT_HOT) || (c->debugFlags & DBG_FORCE_HOT);
        if (hot) CHORD_LAUNCH(c, raster_setup_blocks_kernel<true>, dim3(bb), dim3(256), 0, c->stream, p);
        else     CHORD_LAUNCH(c, raster_setup_blocks_kernel<false>, dim3(bb), dim3(256), 0, c->stream, p);
    }
    if (c->anyMasked) CHORD_LAUNCH(c, raster_setup_kernel<true>, dim3(blocks), dim3(256), 0, c->stream, p);
    else              CHORD_LAUNCH(c, raster_setup_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, p);
    stamp(c, S_R_CLUSTER);
    CHORD_LAUNCH(c, raster_clip_and_bin_large_kernel, dim3(CLIP_BLOCKS + (uint32_t)c->numCUs * 4u), dim3(256), 0, c->stream, p);
    // The first pass of a main-view frame on one GPU writes EVERY tile (it is the clear): its work items are all the tiles whatever
    // their bins hold, and only their order (heaviest first) and the cut of long bins depend on this frame.  Both change slowly from
    // frame to frame, so the schedule of such a pass is kept in a buffer of its own and made again only every orderKeepFrames + 1
    // frames (chordvis_set_tile_schedule_keep, default 7; -DTILE_ORDER_KEEP=0 compiles the path out); in between the tile kernel takes items and order from the kept schedule and a tile's bin length and flags from the
    // counter line (RasterParams::orderKept) -- one launch less in most frames.  The image does not depend on order or cut.  (The
    // hints the schedule kernel leaves for the next frame -- longest bin, cluster count, hot tiles -- age with it.)
    p.orderKept = 0u;
    p.heavyHint = nullptr;
    bool makeOrder = true;
    // LIGHT later passes of a frame (the read-modify-write passes behind the first: config 3's second pass is 205 clusters, 436 of its
    // 2 040 tiles touched with 60 entries each) run WITHOUT a schedule: one workgroup per tile of the target, tile = workgroup index, a
    // tile whose counter line says "no entries" ends after that one load.  What such a pass has no use for -- heaviest-first order,
    // bins cut into slices -- is what the schedule kernel was there for (config 3 0.1792 -> 0.1759 ms, a 1080p frame 0.112 -> 0.109).
    // A pass with work in every tile needs them (config 4's second pass: 57 -> 115 us of tile kernel without), so the pass says what
    // it is in two host-visible words -- its serial under "heavy" when a bin is longer than tileSplitMin or it set up more than
    // TILE_DIRECT_MAX_CLUSTERS clusters, under "light" otherwise; written by the schedule kernel, or by the tile kernel of a direct
    // pass -- and runs direct while the latest report the host has seen says light.  A camera cut that sends the scene
    // through the second pass costs that frame balance, never a pixel (-DTILE_DIRECT=0 compiles the path out; CHORDVIS_TILE_DIRECT=0
    // turns it off at run time: A/B runs).
    static const bool directOn = [] { const char* e = getenv("CHORDVIS_TILE_DIRECT"); return !e || atoi(e) != 0; }();
    // (laterOk: a read-modify-write pass of a frame with the HZB fused into its tile-out -- what the direct form and the kept schedule of
    // a later pass are written for)
    const bool laterOk = c->inFrame && !clearTiles && p.hzbFused && !c->depthOnly && CHORD_MASKED_FUSED && c->dBinHint && !(c->debugFlags & ~(DBG_NO_BLOCKS | DBG_FORCE_BLOCKS | DBG_FORCE_HOT | 524288u));
    if (laterOk) p.heavyHint = c->dBinHint + 4 + pass;
    if (TILE_DIRECT && directOn && laterOk) {
        const uint32_t heavySeen = c->hBinHint[4 + pass], lightSeen = c->hBinHint[6 + pass];
        // (the host runs frames ahead of the device -- a bench loop enqueues hundreds: what it reads is the state of a pass long past,
        // so the rule is "the latest report says light", not "a report of the last few frames"; a pass that reports both is heavy)
        if (lightSeen != 0u && (heavySeen == 0u || (int32_t)(lightSeen - heavySeen) > 0)) {
            p.orderKept = 2u; makeOrder = false;
        }
    }
    // Kept schedules (chordvis_set_tile_schedule_keep != 0; sharded frames too: a rank's work items are its own tiles, and the map they
    // follow changes only through install_tile_owners, which invalidates the schedules).  Slot 0: the first pass of a frame, which writes
    // every tile -- its work items never change, only their order and the cut of long bins.  Slot 1: the second pass; its schedule lists
    // EVERY tile of the rank, the untouched ones last (orderAll) -- a tile that a later frame touches must be a work item --, and an
    // untouched tile's workgroup ends after the load of its counter line, as in a direct pass.  Under a kept schedule the tile kernel
    // takes a tile's bin length and flags from the counter line (orderKept).
    // WHO MAKES THE SCHEDULE (round 6, end): the tile kernel of the frame before.  Its workgroup 0 orders THIS launch's bin counts --
    // final before the kernel starts -- into the slot's other buffer (tileOrderNext) while the other workgroups raster; the next frame
    // reads that buffer.  Every frame then runs under a schedule exactly one frame old and no frame launches a schedule kernel: measured
    // along a moving camera a one-frame-old schedule is as good as a fresh one, while one kept for 3 / 7 frames costs a config-4 or
    // masked frame 3 / 7 % of balance (profiles/r06_experiments.txt item 15) -- the launch it saved was 2 %.  CHORDVIS_TILE_NEXT=0: the
    // form before -- a schedule kernel every (keep + 1)-th frame, the schedule kept in between (A/B runs).
    p.orderAll = 0u; p.tileOrderNext = nullptr;
    static const bool keepLaterOn = [] { const char* e = getenv("CHORDVIS_TILE_KEEP_LATER"); return !e || atoi(e) != 0; }();
    static const bool nextOn = [] { const char* e = getenv("CHORDVIS_TILE_NEXT"); return !e || atoi(e) != 0; }();
    const bool keepOk = TILE_ORDER_KEEP && c->orderKeepFrames && CHORD_MASKED_FUSED && c->inFrame && !c->depthOnly && !(c->debugFlags & ~524288u);
    int slot = -1;
    if (keepOk && clearTiles && pass == 0u && c->dTileOrderKeep) slot = 0;
    else if (keepOk && keepLaterOn && laterOk && pass == 1u && c->dTileOrderKeep1) slot = 1;
    if (slot >= 0) {
        uint32_t& age = slot ? c->orderAge1 : c->orderAge;
        uint2* buf = reinterpret_cast<uint2*>(slot ? c->dTileOrderKeep1 : c->dTileOrderKeep);     // two schedules of 1 + tileItemCap items
        const size_t half = (size_t)1 + c->tileItemCap;
        p.orderAll = slot ? 1u : 0u;
        if (nextOn && p.orderKept == 2u) {
            // a direct pass reads no schedule and makes none: its tile kernel is as long as its slowest tile's chain of fetches (config 3:
            // 23 us), and the schedule part -- one workgroup's three dependent passes over the counter lines -- would be the longest
            // chain of the launch (measured: 23.2 -> 26.4 us).  A pass that turns heavy makes its first schedule with the schedule kernel.
            age = 0xFFFFFFFFu; p.orderAll = 0u;
        } else if (nextOn) {
            uint32_t& flip = c->orderFlip[slot];
            p.tileOrder = buf + flip * half; p.tileOrderNext = buf + (flip ^ 1u) * half;
            if (age != 0xFFFFFFFFu) { p.orderKept = 1u; makeOrder = false; }      // the schedule the last frame's tile kernel made
            age = 0u; flip ^= 1u;
        } else if (p.orderKept != 2u) {
            p.tileOrder = buf;
            if (age < c->orderKeepFrames) { age++; p.orderKept = 1u; makeOrder = false; }
            else age = 0u;
        } else p.orderAll = 0u;
    }
    if (makeOrder) CHORD_LAUNCH(c, raster_tile_order_kernel, dim3(1), dim3(1024), 0, c->stream, p);
#if !CHORD_MASKED_FUSED
    if (c->anyMasked) {
        // alpha-tested triangles: their own pass over the scheduled tiles whose bins hold any (raster_masked_tile_kernel)
        const uint32_t items = tiles;                                     // (the workgroups stride over the device-side item list)
        if (c->depthClamp && !sh) CHORD_LAUNCH(c, (raster_masked_tile_kernel<false, true>), dim3(items), dim3(TB), 0, c->stream, p);
        else if (sh)              CHORD_LAUNCH(c, (raster_masked_tile_kernel<true, false>), dim3(items), dim3(TB), 0, c->stream, p);
        else                      CHORD_LAUNCH(c, (raster_masked_tile_kernel<false, false>), dim3(items), dim3(TB), 0, c->stream, p);
    }
#endif
    stamp(c, S_R_CLIP);
    // first pass of a frame: every tile is written, one block each, dispatched heaviest first; later passes touch
    // few tiles: one resident wave of blocks strides over the (device-side) active list
    // (a static snake assignment of 2 or 4 items per block with the next item prefetched -- half / a quarter of the blocks,
    // start-up round trips paid once, also with a loop-end barrier that does not wait for the visibility stores to drain --
    // was measured: tile kernel +4..6 % on config 3, +11..18 % on config 4; the dispatcher's dynamic hand-out of one item
    // per block balances better than any static split)
    // (sharded frames: the work items are the rank's own tiles)
    // (a rank's slices: its bins are cut into about tileSlots shares when it owns fewer tiles than that; blocks beyond the item count leave at once)
    const uint32_t tileBlocks = (p.tileOrderNext ? 1u : 0u) + ((clearTiles || p.orderKept == 2u || p.orderAll) ? ((sh && clearTiles) ? min(tiles, max(c->shard.slotsPerRank, p.tileSlots + p.tileSlots / 2u)) : tiles) : min(tiles, (uint32_t)c->numCUs * (CHORD_TILE_SHIFT == 6 ? 2u : 6u)));
    // (the tile kernel's instantiations are the opaque ones: alpha-tested triangles were scan-converted by the masked pass above)
#if CHORD_MASKED_FUSED
    if (c->anyMasked) {
        if (c->depthClamp && !sh) CHORD_LAUNCH(c, (raster_tile_kernel<false, true, true>), dim3(tileBlocks), dim3(TB), 0, c->stream, p);
        else if (sh)              CHORD_LAUNCH(c, (raster_tile_kernel<true, true, false>), dim3(tileBlocks), dim3(TB), 0, c->stream, p);
        else                      CHORD_LAUNCH(c, (raster_tile_kernel<false, true, false>), dim3(tileBlocks), dim3(TB), 0, c->stream, p);
    } else
#endif
    if (c->depthClamp && !sh) {
        CHORD_LAUNCH(c, (raster_tile_kernel<false, false, true>), dim3(tileBlocks), dim3(TB), 0, c->stream, p);
    } else {
        if (sh) CHORD_LAUNCH(c, (raster_tile_kernel<true, false, false>), dim3(tileBlocks), dim3(TB), 0, c->stream, p);
        else    CHORD_LAUNCH(c, (raster_tile_kernel<false, false, false>), dim3(tileBlocks), dim3(TB), 0, c->stream, p);
    }
    stamp(c, S_R_CHUNK);
    c->rasterCalls++;
    return hipSuccess;
#undef LR_HIP
}

} // namespace chord
