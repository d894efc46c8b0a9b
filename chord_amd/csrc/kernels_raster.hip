// Software rasterizer of the visibility path, hand-written for gfx950 (wave64, LDS-staged).
//
// Replaces meshRasterPassMS + the fixed-function rasterizer + depth test + meshRasterPassPS
// (mesh_raster.hlsl:51-210; state mesh_raster.cpp:141-156, helper.h:7-13,304-324,395-407) with
//   raster_cluster_kernel  one wave per visible meshlet: coalesced meshletData / position stream ->
//                          clip-space transform -> LDS (SoA x,y,w,u,v,depth) -> per-triangle culls
//                          (mesh_raster.hlsl:143-179) -> setup -> two size classes:
//                            small (bbox <= 16x16 px)  per-lane scan, 32-bit edge functions
//                            big                       appended to a device list as 64x64 chunks
//                          triangles touching the near/guard planes -> clip list
//   raster_clip_kernel     homogeneous Sutherland-Hodgman clipper (rare), one lane per triangle
//   raster_chunk_kernel    one wave per 64x64 chunk of a big triangle: 64 lanes classify the 64
//                          8x8 tiles, then scan the surviving tiles cooperatively
// Every covered pixel does atomicMax(u64) of (asuint(depth) << 32 | ((slot+1)&0xFFFFFF)<<8 | tri):
// reverse-Z "greater wins" + id in one global_atomic_umax_x2 (device scope, resolved at the
// memory side so it is coherent across the 8 XCD L2s).
//
// Arithmetic is the canonical restatement of SURVEY.md §8c: snapped 24.8 coordinates, pixel
// centres at +0.5, top-left rule, integer edge functions, depth = (d0 + l1*(d1-d0)) + l2*(d2-d0).
// Integer / fp32 VALU + atomics; no MFMA.  Built with -ffp-contract=off.

#include "device_layer.h"
#include "device_math.h"

namespace chord {

#define GUARD_BAND 1024.0f
#define LDS_VERTS 256

struct RasterParams {
    const uint32_t* count; const ChordDrawCmd* cmds;
    const DObjFrame* objFrame; const DObjStatic* objStatic;
    const DMeshlet* meshlets; const uint32_t* meshletData; const float* positions;
    unsigned long long* vis;
    float W, H; int32_t Wi, Hi;
    ShardInfo shard;
    BigTri* bigTris; BigChunk* bigChunks; ClipTri* clipTris;
    uint32_t bigTriCap, bigChunkCap, clipTriCap;       // big caps are PER SHARD
    DeviceCounters* counters;
    uint32_t debug;                                    // ablation switches (chordvis_set_debug), 0 in production
};
#define DBG_NO_PIXELS   1u    // skip every visibility write
#define DBG_PLAIN_STORE 2u    // plain store instead of atomicMax (wrong image; timing only)
#define DBG_NO_BIG      4u    // drop big triangles instead of deferring them
#define DBG_NO_EARLYZ   8u    // tile path: no read-before-atomic

__device__ __forceinline__ int32_t bcast(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ uint32_t bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ float bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += n;
    }
    return v;
}

template <bool SH>
__device__ __forceinline__ bool owns_row(const ShardInfo& s, int32_t y)
{
    if (!SH) return true;
    return (((uint32_t)y / s.stripeRows) % s.ranks) == s.rank;
}

template <bool SH>
__device__ __forceinline__ size_t row_base(const ShardInfo& s, int32_t y, int32_t Wi)
{
    if (!SH) return (size_t)y * (size_t)Wi;
    const uint32_t stripe = (uint32_t)y / s.stripeRows;
    const uint32_t local = stripe / s.ranks;
    const uint32_t owner = stripe % s.ranks;
    return ((size_t)(owner * s.stripesPerRank + local) * s.stripeRows + ((uint32_t)y % s.stripeRows)) * (size_t)Wi;
}

__device__ __forceinline__ bool in_fast_volume(const f4& h)
{
    return h.w > 0.0f && (h.w - h.z) >= 0.0f && h.z >= 0.0f &&
           (GUARD_BAND * h.w + h.x) >= 0.0f && (GUARD_BAND * h.w - h.x) >= 0.0f &&
           (GUARD_BAND * h.w + h.y) >= 0.0f && (GUARD_BAND * h.w - h.y) >= 0.0f;
}

__device__ __forceinline__ int32_t floor_shift8(int32_t v) { return v >> 8; }   // arithmetic shift == floor

// Per-triangle setup shared by every path.  Edge i is opposite vertex i:
// E0 = orient(V1,V2,P), E1 = orient(V2,V0,P), E2 = orient(V0,V1,P), times s so the interior is positive.
struct TriSetup {
    int32_t X[3], Y[3];
    float d0, e1, e2, invA;
    int32_t px0, py0, px1, py1;       // clamped pixel bbox
    int64_t area;                     // |2A|
    int32_t s;                        // orientation sign
    uint32_t payload;
};

// Returns false when the triangle is rejected (zero area, back face after snapping, empty bbox).
__device__ __forceinline__ bool tri_setup(TriSetup& ts, bool twoSided, int32_t Wi, int32_t Hi)
{
    const int64_t area2 = (int64_t)(ts.X[1] - ts.X[0]) * (int64_t)(ts.Y[2] - ts.Y[0]) -
                          (int64_t)(ts.X[2] - ts.X[0]) * (int64_t)(ts.Y[1] - ts.Y[0]);
    if (area2 == 0) return false;
    if (!twoSided && area2 > 0) return false;            // VK_CULL_MODE_BACK_BIT (mesh_raster.cpp:235)
    ts.s = area2 < 0 ? -1 : 1;
    ts.area = area2 < 0 ? -area2 : area2;
    const int32_t minX = min(ts.X[0], min(ts.X[1], ts.X[2])), maxX = max(ts.X[0], max(ts.X[1], ts.X[2]));
    const int32_t minY = min(ts.Y[0], min(ts.Y[1], ts.Y[2])), maxY = max(ts.Y[0], max(ts.Y[1], ts.Y[2]));
    ts.px0 = max(0, floor_shift8(minX + 127));
    ts.py0 = max(0, floor_shift8(minY + 127));
    ts.px1 = min(Wi - 1, floor_shift8(maxX - 128));
    ts.py1 = min(Hi - 1, floor_shift8(maxY - 128));
    if (ts.px1 < ts.px0 || ts.py1 < ts.py0) return false;
    ts.invA = 1.0f / (float)(double)ts.area;
    return true;
}

__device__ __forceinline__ void vis_write(unsigned long long* p, float z, uint32_t payload, uint32_t debug = 0)
{
    const unsigned long long packed = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)payload;
    if (debug) {
        if (debug & DBG_NO_PIXELS) return;
        if (debug & DBG_PLAIN_STORE) { *p = packed; return; }
    }
    atomicMax(p, packed);
}

// ---- small triangles: one lane scans its own bbox with 32-bit edge functions ------------------
template <bool SH>
__device__ __forceinline__ void raster_small(const RasterParams& p, const TriSetup& ts, float d1m0, float d2m0)
{
    // F_i(P) = s * (dx_i * (P.y - Ya) - dy_i * (P.x - Xa)),  a = dF/dx = -s*dy, b = dF/dy = s*dx
    const int32_t s = ts.s;
    const int32_t dx0 = ts.X[2] - ts.X[1], dy0 = ts.Y[2] - ts.Y[1];
    const int32_t dx1 = ts.X[0] - ts.X[2], dy1 = ts.Y[0] - ts.Y[2];
    const int32_t dx2 = ts.X[1] - ts.X[0], dy2 = ts.Y[1] - ts.Y[0];
    const int32_t a0 = -s * dy0, b0 = s * dx0;
    const int32_t a1 = -s * dy1, b1 = s * dx1;
    const int32_t a2 = -s * dy2, b2 = s * dx2;
    const int32_t bias0 = (a0 > 0 || (a0 == 0 && b0 > 0)) ? 0 : -1;
    const int32_t bias1 = (a1 > 0 || (a1 == 0 && b1 > 0)) ? 0 : -1;
    const int32_t bias2 = (a2 > 0 || (a2 == 0 && b2 > 0)) ? 0 : -1;
    const int32_t cx0 = ts.px0 * 256 + 128, cy0 = ts.py0 * 256 + 128;
    int32_t r0 = s * (dx0 * (cy0 - ts.Y[1]) - dy0 * (cx0 - ts.X[1]));
    int32_t r1 = s * (dx1 * (cy0 - ts.Y[2]) - dy1 * (cx0 - ts.X[2]));
    int32_t r2 = s * (dx2 * (cy0 - ts.Y[0]) - dy2 * (cx0 - ts.X[0]));
    for (int32_t py = ts.py0; py <= ts.py1; py++) {
        if (owns_row<SH>(p.shard, py)) {
            unsigned long long* row = p.vis + row_base<SH>(p.shard, py, p.Wi);
            int32_t E0 = r0, E1 = r1, E2 = r2;
            bool entered = false;
            for (int32_t px = ts.px0; px <= ts.px1; px++) {
                if (((E0 + bias0) | (E1 + bias1) | (E2 + bias2)) >= 0) {
                    const float l1 = (float)E1 * ts.invA, l2 = (float)E2 * ts.invA;
                    const float z = (ts.d0 + l1 * d1m0) + l2 * d2m0;
                    vis_write(row + px, z, ts.payload, p.debug);
                    entered = true;
                } else if (entered) {
                    break;                       // convex: the span of this row is over
                }
                E0 += a0 * 256; E1 += a1 * 256; E2 += a2 * 256;
            }
        }
        r0 += b0 * 256; r1 += b1 * 256; r2 += b2 * 256;
    }
}

// ---- cooperative 8x8 tile scan (64-bit edge functions) ---------------------------------------
struct WideEdges {
    int64_t a[3], b[3];     // dF/dx, dF/dy per subpixel unit
    int64_t bias[3];
    int64_t dx[3], dy[3];
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

// All 64 lanes: lane (lx, ly) handles pixel (tileX + lx, tileY + ly).
template <bool SH>
__device__ __forceinline__ void raster_tile(const RasterParams& p, const TriSetup& ts, const WideEdges& w,
                                            float d1m0, float d2m0, int32_t tileX, int32_t tileY, uint32_t lane)
{
    const int32_t px = tileX + (int32_t)(lane & 7u), py = tileY + (int32_t)(lane >> 3);
    if (px < ts.px0 || px > ts.px1 || py < ts.py0 || py > ts.py1) return;
    if (!owns_row<SH>(p.shard, py)) return;
    const int64_t E0 = edge_at(w, 0, px, py), E1 = edge_at(w, 1, px, py), E2 = edge_at(w, 2, px, py);
    if (((E0 + w.bias[0]) | (E1 + w.bias[1]) | (E2 + w.bias[2])) < 0) return;
    const float l1 = (float)(double)E1 * ts.invA, l2 = (float)(double)E2 * ts.invA;
    const float z = (ts.d0 + l1 * d1m0) + l2 * d2m0;
    unsigned long long* dst = p.vis + row_base<SH>(p.shard, py, p.Wi) + px;
    const unsigned long long packed = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)ts.payload;
    if (p.debug) {
        if (p.debug & DBG_NO_PIXELS) return;
        if (p.debug & DBG_PLAIN_STORE) { *dst = packed; return; }
        if (p.debug & DBG_NO_EARLYZ) { atomicMax(dst, packed); return; }
    }
    if (packed > *dst) atomicMax(dst, packed);        // stale read only costs a redundant atomic
}

// conservative: does the 8x8 tile at (tileX, tileY) touch the triangle (and its clamped bbox)?
__device__ __forceinline__ bool tile_overlaps(const TriSetup& ts, const WideEdges& w, int32_t tileX, int32_t tileY)
{
    if (tileX > ts.px1 || tileX + 7 < ts.px0 || tileY > ts.py1 || tileY + 7 < ts.py0) return false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int32_t cxp = w.a[i] > 0 ? tileX + 7 : tileX;
        const int32_t cyp = w.b[i] > 0 ? tileY + 7 : tileY;
        if (edge_at(w, i, cxp, cyp) + w.bias[i] < 0) return false;
    }
    return true;
}

// ---- deferred lists ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t chunk_count(const TriSetup& ts)
{
    return (uint32_t)(((ts.px1 >> 6) - (ts.px0 >> 6) + 1) * ((ts.py1 >> 6) - (ts.py0 >> 6) + 1));
}

__device__ __forceinline__ void write_big(const RasterParams& p, uint32_t shard, uint32_t ti, uint32_t ci,
                                          const TriSetup& ts, const float d[3], bool twoSided)
{
    const uint32_t n = chunk_count(ts);
    if (ti >= p.bigTriCap || ci + n > p.bigChunkCap) { atomicOr(&p.counters->overflow, 1u); return; }
    const uint32_t gti = shard * p.bigTriCap + ti;
    BigTri bt;
#pragma unroll
    for (int i = 0; i < 3; i++) { bt.X[i] = ts.X[i]; bt.Y[i] = ts.Y[i]; bt.d[i] = d[i]; }
    bt.payload = ts.payload; bt.twoSided = twoSided ? 1u : 0u; bt.pad = 0;
    p.bigTris[gti] = bt;
    BigChunk* dst = p.bigChunks + (size_t)shard * p.bigChunkCap + ci;
    const int32_t cx0 = ts.px0 >> 6, cx1 = ts.px1 >> 6, cy0 = ts.py0 >> 6, cy1 = ts.py1 >> 6;
    for (int32_t cy = cy0; cy <= cy1; cy++)
        for (int32_t cx = cx0; cx <= cx1; cx++) {
            BigChunk bc; bc.tri = gti; bc.cxy = (uint32_t)cx | ((uint32_t)cy << 16);
            *dst++ = bc;
        }
}

// one lane on its own (clip kernel, rare)
__device__ __forceinline__ void emit_big_single(const RasterParams& p, uint32_t shard, const TriSetup& ts,
                                                const float d[3], bool twoSided)
{
    const uint32_t ti = atomicAdd(&p.counters->bigTriCount[shard], 1u);
    const uint32_t ci = atomicAdd(&p.counters->bigChunkCount[shard], chunk_count(ts));
    write_big(p, shard, ti, ci, ts, d, twoSided);
}

// ---- the per-cluster kernel -------------------------------------------------------------------
enum { K_NONE = 0, K_SMALL = 1, K_BIG = 3, K_CLIP = 4 };
#define SMALL_MAX 16     // per-lane scan up to 16x16 pixels; larger triangles go to the chunk kernel

template <bool SH>
__global__ __launch_bounds__(256) void raster_cluster_kernel(RasterParams p)
{
    __shared__ float sX[4][LDS_VERTS], sY[4][LDS_VERTS], sW[4][LDS_VERTS];
    __shared__ float sU[4][LDS_VERTS], sV[4][LDS_VERTS], sD[4][LDS_VERTS];

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float* lX = sX[wave]; float* lY = sY[wave]; float* lW = sW[wave];
    float* lU = sU[wave]; float* lV = sV[wave]; float* lD = sD[wave];

    const uint32_t count = *p.count;
    const uint32_t listShard = (blockIdx.x * 4u + wave) % CHORD_LIST_SHARDS;

    for (uint32_t c = blockIdx.x * 4u + wave; c < count; c += gridDim.x * 4u) {
        // wave-uniform record fetches (scalarised by the compiler: addresses are uniform)
        const uint32_t cu = __builtin_amdgcn_readfirstlane(c);
        const ChordDrawCmd cmd = p.cmds[cu];
        const uint32_t objectId = __builtin_amdgcn_readfirstlane(cmd.objectId);
        const uint32_t meshletId = __builtin_amdgcn_readfirstlane(cmd.meshletId);
        const uint32_t slot = __builtin_amdgcn_readfirstlane(cmd.slot);
        const DMeshlet* __restrict__ m = &p.meshlets[meshletId];
        const uint32_t vt = __builtin_amdgcn_readfirstlane(m->vertexTriangleCount);
        const uint32_t V = vt & 0xFFu, T = (vt >> 8) & 0xFFu;
        const uint32_t dataOffset = __builtin_amdgcn_readfirstlane(m->dataOffset);
        const uint32_t vertexBase = __builtin_amdgcn_readfirstlane(m->vertexBase);
        const bool twoSided = __builtin_amdgcn_readfirstlane(p.objStatic[objectId].twoSided) != 0;
        const float* __restrict__ mv = p.objFrame[objectId].mvp;
        Mat4 mvp;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) mvp.r[r][cc] = mv[r * 4 + cc];

        // ---- vertex phase: coalesced index + position stream -> clip space -> LDS -------------
        bool notFast = false;
        for (uint32_t i = lane; i < V; i += 64u) {
            const uint32_t vi = p.meshletData[dataOffset + i] + vertexBase;
            const float* __restrict__ pos = p.positions + (size_t)vi * 3;
            const f4 h = mul_mv(mvp, pos[0], pos[1], pos[2], 1.0f);              // mesh_raster.hlsl:99
            const float aw = fabsf(h.w);
            lX[i] = h.x; lY[i] = h.y; lW[i] = h.w;
            lU[i] = h.x / aw * 0.5f + 0.5f;                                      // :159-161
            lV[i] = h.y / aw * -0.5f + 0.5f;
            const bool fast = in_fast_volume(h);
            lD[i] = fast ? h.z / h.w : __builtin_nanf("");
            notFast = notFast || !fast;
        }
        const bool allFast = __ballot(notFast) == 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- triangle phase -------------------------------------------------------------------
        for (uint32_t tb = 0; tb < T; tb += 64u) {
            const uint32_t t = tb + lane;
            int kind = K_NONE;
            TriSetup ts;
            float d[3] = {0.0f, 0.0f, 0.0f};
            if (t < T) {
                const uint32_t packedIdx = p.meshletData[dataOffset + V + t];
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
                    ts.payload = encode_triangle_instance(t, slot);
                    if (!allFast && (d[0] != d[0] || d[1] != d[1] || d[2] != d[2])) {
                        kind = K_CLIP;
                    } else {
                        ts.X[0] = (int32_t)rintf((u0 * p.W) * 256.0f); ts.Y[0] = (int32_t)rintf((v0 * p.H) * 256.0f);
                        ts.X[1] = (int32_t)rintf((u1 * p.W) * 256.0f); ts.Y[1] = (int32_t)rintf((v1 * p.H) * 256.0f);
                        ts.X[2] = (int32_t)rintf((u2 * p.W) * 256.0f); ts.Y[2] = (int32_t)rintf((v2 * p.H) * 256.0f);
                        if (tri_setup(ts, twoSided, p.Wi, p.Hi)) {
                            ts.d0 = d[0]; ts.e1 = d[1] - d[0]; ts.e2 = d[2] - d[0];
                            const int32_t bw = ts.px1 - ts.px0 + 1, bh = ts.py1 - ts.py0 + 1;
                            const int32_t extX = max(ts.X[0], max(ts.X[1], ts.X[2])) - min(ts.X[0], min(ts.X[1], ts.X[2]));
                            const int32_t extY = max(ts.Y[0], max(ts.Y[1], ts.Y[2])) - min(ts.Y[0], min(ts.Y[1], ts.Y[2]));
                            const bool narrow = extX <= (1 << 14) && extY <= (1 << 14);   // 32-bit edge functions are exact
                            if (bw <= SMALL_MAX && bh <= SMALL_MAX && narrow) kind = K_SMALL;
                            else kind = K_BIG;
                        }
                    }
                }
            }

            if (kind == K_SMALL) raster_small<SH>(p, ts, ts.e1, ts.e2);

            // clip list: wave-aggregated append
            {
                const unsigned long long cm = __ballot(kind == K_CLIP);
                if (cm) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&p.counters->clipTriCount, (uint32_t)__popcll(cm));
                    base = bcast(base, 0);
                    if (kind == K_CLIP) {
                        const uint32_t k = base + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull));
                        if (k < p.clipTriCap) { ClipTri ct; ct.cmdIndex = cu; ct.tri = t; p.clipTris[k] = ct; }
                        else atomicOr(&p.counters->overflow, 2u);
                    }
                }
            }
            // big list: one reservation per wave per sub-list (wave-aggregated)
            {
                const unsigned long long bm = __ballot(kind == K_BIG);
                if (bm) {
                    const uint32_t n = kind == K_BIG ? chunk_count(ts) : 0u;
                    const uint32_t incl = wave_incl_scan_u32(n, lane);
                    const uint32_t total = bcast(incl, 63);
                    uint32_t tbase = 0, cbase = 0;
                    if (lane == 0) {
                        tbase = atomicAdd(&p.counters->bigTriCount[listShard], (uint32_t)__popcll(bm));
                        cbase = atomicAdd(&p.counters->bigChunkCount[listShard], total);
                    }
                    tbase = bcast(tbase, 0); cbase = bcast(cbase, 0);
                    if (kind == K_BIG && !(p.debug & DBG_NO_BIG))
                        write_big(p, listShard, tbase + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull)), cbase + incl - n, ts, d, twoSided);
                }
            }
        }
        // LDS of this wave is rewritten by the next cluster: order the reads above before those writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
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

template <bool SH>
__device__ void raster_serial_wide(const RasterParams& p, const TriSetup& ts)
{
    WideEdges w;
    wide_edges(ts, w);
    for (int32_t py = ts.py0; py <= ts.py1; py++) {
        if (!owns_row<SH>(p.shard, py)) continue;
        unsigned long long* row = p.vis + row_base<SH>(p.shard, py, p.Wi);
        for (int32_t px = ts.px0; px <= ts.px1; px++) {
            const int64_t E0 = edge_at(w, 0, px, py), E1 = edge_at(w, 1, px, py), E2 = edge_at(w, 2, px, py);
            if (((E0 + w.bias[0]) | (E1 + w.bias[1]) | (E2 + w.bias[2])) < 0) continue;
            const float l1 = (float)(double)E1 * ts.invA, l2 = (float)(double)E2 * ts.invA;
            const float z = (ts.d0 + l1 * ts.e1) + l2 * ts.e2;
            vis_write(row + px, z, ts.payload);
        }
    }
}

template <bool SH>
__global__ __launch_bounds__(256) void raster_clip_kernel(RasterParams p)
{
    const uint32_t n = min(p.counters->clipTriCount, p.clipTriCap);
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) {
        const ClipTri ct = p.clipTris[k];
        const ChordDrawCmd cmd = p.cmds[ct.cmdIndex];
        const DMeshlet& m = p.meshlets[cmd.meshletId];
        const uint32_t V = m.vertexTriangleCount & 0xFFu;
        const bool twoSided = p.objStatic[cmd.objectId].twoSided != 0;
        const float* mv = p.objFrame[cmd.objectId].mvp;
        Mat4 mvp;
        for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) mvp.r[r][cc] = mv[r * 4 + cc];
        const uint32_t packedIdx = p.meshletData[m.dataOffset + V + ct.tri];
        f4 poly[2][12];
        for (int i = 0; i < 3; i++) {
            const uint32_t li = (packedIdx >> (8 * i)) & 0xFFu;
            const uint32_t vi = p.meshletData[m.dataOffset + li] + m.vertexBase;
            const float* pos = p.positions + (size_t)vi * 3;
            poly[0][i] = mul_mv(mvp, pos[0], pos[1], pos[2], 1.0f);
        }
        int np = 3, cur = 0;
        for (int pl = 0; pl < 6 && np >= 3; pl++) {
            int m2 = 0;
            for (int i = 0; i < np; i++) {
                const f4 P = poly[cur][i], Q = poly[cur][(i + 1) % np];
                const float dp = clip_dist(P, pl), dq = clip_dist(Q, pl);
                const bool pin = dp >= 0.0f, qin = dq >= 0.0f;
                if (pin) poly[cur ^ 1][m2++] = P;
                if (pin && !qin) poly[cur ^ 1][m2++] = clip_intersect(P, Q, dp, dq);
                else if (!pin && qin) poly[cur ^ 1][m2++] = clip_intersect(Q, P, dq, dp);
            }
            np = m2; cur ^= 1;
        }
        if (np < 3) continue;
        bool ok = true;
        int32_t PX[12], PY[12]; float PD[12];
        for (int i = 0; i < np; i++) {
            const f4 h = poly[cur][i];
            if (!(h.w > 0.0f)) { ok = false; break; }
            const float u = h.x / fabsf(h.w) * 0.5f + 0.5f;
            const float v = h.y / fabsf(h.w) * -0.5f + 0.5f;
            PX[i] = (int32_t)rintf((u * p.W) * 256.0f);
            PY[i] = (int32_t)rintf((v * p.H) * 256.0f);
            PD[i] = h.z / h.w;
        }
        if (!ok) continue;
        const uint32_t payload = encode_triangle_instance(ct.tri, cmd.slot);
        for (int i = 1; i + 1 < np; i++) {
            TriSetup ts;
            ts.X[0] = PX[0]; ts.X[1] = PX[i]; ts.X[2] = PX[i + 1];
            ts.Y[0] = PY[0]; ts.Y[1] = PY[i]; ts.Y[2] = PY[i + 1];
            const float d[3] = {PD[0], PD[i], PD[i + 1]};
            ts.payload = payload;
            if (!tri_setup(ts, twoSided, p.Wi, p.Hi)) continue;
            ts.d0 = d[0]; ts.e1 = d[1] - d[0]; ts.e2 = d[2] - d[0];
            const int32_t bw = ts.px1 - ts.px0 + 1, bh = ts.py1 - ts.py0 + 1;
            if (bw <= 16 && bh <= 16) raster_serial_wide<SH>(p, ts);
            else emit_big_single(p, (blockIdx.x * 4u + (threadIdx.x >> 6)) % CHORD_LIST_SHARDS, ts, d, twoSided);
        }
    }
}

// ---- big-triangle chunk kernel ----------------------------------------------------------------
template <bool SH>
__global__ __launch_bounds__(256) void raster_chunk_kernel(RasterParams p)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // lane s holds sub-list s: inclusive prefix of the 64 chunk counts
    const uint32_t mine = min(p.counters->bigChunkCount[lane], p.bigChunkCap);
    const uint32_t incl = wave_incl_scan_u32(mine, lane);
    const uint32_t n = bcast(incl, 63);
    for (uint32_t k = blockIdx.x * 4u + wave; k < n; k += gridDim.x * 4u) {
        const uint32_t ku = __builtin_amdgcn_readfirstlane(k);
        const uint32_t shard = (uint32_t)__popcll(__ballot(incl <= ku));          // first sub-list with incl > k
        const uint32_t local = ku - (bcast(incl, (int)shard) - bcast(mine, (int)shard));
        const BigChunk bc = p.bigChunks[(size_t)shard * p.bigChunkCap + local];
        const uint32_t triIdx = __builtin_amdgcn_readfirstlane(bc.tri);
        const uint32_t cxy = __builtin_amdgcn_readfirstlane(bc.cxy);
        const BigTri* __restrict__ bt = &p.bigTris[triIdx];
        TriSetup ts;
#pragma unroll
        for (int i = 0; i < 3; i++) { ts.X[i] = bt->X[i]; ts.Y[i] = bt->Y[i]; }
        ts.payload = bt->payload;
        const bool twoSided = bt->twoSided != 0;
        if (!tri_setup(ts, twoSided, p.Wi, p.Hi)) continue;
        ts.d0 = bt->d[0]; ts.e1 = bt->d[1] - bt->d[0]; ts.e2 = bt->d[2] - bt->d[0];
        WideEdges we;
        wide_edges(ts, we);
        const int32_t ox = (int32_t)(cxy & 0xFFFFu) * 64, oy = (int32_t)(cxy >> 16) * 64;
        // 64 lanes classify the chunk's 64 tiles
        const int32_t tX = ox + (int32_t)(lane & 7u) * 8, tY = oy + (int32_t)(lane >> 3) * 8;
        unsigned long long tm = __ballot(tile_overlaps(ts, we, tX, tY));
        while (tm) {
            const int tile = __ffsll((long long)tm) - 1;
            tm &= tm - 1ull;
            raster_tile<SH>(p, ts, we, ts.e1, ts.e2, ox + (tile & 7) * 8, oy + (tile >> 3) * 8, lane);
        }
    }
}

// ---- launcher ---------------------------------------------------------------------------------
void launch_raster(ChordCtx* c, const CmdList& in)
{
    RasterParams p;
    p.count = in.count; p.cmds = in.cmds;
    p.objFrame = c->dObjFrame; p.objStatic = c->dObjStatic;
    p.meshlets = c->dMeshlets; p.meshletData = c->dMeshletData; p.positions = c->dPositions;
    p.vis = (unsigned long long*)c->dVis;
    p.W = (float)c->width; p.H = (float)c->height; p.Wi = (int32_t)c->width; p.Hi = (int32_t)c->height;
    p.shard = c->shard;
    p.bigTris = c->dBigTris; p.bigChunks = c->dBigChunks; p.clipTris = c->dClipTris;
    p.bigTriCap = c->bigTriCap / CHORD_LIST_SHARDS; p.bigChunkCap = c->bigChunkCap / CHORD_LIST_SHARDS; p.clipTriCap = c->clipTriCap;
    p.counters = c->dCounters;
    p.debug = c->debugFlags;

    // reset the deferred-list counts (leading bytes of DeviceCounters)
    (void)hipMemsetAsync(c->dCounters, 0, CHORD_COUNTERS_RESET_BYTES, c->stream);

    uint32_t blocks = (in.capacity + 3u) / 4u;
    const uint32_t maxBlocks = (uint32_t)c->numCUs * 6u;
    if (blocks > maxBlocks) blocks = maxBlocks;
    if (blocks < 1) blocks = 1;
    const uint32_t chunkBlocks = (uint32_t)c->numCUs * 8u;
    const uint32_t clipBlocks = (uint32_t)c->numCUs;
    // optional GPU timestamps after each of the three kernels
    const bool sh = c->shard.ranks > 1;
    stamp(c, S_HZBCULL);      // closes whatever preceded the raster (HZB cull / list reset)
    if (sh) hipLaunchKernelGGL(raster_cluster_kernel<true>, dim3(blocks), dim3(256), 0, c->stream, p);
    else    hipLaunchKernelGGL(raster_cluster_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, p);
    stamp(c, S_R_CLUSTER);
    if (sh) hipLaunchKernelGGL(raster_clip_kernel<true>, dim3(clipBlocks), dim3(256), 0, c->stream, p);
    else    hipLaunchKernelGGL(raster_clip_kernel<false>, dim3(clipBlocks), dim3(256), 0, c->stream, p);
    stamp(c, S_R_CLIP);
    if (sh) hipLaunchKernelGGL(raster_chunk_kernel<true>, dim3(chunkBlocks), dim3(256), 0, c->stream, p);
    else    hipLaunchKernelGGL(raster_chunk_kernel<false>, dim3(chunkBlocks), dim3(256), 0, c->stream, p);
    stamp(c, S_R_CHUNK);
    c->rasterCalls++;
}

} // namespace chord
