// Culling kernels of the visibility path, hand-written for gfx950 (wave64).
//
//   object_frame (device fn)    <- instanceCullingCS          instance_culling.hlsl:47-131 (runs inside group_cull_count_kernel)
//   group_cull_count/scatter    <- clusterGroupCullingCS      instance_culling.hlsl:133-208
//   hzb_cull_kernel<PHASE>      <- hzbMainViewCullingCS       hzb_mainview_culling.hlsl:35-213
//
// Design differences from the reference (results identical as sets; slots deterministic):
//  * the reference spends a 64-thread group per object with 63 idle lanes and orders objects by
//    an atomic; here one thread tests one object and the (object, group) expansion is a static
//    prefix table built at upload, so the cluster list is produced in (object, group, meshlet)
//    order by a two-kernel block scan (count -> scatter) instead of wave atomics.  That makes the
//    visibility payload (slot + 1) reproducible run to run and identical on every rank.
//  * one-thread "indirect args" kernels (indirect_cmd.hlsl, pipeline_filter.hlsl:27-38) vanish:
//    count-driven kernels read the device-side count and grid-stride.
//  * record fetches go through pre-resolved offsets (DPrim / DObjStatic) instead of 4 levels of
//    bindless indirection; per-object matrices (VP*M, V*M) are computed once per object per frame.
//
// Memory-bound integer/fp32 VALU work: no MFMA.  Built with -ffp-contract=off.

#include "hzb_device.h"

#include <algorithm>
#include <cstring>

namespace chord {

// The previous frame's HZB tail (mips 6.. of the chain kept as history + its valid range; renderer.cpp:343) rides on the
// first kernel of the NEXT frame as one extra workgroup: it is a one-block, latency-bound job (9 us as a launch of its
// own) whose result nothing needs before the next frame's HZB cull, two kernels later (chordvis_history_hzb and every
// other reader launch it on demand).
struct FrameTail { HzbParams p; uint32_t run; };

// ------------------------------------------------------------------------------ object stage --

// The first kernel of a frame also (a) publishes the frame constants, which arrive as a 600-byte kernel
// argument instead of a host-to-device copy, and (b) zeroes the FrameState block (counters, list counts,
// tile-bin counts) — two launches (~5 us each on this GPU) that the frame no longer pays.
// instanceCullingCS (instance_culling.hlsl:47-131) for one object: OBB-vs-frustum and the per-object matrices every
// later kernel of the frame reads.  Runs inside group_cull_count_kernel (below): a launch of its own cost 8 us of a
// 216-us frame for 352 objects' worth of work.
// (in pieces that take the object's matrices by value, so that frame_cull_fused_kernel's lanes can compute the one piece each of them
// owes its quad with the SAME arithmetic, from records they asked for early)
__device__ __forceinline__ Mat4 obj_mvp(const Mat4& M, const DView& dv) { return mul_mm(load_mat(dv.iv.translatedWorldToClip), M); }   // instance_culling.hlsl:71
__device__ __forceinline__ bool obj_visible(const Mat4& M, const float* posMin, const float* posMax, const DView& dv, const Mat4& mvp)
{
    if (!(dv.flags & CHORD_FLAG_FRUSTUM_CULL)) return true;                 // instance_culling.hlsl:79-89
    const bool ortho = mvp.r[3][3] == 1.0f;                                 // base.hlsli:243-246
    f3 c, e;
    aabb_center_extent(posMin, posMax, c, e);
    if (ortho) return !ortho_frustum_culling(c, e, mvp);
    return !frustum_culling(&dv.iv.frustumPlanesRS[0][0], c, e, M);
}
__device__ __forceinline__ Mat4 obj_mvp_last(const Mat4& Ml, const DView& dv) { return mul_mm(load_mat(dv.view.translatedWorldToClipLastFrame), Ml); }   // hzb_mainview_culling.hlsl:77-83
__device__ __forceinline__ Mat4 obj_local_to_view(const Mat4& M, const DView& dv) { return mul_mm(load_mat(dv.view.translatedWorldToView), M); }        // instance_culling.hlsl:170 -- always the main view
__device__ __forceinline__ f4 obj_cam_ls(const Mat4& W2L) { return mul_mv(W2L, 0.0f, 0.0f, 0.0f, 1.0f); }                                              // nanite_shared.hlsli:65

__device__ __forceinline__ void object_frame(const ChordObject* __restrict__ objects, const DObjStatic* __restrict__ objStatic,
                                             const DPrim* __restrict__ prims, const DView& dv, DObjFrame* __restrict__ objFrame, uint32_t o)
{
    const ChordObject& obj = objects[o];
    const Mat4 M = load_mat(obj.basicData.localToTranslatedWorld);
    const Mat4 mvp = obj_mvp(M, dv);
    const bool ortho = mvp.r[3][3] == 1.0f;                                 // base.hlsli:243-246
    bool visible = true;
    if (dv.flags & CHORD_FLAG_FRUSTUM_CULL) {                               // (the primitive's record is read only then)
        const DPrim& prim = prims[objStatic[o].prim];
        visible = obj_visible(M, prim.posMin, prim.posMax, dv, mvp);
    }

    DObjFrame& of = objFrame[o];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) of.mvp[r * 4 + c] = mvp.r[r][c];
    {
        const Mat4 ml = obj_mvp_last(load_mat(obj.basicData.localToTranslatedWorldLastFrame), dv);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) of.mvpLast[r * 4 + c] = ml.r[r][c];
    }
    {
        const Mat4 l2v = obj_local_to_view(M, dv);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) of.localToView[r * 4 + c] = l2v.r[r][c];
    }
    {
        const f4 cam = obj_cam_ls(load_mat(obj.basicData.translatedWorldToLocal));
        of.camLS[0] = cam.x; of.camLS[1] = cam.y; of.camLS[2] = cam.z;
    }
    of.maxScale = obj.basicData.scaleExtractFromMatrix[3];
    of.visible = visible ? 1u : 0u;
    of.isOrtho = ortho ? 1u : 0u;
}

// ------------------------------------------------------------------------------- group stage --

// projectSphereToScreen(transformSphere(...)) — base.hlsli:233-241, 503-518
__device__ __forceinline__ float projected_error_px(float lodScale, const float* __restrict__ l2v, float maxScale,
                                                    const float* center, float radius)
{
    f3 q;
    q.x = ((l2v[0] * center[0] + l2v[1] * center[1]) + l2v[2] * center[2]) + l2v[3] * 1.0f;
    q.y = ((l2v[4] * center[0] + l2v[5] * center[1]) + l2v[6] * center[2]) + l2v[7] * 1.0f;
    q.z = ((l2v[8] * center[0] + l2v[9] * center[1]) + l2v[10] * center[2]) + l2v[11] * 1.0f;
    const float R = maxScale * radius;
    const float d2 = dot3(q, q);
    const float r2 = R * R;
    if (d2 <= r2) return -1.0f;
    return lodScale * R / sqrtf(d2 - r2);
}

// isMeshletGroupVisibile — nanite_shared.hlsli:15-49
__device__ __forceinline__ bool group_visible(float lodScale, const float* __restrict__ l2v, float maxScale, const DGroup& g)
{
    const bool finalLod = g.parentError > CHORD_ERROR_RADIUS_ROOT;
    const bool firstLod = g.error < -0.5f;
    if (!finalLod) {
        const float pe = projected_error_px(lodScale, l2v, maxScale, g.parentPosCenter, g.parentError);
        if (pe > 0.0f && pe <= CHORD_ERROR_PIXEL_THRESHOLD) return false;
    }
    if (!firstLod) {
        const float er = projected_error_px(lodScale, l2v, maxScale, g.clusterPosCenter, g.error);
        if (er < 0.0f || er > CHORD_ERROR_PIXEL_THRESHOLD) return false;
    }
    return true;
}

// isMeshletVisible — nanite_shared.hlsli:51-91
__device__ __forceinline__ bool meshlet_visible(uint32_t flags, const float* __restrict__ planes, const DObjFrame& of,
                                                const Mat4& M, bool twoSided, const DMeshlet& m)
{
    if (!twoSided && (flags & CHORD_FLAG_CONE_CULL)) {
        const f3 v = {m.coneApex[0] - of.camLS[0], m.coneApex[1] - of.camLS[1], m.coneApex[2] - of.camLS[2]};
        const float len = sqrtf(dot3(v, v));
        const f3 n = {v.x / len, v.y / len, v.z / len};
        const f3 axis = {m.coneAxis[0], m.coneAxis[1], m.coneAxis[2]};
        if (dot3(n, axis) >= m.coneCutOff) return false;
    }
    if (flags & CHORD_FLAG_FRUSTUM_CULL) {
        f3 c, e;
        aabb_center_extent(m.posMin, m.posMax, c, e);
        if (of.isOrtho) {
            Mat4 mvp;
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) mvp.r[r][cc] = of.mvp[r * 4 + cc];
            return !ortho_frustum_culling(c, e, mvp);
        }
        return !frustum_culling(planes, c, e, M);
    }
    return true;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += n;
    }
    return v;
}

// exclusive scan of one value per thread over a 256-thread block; returns (exclusive, blockTotal)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* blockTotal)
{
    __shared__ uint32_t waveSums[4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) waveSums[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t s = waveSums[w];
        if (w < wave) base += s;
        total += s;
    }
    __syncthreads();
    *blockTotal = total;
    return base + incl - v;
}

// totals of two values per thread over a block of BT threads (no scan: the group cull only keeps its blocks' totals)
template <uint32_t BT>
__device__ __forceinline__ void block_totals2(uint32_t a, uint32_t b, uint32_t* totalA, uint32_t* totalB)
{
    __shared__ uint32_t sums[2][BT / 64u];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t ia = wave_incl_scan(a, lane), ib = wave_incl_scan(b, lane);
    if (lane == 63u) { sums[0][wave] = ia; sums[1][wave] = ib; }
    __syncthreads();
    uint32_t ta = 0, tb = 0;
#pragma unroll
    for (uint32_t w = 0; w < BT / 64u; w++) { ta += sums[0][w]; tb += sums[1][w]; }
    *totalA = ta; *totalB = tb;
}

// the value lane L of the quad holds, in every lane of the quad (DPP quad_perm [L,L,L,L]; all lanes active)
template <int L> __device__ __forceinline__ uint32_t quad_bcast_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, L | (L << 2) | (L << 4) | (L << 6), 0xF, 0xF, true); }
template <int L> __device__ __forceinline__ float quad_bcast(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), L | (L << 2) | (L << 4) | (L << 6), 0xF, 0xF, true)); }
// the value of lane ^ 1 / lane ^ 2 of the quad (DPP quad_perm [1,0,3,2] / [2,3,0,1]; all lanes active)
__device__ __forceinline__ uint32_t quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); }

struct GroupCullParams {
    const ChordObject* objects; const DObjStatic* objStatic; const DObjFrame* objFrame; const DPrim* prims;
    const DGroup* groups; const uint32_t* groupIndices; const DMeshlet* meshlets; const DGroupRef* groupRefs;
    const DView* dview; uint8_t* groupMask; uint32_t* blockCounts; uint32_t groupInstances;
    // sharded frames: the commands whose clusters touch one of this rank's screen tiles are ALSO written to the rank's own
    // list (in the same deterministic order), decided where the meshlet record is already in registers
    ShardInfo shard; float W, H; int32_t Wi, Hi; ChordDrawCmd* mineCmds; uint32_t* mineCount;
};

// The screen tiles a cluster may touch.  Conservative: the pixel rectangle of the 8 projected AABB corners, one pixel of slack;
// any corner at or behind the camera plane makes the cluster "unbounded" (every tile).  Returns 0: no tile (the rectangle lies off
// screen), 1: the tile rectangle [tx0, tx1] x [ty0, ty1], 2: unbounded.
__device__ __forceinline__ uint32_t cluster_tile_rect(const float* __restrict__ mv, const DMeshlet& m, float W, float H, int32_t Wi, int32_t Hi,
                                                      uint32_t& tx0, uint32_t& ty0, uint32_t& tx1, uint32_t& ty1)
{
    Mat4 mvp;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int cc = 0; cc < 4; cc++) mvp.r[r][cc] = mv[r * 4 + cc];
    float xlo = 3.0e38f, xhi = -3.0e38f, ylo = 3.0e38f, yhi = -3.0e38f;
    bool unbounded = false;
#pragma unroll
    for (uint32_t q = 0; q < 8u; q++) {
        const f4 h = mul_mv(mvp, (q & 1u) ? m.posMax[0] : m.posMin[0], (q & 2u) ? m.posMax[1] : m.posMin[1],
                            (q & 4u) ? m.posMax[2] : m.posMin[2], 1.0f);
        const float x = (h.x / h.w * 0.5f + 0.5f) * W, y = (h.y / h.w * -0.5f + 0.5f) * H;
        if (!(h.w > 1.0e-6f) || !(fabsf(y) < 1.0e7f) || !(fabsf(x) < 1.0e7f)) unbounded = true;
        else { xlo = fminf(xlo, x); xhi = fmaxf(xhi, x); ylo = fminf(ylo, y); yhi = fmaxf(yhi, y); }
    }
    if (unbounded) return 2u;
    const int32_t x0 = max((int32_t)floorf(xlo) - 1, 0), x1 = min((int32_t)ceilf(xhi) + 1, Wi - 1);
    const int32_t y0 = max((int32_t)floorf(ylo) - 1, 0), y1 = min((int32_t)ceilf(yhi) + 1, Hi - 1);
    if (x1 < x0 || y1 < y0) return 0u;
    tx0 = (uint32_t)x0 >> CHORD_TILE_SHIFT; ty0 = (uint32_t)y0 >> CHORD_TILE_SHIFT;
    tx1 = (uint32_t)x1 >> CHORD_TILE_SHIFT; ty1 = (uint32_t)y1 >> CHORD_TILE_SHIFT;
    return 1u;
}

// Does the cluster touch one of this rank's screen tiles?  Dropping a cluster that fails is invisible in
// the image -- only triangles without an owned pixel go, exactly as the per-tile ownership test of the binning would decide.
__device__ __forceinline__ bool cluster_touches_rank(const ShardInfo& shard, const float* __restrict__ mv, const DMeshlet& m, float W, float H, int32_t Wi, int32_t Hi)
{
    uint32_t tx0, ty0, tx1, ty1;
    const uint32_t kind = cluster_tile_rect(mv, m, W, H, Wi, Hi, tx0, ty0, tx1, ty1);
    if (kind != 1u) return kind == 2u;
    return shard_owns_any_tile(shard, tx0, ty0, tx1, ty1);
}

// The same for EVERY rank at once (sharded cull): bit r = the cluster touches a tile of rank r.  A rectangle of more than 256 tiles
// is given to every rank (conservative: the binning decides per tile); a cluster that passed the culls but whose rectangle lies off
// screen must still be VISIBLE in the exchanged word (it takes a slot of the full list), so it goes to one rank, `nobody`.
__device__ __forceinline__ uint32_t cluster_rank_mask(const uint8_t* __restrict__ tileOwner, uint32_t tilesX, uint32_t ranks, uint32_t nobody,
                                                      const float* __restrict__ mv, const DMeshlet& m, float W, float H, int32_t Wi, int32_t Hi)
{
    const uint32_t all = (1u << ranks) - 1u;
    uint32_t tx0, ty0, tx1, ty1;
    const uint32_t kind = cluster_tile_rect(mv, m, W, H, Wi, Hi, tx0, ty0, tx1, ty1);
    if (kind == 2u) return all;
    if (kind == 0u) return 1u << nobody;
    if ((tx1 - tx0 + 1u) * (ty1 - ty0 + 1u) > 256u) return all;
    uint32_t mask = 0u;
    for (uint32_t ty = ty0; ty <= ty1 && mask != all; ty++)
        for (uint32_t tx = tx0; tx <= tx1; tx++) mask |= 1u << tileOwner[ty * tilesX + tx];
    return mask;
}

// The object pass as a kernel of its own: for long scenes (thousands of count blocks) the fused form below makes every count
// block wait for its objects first, which costs more than the launch it saves (config 4: +4 us).
__global__ __launch_bounds__(256) void object_cull_kernel(const ChordObject* __restrict__ objects, const DObjStatic* __restrict__ objStatic,
                                                          const DPrim* __restrict__ prims, const DView dv, DView* __restrict__ dviewOut,
                                                          DObjFrame* __restrict__ objFrame, uint32_t objectCount,
                                                          uint4* __restrict__ zeroBase, uint32_t zeroVec4, FrameTail tail,
                                                          uint4* __restrict__ zeroBase2, uint32_t zeroVec4b)
{
    if (tail.run && blockIdx.x == gridDim.x - 1u) { hzb_tail_block(tail.p, 1, 1, (uint32_t)CHORD_TILE_SHIFT); return; }
    const uint32_t o = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t i = o; i < zeroVec4b; i += (gridDim.x - tail.run) * 256u) zeroBase2[i] = make_uint4(0u, 0u, 0u, 0u);   // group masks (hierarchical mode)
    if (dviewOut && blockIdx.x == 0) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&dv);
        uint32_t* dst = reinterpret_cast<uint32_t*>(dviewOut);
        for (uint32_t i = threadIdx.x; i < sizeof(DView) / 4u; i += 256u) dst[i] = src[i];
    }
    for (uint32_t i = o; i < zeroVec4; i += (gridDim.x - tail.run) * 256u) zeroBase[i] = make_uint4(0u, 0u, 0u, 0u);
    if (o < objectCount) object_frame(objects, objStatic, prims, dv, objFrame, o);
}

// Short scenes: first the objects this block's group instances belong to (instanceCullingCS, one thread per object; an object whose
// groups span several blocks is done by each of them -- same values), the frame's housekeeping that used to ride on the
// object kernel (view block published for the later kernels, FrameState zeroed), then the groups.
// FROM_MASK: the per-group meshlet masks were written by bvh_cull_kernel (hierarchical mode); this kernel only counts them.
// SHARDED: the rank's share of the list is determined too (the unsharded instantiation is the round-2 kernel: the ownership
// test costs the single-GPU frame nothing)
// FUSED: the short-scene form described above; long scenes run object_cull_kernel first and this kernel without the object
// pass (its matrices are what set the fused form's register count: 94 VGPRs = 5 waves/SIMD).
#ifndef CULL_QUAD
#define CULL_QUAD 1
#endif
#ifndef CULL_FUSED
#define CULL_FUSED 1               // 0: short scenes keep the three launches count / scatter / phase-0 cull (A/B builds)
#endif
template <bool FROM_MASK, bool SHARDED, bool FUSED, bool QUAD = false>
__global__ __launch_bounds__(QUAD ? 1024 : 256) void group_cull_count_kernel(GroupCullParams p, const DView dv, DView* __restrict__ dviewOut,
                                                               DObjFrame* __restrict__ objFrameOut, uint4* __restrict__ zeroBase, uint32_t zeroVec4,
                                                               uint32_t cullBlocks, FrameTail tail)
{
    static_assert(!QUAD || (FUSED && !FROM_MASK), "the four-lane form is the short-scene kernel");
    constexpr uint32_t BT = QUAD ? 1024u : 256u;                    // threads of a block; a block tests 256 group instances either way
    const uint32_t qi = QUAD ? threadIdx.x & 3u : 0u;               // QUAD: the meshlet of its group this lane tests
    const uint32_t t = blockIdx.x * 256u + (QUAD ? threadIdx.x >> 2 : threadIdx.x);
    // QUAD: what the group's tests read that does not depend on this frame's object records -- the reference, the group, the object's
    // matrix and flags, the lane's meshlet -- is requested BEFORE the object phase and its barrier: behind the barrier a thread then
    // waits for the object's frame record alone instead of for three dependent round trips (the compiler moves no load across a barrier)
    DGroupRef refQ; DGroup gQ; Mat4 MQ; DMeshlet mQ; uint32_t matFlagsQ = 0u;
    if (QUAD && !(tail.run && blockIdx.x == cullBlocks) && t < p.groupInstances) {
        refQ = p.groupRefs[t];
        gQ = p.groups[refQ.group & 0x0FFFFFFFu];
        MQ = load_mat(p.objects[refQ.object].basicData.localToTranslatedWorld);
        matFlagsQ = p.objStatic[refQ.object].matFlags;
        mQ = p.meshlets[qi == 0u ? refQ.meshlet[0] : qi == 1u ? refQ.meshlet[1] : qi == 2u ? refQ.meshlet[2] : refQ.meshlet[3]];
    }
    if (FUSED) {
        if (tail.run && blockIdx.x == cullBlocks) {
            if (QUAD && threadIdx.x >= 256u) return;               // (hzb_tail_block is written for 256 threads; whole waves leave, its barriers count the rest)
            hzb_tail_block(tail.p, 1, 1, (uint32_t)CHORD_TILE_SHIFT); return;
        }
        if (dviewOut && blockIdx.x == 0) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(&dv);
            uint32_t* dst = reinterpret_cast<uint32_t*>(dviewOut);
            for (uint32_t i = threadIdx.x; i < sizeof(DView) / 4u; i += BT) dst[i] = src[i];
        }
        for (uint32_t i = blockIdx.x * BT + threadIdx.x; i < zeroVec4; i += cullBlocks * BT) zeroBase[i] = make_uint4(0u, 0u, 0u, 0u);
        if (p.groupInstances && objFrameOut) {
            const uint32_t first = blockIdx.x * 256u;
            if (first < p.groupInstances) {
                const uint32_t oFirst = p.groupRefs[first].object, oLast = p.groupRefs[min(first + 255u, p.groupInstances - 1u)].object;
                // (256 group instances may span more than 256 objects when primitives without groups sit in between)
                for (uint32_t k = threadIdx.x; k <= oLast - oFirst; k += BT) object_frame(p.objects, p.objStatic, p.prims, dv, objFrameOut, oFirst + k);
            }
        }
        __syncthreads();                                   // the object records of this block are written (and visible to it)
    }
    uint32_t mask = 0, tris = 0, mine = 0;
    constexpr bool sharded = SHARDED;
    if (FROM_MASK) {
        if (t < p.groupInstances) mask = p.groupMask[t] & 15u;
        if (mask) {
            const DGroupRef ref = p.groupRefs[t];
            const uint32_t o = ref.object;
            for (uint32_t i = 0; i < CHORD_GROUP_MAX_MESHLETS; i++)
                if (mask & (1u << i)) {
                    const DMeshlet& m = p.meshlets[ref.meshlet[i]];
                    tris += (m.vertexTriangleCount >> 8) & 0xFFu;
                    if (sharded && cluster_touches_rank(p.shard, p.objFrame[o].mvp, m, p.W, p.H, p.Wi, p.Hi)) mine |= 1u << i;
                }
            if (sharded) p.groupMask[t] = (uint8_t)(mask | (mine << 4));
        }
    } else
    if (t < p.groupInstances) {
        // ONE fetch names the owner, the group record and the group's meshlets (DGroupRef, resolved at upload); everything the
        // tests read -- the object's frame record, the group, its (up to) four meshlets -- is a second, independent round trip
        const DGroupRef ref = QUAD ? refQ : p.groupRefs[t];
        const uint32_t o = ref.object;
        const DObjFrame& of = p.objFrame[o];
        if (of.visible) {
            const uint32_t matFlags = QUAD ? matFlagsQ : p.objStatic[o].matFlags;
            const DGroup g = QUAD ? gQ : p.groups[ref.group & 0x0FFFFFFFu];
            const uint32_t cnt = ref.group >> 28;
            if (cnt != 0u && group_visible(dv.view.lodScale, of.localToView, of.maxScale, g)) {     // instance_culling.hlsl:174
                const Mat4 M = QUAD ? MQ : load_mat(p.objects[o].basicData.localToTranslatedWorld);
                if (QUAD) {
                    // one meshlet per lane of the quad (every lane of it has done the group's test): the bits meet below
                    const DMeshlet m = mQ;
                    if (qi < cnt && meshlet_visible(dv.flags, &dv.iv.frustumPlanesRS[0][0], of, M, (matFlags & CHORD_MATFLAG_TWO_SIDED) != 0, m)) {
                        mask = 1u << qi;
                        tris = (m.vertexTriangleCount >> 8) & 0xFFu;
                        if (sharded && cluster_touches_rank(p.shard, of.mvp, m, p.W, p.H, p.Wi, p.Hi)) mine = 1u << qi;
                    }
                } else
#pragma unroll
                for (uint32_t i = 0; i < CHORD_GROUP_MAX_MESHLETS; i++) {
                    const DMeshlet m = p.meshlets[ref.meshlet[i]];                              // :178-180 (a slot beyond the group's count re-reads its first meshlet)
                    if (i < cnt && meshlet_visible(dv.flags, &dv.iv.frustumPlanesRS[0][0], of, M, (matFlags & CHORD_MATFLAG_TWO_SIDED) != 0, m)) {
                        mask |= 1u << i;
                        tris += (m.vertexTriangleCount >> 8) & 0xFFu;
                        if (sharded && cluster_touches_rank(p.shard, of.mvp, m, p.W, p.H, p.Wi, p.Hi)) mine |= 1u << i;
                    }
                }
            }
        }
        if (!QUAD) p.groupMask[t] = (uint8_t)(mask | (mine << 4));        // low nibble: visible meshlets, high nibble: of which this rank's
    }
    if (QUAD) {
        // the quad's four lanes hold one bit / one triangle count each (0 where the group or the lane's meshlet failed, or t is past the
        // list): every lane of the quad ends up with the group's values, lane 0 of it stores and counts them
        mask |= quad_xor1(mask); mask |= quad_xor2(mask);
        mine |= quad_xor1(mine); mine |= quad_xor2(mine);
        tris += quad_xor1(tris); tris += quad_xor2(tris);
        if (qi != 0u) { mask = 0u; mine = 0u; tris = 0u; }
        else if (t < p.groupInstances) p.groupMask[t] = (uint8_t)(mask | (mine << 4));
    }
    uint32_t total, blockTris;
    // (counts below 2^12 per block: the visible count in the low half, the rank's in the high half of one sum)
    block_totals2<BT>(sharded ? (uint32_t)__popc(mask) | ((uint32_t)__popc(mine) << 16) : (uint32_t)__popc(mask), tris, &total, &blockTris);
    if (threadIdx.x == 0) {
        p.blockCounts[blockIdx.x] = total & 0xFFFFu; p.blockCounts[cullBlocks + blockIdx.x] = blockTris;
        if (sharded) p.blockCounts[2u * cullBlocks + blockIdx.x] = total >> 16;
    }
}

// ---- sharded group cull (SURVEY 8e "shard by object range"; the shape it replaces: instance_culling.hlsl:133-208, one dispatch over
// all groups on the one device the reference has) -------------------------------------------------------------------------------
// Every rank used to read every group and cluster record (config 5: 1.1 GB, 0.22 ms of a 2.4-ms 8-rank frame -- the largest term that
// did not shrink with the rank count).  Now rank r tests only the count blocks [r * chunkBlocks, (r + 1) * chunkBlocks) and writes, per
// group instance, ONE WORD: byte i = the set of ranks whose screen tiles meshlet i of the group touches (0 = culled).  What travels
// is dense and indexable -- groupInstances x 4 bytes over all ranks, one fixed-size all-gather, no counts, no variable-length lists --
// and everything else is local again: group_mask_unpack_kernel turns the words into the per-group nibbles (visible | this rank's)
// and the per-block counts the count kernel used to leave behind, the prefix and scatter kernels run unchanged, slots are positions
// in the full order exactly as before, and the full list can still be made later from the masks (launch_full_list).
struct CullMaskParams {
    GroupCullParams g;
    const uint8_t* tileOwner; uint32_t tilesX, ranks;
    uint32_t firstBlock, blockCount, chunkBlocks;   // this launch's range of count blocks; blocks per chunk (the triangle sums sit behind chunkBlocks * 256 words)
    uint32_t* out;                                  // the chunk the range belongs to
};

// QUAD: a rank's share of a mid-sized scene is a short list (config 4 at 8 ranks: 98 k group instances, 385 blocks on a chip that holds
// 1 024 of them at once), and a short list's time is one thread's chain of dependent instructions -- four meshlets tested one after the
// other.  As in the short-scene count kernel, four lanes share a group instance: every lane does the group's test, lane i the cone /
// frustum test and the rank mask of meshlet i, the bytes meet by DPP.  The exchanged words are the same.
template <bool QUAD>
__global__ __launch_bounds__(QUAD ? 1024 : 256) void group_cull_masks_kernel(CullMaskParams q, const DView dv)
{
    const GroupCullParams& p = q.g;
    const uint32_t lb = blockIdx.x;                               // block within the chunk
    const uint32_t qi = QUAD ? threadIdx.x & 3u : 0u, gt = QUAD ? threadIdx.x >> 2 : threadIdx.x;
    const uint32_t t = (q.firstBlock + lb) * 256u + gt;
    uint32_t word = 0, tris = 0;
    if (lb < q.blockCount && t < p.groupInstances) {
        const DGroupRef ref = p.groupRefs[t];
        const uint32_t o = ref.object;
        const DObjFrame& of = p.objFrame[o];
        if (of.visible) {
            const uint32_t matFlags = p.objStatic[o].matFlags;
            const DGroup g = p.groups[ref.group & 0x0FFFFFFFu];
            const uint32_t cnt = ref.group >> 28;
            if (cnt != 0u && group_visible(dv.view.lodScale, of.localToView, of.maxScale, g)) {     // instance_culling.hlsl:174
                const Mat4 M = load_mat(p.objects[o].basicData.localToTranslatedWorld);
                if (QUAD) {
                    const DMeshlet m = p.meshlets[qi == 0u ? ref.meshlet[0] : qi == 1u ? ref.meshlet[1] : qi == 2u ? ref.meshlet[2] : ref.meshlet[3]];
                    if (qi < cnt && meshlet_visible(dv.flags, &dv.iv.frustumPlanesRS[0][0], of, M, (matFlags & CHORD_MATFLAG_TWO_SIDED) != 0, m)) {
                        tris = (m.vertexTriangleCount >> 8) & 0xFFu;
                        word = cluster_rank_mask(q.tileOwner, q.tilesX, q.ranks, t % q.ranks, of.mvp, m, p.W, p.H, p.Wi, p.Hi) << (8u * qi);
                    }
                } else
#pragma unroll
                for (uint32_t i = 0; i < CHORD_GROUP_MAX_MESHLETS; i++) {
                    const DMeshlet m = p.meshlets[ref.meshlet[i]];                              // :178-180
                    if (i < cnt && meshlet_visible(dv.flags, &dv.iv.frustumPlanesRS[0][0], of, M, (matFlags & CHORD_MATFLAG_TWO_SIDED) != 0, m)) {
                        tris += (m.vertexTriangleCount >> 8) & 0xFFu;
                        word |= cluster_rank_mask(q.tileOwner, q.tilesX, q.ranks, t % q.ranks, of.mvp, m, p.W, p.H, p.Wi, p.Hi) << (8u * i);
                    }
                }
            }
        }
    }
    if (QUAD) {
        // (all lanes active: whole quads leave the tests together or hold zeros)
        word |= quad_xor1(word); word |= quad_xor2(word);
        tris += quad_xor1(tris); tris += quad_xor2(tris);
        if (qi != 0u) tris = 0u;
        else q.out[lb * 256u + gt] = word;                        // (blocks beyond the range: zeros -- defined bytes on the wire)
        uint32_t none, blockTris;
        block_totals2<1024u>(0u, tris, &none, &blockTris);
        if (threadIdx.x == 0) q.out[q.chunkBlocks * 256u + lb] = blockTris;
    } else {
        q.out[lb * 256u + threadIdx.x] = word;                    // (blocks beyond the range: zeros -- defined bytes on the wire)
        uint32_t blockTris;
        (void)block_excl_scan(tris, &blockTris);
        if (threadIdx.x == 0) q.out[q.chunkBlocks * 256u + lb] = blockTris;
    }
}

// The exchanged words -> what group_cull_count_kernel<., true, .> leaves behind: groupMask (low nibble visible, high nibble this rank's)
// and the three per-block counts.  4 bytes read + 1 written per group instance.
struct CullUnpackParams {
    const uint32_t* words; uint32_t chunkBlocks, rank;
    uint8_t* groupMask; uint32_t* blockCounts; uint32_t cullBlocks, groupInstances;
};

__global__ __launch_bounds__(256) void group_mask_unpack_kernel(CullUnpackParams p)
{
    const uint32_t b = blockIdx.x, src = b / p.chunkBlocks, lb = b - src * p.chunkBlocks;
    const uint32_t* __restrict__ chunk = p.words + (size_t)src * (p.chunkBlocks * 257u);
    const uint32_t t = b * 256u + threadIdx.x;
    uint32_t vis = 0, mine = 0;
    if (t < p.groupInstances) {
        const uint32_t w = chunk[lb * 256u + threadIdx.x];
#pragma unroll
        for (uint32_t i = 0; i < CHORD_GROUP_MAX_MESHLETS; i++) {
            const uint32_t byte = (w >> (8u * i)) & 0xFFu;
            if (byte) vis |= 1u << i;
            mine |= ((byte >> p.rank) & 1u) << i;
        }
        p.groupMask[t] = (uint8_t)(vis | (mine << 4));
    }
    uint32_t total;
    (void)block_excl_scan((uint32_t)__popc(vis) | ((uint32_t)__popc(mine) << 16), &total);
    if (threadIdx.x == 0) {
        p.blockCounts[b] = total & 0xFFFFu;
        p.blockCounts[p.cullBlocks + b] = chunk[p.chunkBlocks * 256u + lb];
        p.blockCounts[2u * p.cullBlocks + b] = total >> 16;
    }
}

// ---- hierarchical cull (chordvis_set_cull_mode 1) ---------------------------------------------------------------------
// Resident waves walk the GPUBVHNode trees (gltf.h:16-24) the reference builds and never reads: one wave per visible
// object at a time (grid stride over the objects), LEVEL by level -- the nodes are stored breadth first, so a level is a
// contiguous range (its end is kept in every node at upload) -- with one node per lane and the set of live nodes as a
// bitmap in LDS.  A node's sphere bounds the parent-error spheres of every group in its subtree (checked at upload),
// and the projected error of a sphere is monotone in containment, so when the NODE's sphere already projects to at most
// (1 - 1/64) px with the eye outside it, every group beneath fails the "parent is too coarse" half of
// isMeshletGroupVisibile (nanite_shared.hlsli:15-49) -- the 1/64 px margin is far above what the fp32 evaluation of
// either side can differ by -- and its children are never marked live.  The groups of a surviving node run exactly the
// flat test and record their meshlet mask; group_cull_count_kernel<true> / the scatter then build the same command
// array the flat dispatch builds.  The root's own leaves are the un-parented groups (and the parented ones of a
// primitive too small to split): always tested, spread over the wave's lanes.
// (First version: one node per step from a stack -- a 585-node tree cost its wave 2.8 ms of dependent round trips.)
struct BvhCullParams {
    GroupCullParams g;
    const DBVHNode* nodes;
    uint32_t objectCount;
};
#define BVH_LIVE_WORDS 256u                 // nodes beyond 8192 of one tree are simply treated as live (their groups are tested)
#define BVH_NODE_MARGIN (1.0f - 1.0f / 64.0f)

__device__ __forceinline__ uint32_t bvh_group_mask(const GroupCullParams& p, const DView& dv, const DObjFrame& of, const Mat4& M, bool twoSided,
                                                   const DPrim& prim, uint32_t gl)
{
    const DGroup g = p.groups[prim.groupBase + gl];
    uint32_t mask = 0;
    if (group_visible(dv.view.lodScale, of.localToView, of.maxScale, g)) {
        const uint32_t idxBase = prim.groupIndicesBase + g.meshletOffset;
        for (uint32_t i = 0; i < g.meshletCount && i < CHORD_GROUP_MAX_MESHLETS; i++) {
            const uint32_t mi = prim.meshletBase + p.groupIndices[idxBase + i];
            if (meshlet_visible(dv.flags, &dv.iv.frustumPlanesRS[0][0], of, M, twoSided, p.meshlets[mi])) mask |= 1u << i;
        }
    }
    return mask;
}

__global__ __launch_bounds__(256) void bvh_cull_kernel(BvhCullParams bp, const DView dv)
{
    __shared__ uint32_t sLive[4][BVH_LIVE_WORDS];
    const GroupCullParams& p = bp.g;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* live = sLive[wave];
    const uint32_t waves = gridDim.x * 4u;
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
    // work unit = (object, part): part 0 is the root's own leaf groups, parts 1..8 the subtrees of the root's children --
    // a big primitive's tree (the street's ground: 585 nodes, 4 k groups) is shared by nine waves instead of one
    for (uint32_t unit = blockIdx.x * 4u + wave; unit < bp.objectCount * 9u; unit += waves) {
        const uint32_t o = unit / 9u, part = unit - o * 9u;
        const DObjFrame& of = p.objFrame[o];
        if (!__builtin_amdgcn_readfirstlane(of.visible)) continue;
        const DObjStatic st = p.objStatic[o];
        const DPrim& prim = p.prims[st.prim];
        const DBVHNode* __restrict__ nodes = bp.nodes + __builtin_amdgcn_readfirstlane(prim.bvhBase);
        const Mat4 M = load_mat(p.objects[o].basicData.localToTranslatedWorld);
        const bool twoSided = (st.matFlags & CHORD_MATFLAG_TWO_SIDED) != 0;
        const uint32_t N = __builtin_amdgcn_readfirstlane(nodes[0].bvhNodeCount);
        if (part == 0u) {
            // the root's own leaves are always tested (its sphere only decides for its children)
            const uint32_t leafOff = nodes[0].leafGroupOffset, leafCnt = nodes[0].leafGroupCount;
            for (uint32_t gi = lane; gi < leafCnt; gi += 64u) {
                const uint32_t mask = bvh_group_mask(p, dv, of, M, twoSided, prim, leafOff + gi);
                if (mask) p.groupMask[st.groupBase + leafOff + gi] = (uint8_t)mask;   // (the array was zeroed by the frame's first kernel)
            }
            continue;
        }
        const uint32_t top = __builtin_amdgcn_readfirstlane(nodes[0].children[part - 1u]);
        if (top == CHORD_BVH_NO_CHILD) continue;
        {
            const float r = nodes[0].sphere[3];
            if (r > 0.0f) {
                const float pe = projected_error_px(dv.view.lodScale, of.localToView, of.maxScale, nodes[0].sphere, r);
                if (pe > 0.0f && pe <= BVH_NODE_MARGIN) continue;             // every parented group of the primitive is below the threshold
            }
        }
        for (uint32_t w = lane; w < min((N + 31u) >> 5, BVH_LIVE_WORDS); w += 64u) live[w] = 0u;
        WAVE_SYNC();
        if (lane == 0u && (top >> 5) < BVH_LIVE_WORDS) live[top >> 5] = 1u << (top & 31u);
        WAVE_SYNC();
        // ---- level by level: whole levels are scanned, only this subtree's nodes are live
        uint32_t ls = 1u;
        while (ls < N) {
            const uint32_t le = min(__builtin_amdgcn_readfirstlane(nodes[ls].pad), N);     // one past the last node of this level
            bool anyChild = false;
            for (uint32_t base = ls; base < le; base += 64u) {
                const uint32_t n = base + lane;
                bool alive = n < le;
                if (alive && (n >> 5) < BVH_LIVE_WORDS) alive = (live[n >> 5] >> (n & 31u)) & 1u;
                if (!alive) continue;
                const DBVHNode& nd = nodes[n];
                const float pe = projected_error_px(dv.view.lodScale, of.localToView, of.maxScale, nd.sphere, nd.sphere[3]);
                if (pe > 0.0f && pe <= BVH_NODE_MARGIN) continue;                          // the whole subtree is below the threshold
#pragma unroll
                for (uint32_t k = 0; k < CHORD_BVH_WIDTH; k++) {
                    const uint32_t ch = nd.children[k];
                    if (ch != CHORD_BVH_NO_CHILD) { anyChild = true; if ((ch >> 5) < BVH_LIVE_WORDS) atomicOr(&live[ch >> 5], 1u << (ch & 31u)); }
                }
                const uint32_t leafOff = nd.leafGroupOffset, leafCnt = nd.leafGroupCount;  // (fewer than 8 below the root)
                for (uint32_t gi = 0; gi < leafCnt; gi++) {
                    const uint32_t mask = bvh_group_mask(p, dv, of, M, twoSided, prim, leafOff + gi);
                    if (mask) p.groupMask[st.groupBase + leafOff + gi] = (uint8_t)mask;
                }
            }
            WAVE_SYNC();
            if (__ballot(anyChild) == 0ull) break;
            ls = le;
        }
    }
#undef WAVE_SYNC
}

// Long scenes (thousands of count blocks): one workgroup turns the per-block counts into exclusive offsets, so the
// scatter kernel reads its base instead of summing all preceding counts itself (that is quadratic in the number of
// blocks: 34 us at config 4), and adds up the triangles of the list (the Gtri/s unit).
__global__ __launch_bounds__(1024) void group_cull_prefix_kernel(uint32_t* __restrict__ blockCounts, uint32_t blocks,
                                                                 uint32_t* __restrict__ outCount, DeviceCounters* __restrict__ counters,
                                                                 uint32_t* __restrict__ mineCount)
{
    // One workgroup walks the counts in steps of 1 024 (coalesced), the visible counts and -- sharded frames: blockCounts[2 * blocks ..) --
    // the rank's own scanned TOGETHER as one 64-bit value (round 4 walked the array twice), and the next step's counts are requested
    // before this step's scan and barriers, so a step costs the scan, not a memory round trip plus the scan.
    // (Round 5 also tried a thread per contiguous run of blocks with every load up front: strided 4-byte loads, 15 us where this form
    // takes 7 on config 4's 3 076 blocks; profiles/r05_config4_x64_4k_hzb_kernel_stats.csv of that run.)
    __shared__ unsigned long long sWave[16], sTris[16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool two = mineCount != nullptr;
    unsigned long long tris = 0, carry = 0;
    uint32_t nv = threadIdx.x < blocks ? blockCounts[threadIdx.x] : 0u;
    uint32_t nm = (two && threadIdx.x < blocks) ? blockCounts[2u * blocks + threadIdx.x] : 0u;
    uint32_t nt = threadIdx.x < blocks ? blockCounts[blocks + threadIdx.x] : 0u;
    for (uint32_t base = 0; base < blocks; base += 1024u) {
        const uint32_t b = base + threadIdx.x;
        const unsigned long long v = (unsigned long long)nv | ((unsigned long long)nm << 32);
        tris += nt;
        const uint32_t bn = b + 1024u;
        nv = bn < blocks ? blockCounts[bn] : 0u;
        nm = (two && bn < blocks) ? blockCounts[2u * blocks + bn] : 0u;
        nt = bn < blocks ? blockCounts[blocks + bn] : 0u;
        unsigned long long incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(incl, d, 64); if (lane >= (uint32_t)d) incl += o; }
        if (lane == 63u) sWave[wave] = incl;
        __syncthreads();
        unsigned long long before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16u; w++) { const unsigned long long c = sWave[w]; if (w < wave) before += c; all += c; }
        const unsigned long long excl = carry + before + incl - v;
        if (b < blocks) { blockCounts[b] = (uint32_t)excl; if (two) blockCounts[2u * blocks + b] = (uint32_t)(excl >> 32); }
        carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) { *outCount = (uint32_t)carry; if (two) *mineCount = (uint32_t)(carry >> 32); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tris += __shfl_down(tris, off, 64);
    if (lane == 0u) sTris[wave] = tris;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (uint32_t w = 0; w < 16u; w++) t += sTris[w];
        if (t) atomicAdd(&counters->trisInstanceCulled, t);
    }
}

// -------------------------------------------------------------------------------- HZB stage --

struct HzbCullParams {
    const DObjFrame* objFrame; const DMeshlet* meshlets; const DView* dview;
    const uint16_t* hzbMin; ChordHZBDesc desc;
    uint16_t* hzbTailOut;                  // TAIL kernels: block 0 also stores the levels it reduced (the chain stays complete in memory)
    const uint32_t* inCount; const ChordDrawCmd* inCmds;
    uint32_t* visCount; ChordDrawCmd* visCmds;
    uint32_t* rejCount; ChordDrawCmd* rejCmds;
    DeviceCounters* counters;
    uint32_t inCapacity;                   // entries the input list has room for (a command may be read ahead of the count)
};

// The tail of the min chain (levels 6.. from the stored level 5: 60 x 34 texels at 4K) reduced by the calling block into
// LDS, level by level: what hzb_tail_kernel computes (hzb_device.h: 2x2 taps clamped to the valid extent of the level
// below; min of binary16 values is one of them, so f32 copies are exact), without a launch of its own between the first
// raster pass and the phase-1 cull -- a few hundred texels per block against ~6 us of launch + drain.  sOff[l] = offset of
// level l, stored at the level's full pitch like the chain in memory.
#define HZB_TAIL_FIRST 6u
#define HZB_TAIL_FLOATS 1408u              // a 4096^2 target (the largest): 32^2 + 16^2 + 8^2 + 4^2 + 2^2 + 1 = 1365
// (in two parts for callers that have other fetches to wait for: hzb_tail_fetch asks for the level-5 texels of the thread's level-6
// texels -- at most four of them per thread of a 256-thread workgroup -- and hzb_tail_to_lds takes them when it is handed them)
struct HzbTailRegs { uint16_t t[4][4]; };
__device__ __forceinline__ void hzb_tail_fetch(const uint16_t* __restrict__ hzbMin, const ChordHZBDesc& d, uint32_t threads, HzbTailRegs& r)
{
    const uint32_t l = HZB_TAIL_FIRST;
    if (l >= d.mipCount) return;
    const uint32_t vw = valid_w(d, l), vh = valid_h(d, l);
    const uint32_t gw = valid_w(d, l - 1), gh = valid_h(d, l - 1), pmw = max(1u, d.width >> (l - 1));
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t i = threadIdx.x + k * threads;
        if (i < vw * vh) {
            const uint32_t x = i % vw, y = i / vw;
#pragma unroll
            for (int jj = 0; jj < 2; jj++)
#pragma unroll
                for (int ii = 0; ii < 2; ii++) {
                    const uint32_t cx = min(2u * x + ii, gw - 1u), cy = min(2u * y + jj, gh - 1u);
                    r.t[k][jj * 2 + ii] = hzbMin[d.mipOffset[l - 1] + cy * pmw + cx];
                }
        }
    }
}
__device__ __forceinline__ void hzb_tail_to_lds(const uint16_t* __restrict__ hzbMin, uint16_t* out, const ChordHZBDesc& d, float* sTail, uint32_t* sOff, uint32_t threads = 256u,
                                                const HzbTailRegs* fetched = nullptr)
{
    uint32_t off = 0, poff = 0;
    uint32_t l0 = HZB_TAIL_FIRST;
    if (fetched && l0 < d.mipCount && valid_w(d, l0) * valid_h(d, l0) <= 4u * threads) {
        // level 6 from the texels already in registers (same taps, same order as the loop below)
        const uint32_t vw = valid_w(d, l0), vh = valid_h(d, l0), mw = max(1u, d.width >> l0), mh = max(1u, d.height >> l0);
        if (threadIdx.x == 0) sOff[l0] = 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t i = threadIdx.x + k * threads;
            if (i < vw * vh) {
                const uint32_t x = i % vw, y = i / vw;
                float mn = f16_to_f32(fetched->t[k][0]);
                mn = fminf(mn, f16_to_f32(fetched->t[k][1])); mn = fminf(mn, f16_to_f32(fetched->t[k][2])); mn = fminf(mn, f16_to_f32(fetched->t[k][3]));
                sTail[y * mw + x] = mn;
                if (out) out[d.mipOffset[l0] + y * mw + x] = f32_to_f16(mn);
            }
        }
        off = mw * mh;
        l0++;
        __syncthreads();
    }
    for (uint32_t l = l0; l < d.mipCount; l++) {
        const uint32_t vw = valid_w(d, l), vh = valid_h(d, l), mw = max(1u, d.width >> l), mh = max(1u, d.height >> l);
        const uint32_t gw = valid_w(d, l - 1), gh = valid_h(d, l - 1), pmw = max(1u, d.width >> (l - 1));
        if (threadIdx.x == 0) sOff[l] = off;
        for (uint32_t i = threadIdx.x; i < vw * vh; i += threads) {
            const uint32_t x = i % vw, y = i / vw;
            float mn = 0.0f;
#pragma unroll
            for (int jj = 0; jj < 2; jj++)
#pragma unroll
                for (int ii = 0; ii < 2; ii++) {
                    const uint32_t cx = min(2u * x + ii, gw - 1u), cy = min(2u * y + jj, gh - 1u);
                    const float a = l == HZB_TAIL_FIRST ? f16_to_f32(hzbMin[d.mipOffset[l - 1] + cy * pmw + cx]) : sTail[poff + cy * pmw + cx];
                    mn = (ii == 0 && jj == 0) ? a : fminf(mn, a);
                }
            sTail[off + y * mw + x] = mn;
            if (out) out[d.mipOffset[l] + y * mw + x] = f32_to_f16(mn);
        }
        poff = off; off += mw * mh;
        __syncthreads();
    }
}

// min / max over the 8 lanes of an octet (lanes 8j .. 8j + 7 of a wave, all of them active): within the quads, then across the two
// quads of the octet (DPP quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror); every lane ends up with the octet's value
__device__ __forceinline__ float oct_min(float v)
{
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
    return fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
}
__device__ __forceinline__ float oct_max(float v)
{
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
}

// occlusion test of one command (hzb_mainview_culling.hlsl:60-161)
// OCT: the eight lanes of an octet test ONE command together -- lane `sub` projects corner `sub` of the bounds and fetches two of the
// sixteen texels, the minima / maxima meet by DPP, every lane of the octet returns the same answer.  (Short lists: a handful of waves
// on the whole chip, each thread a chain of ~700 dependent instructions -- eight projections with three IEEE divisions each.)  The
// folds are min / max over the same values as the one-thread form's loops: the same result whatever the order (the sign of a zero,
// which the order could change, decides nothing below).
// (hzb_meshlet_visible: the test itself, for a caller that holds the meshlet record and the object's frame record already --
// frame_cull_fused_kernel; hzb_cmd_visible below: from a draw command)
// (hzb_bounds_visible: with the projection matrix in registers already -- frame_cull_fused_kernel asks for it beside the object's other
// fields, one round trip instead of two)
template <int PHASE, bool TAIL = false, bool OCT = false>
__device__ __forceinline__ bool hzb_bounds_visible(const HzbCullParams& p, const DView& dv, const Mat4& mvp, const DMeshlet& m,
                                                   const float* sTail = nullptr, const uint32_t* sTailOff = nullptr, const uint32_t sub = 0u)
{
    bool visible = true;
    {
        {
            if (dv.flags & CHORD_FLAG_HZB_CULL) {
                f3 c, e;
                aabb_center_extent(m.posMin, m.posMax, c, e);

                f3 mx = {-10.0f, -10.0f, -10.0f}, mn = {10.0f, 10.0f, 10.0f};
                if (OCT) {
                    const f3 uvz = project_pos_to_uvz(extent_corner(c, e, (int)sub), mvp);
                    mn.x = fminf(mn.x, oct_min(uvz.x)); mn.y = fminf(mn.y, oct_min(uvz.y)); mn.z = fminf(mn.z, oct_min(uvz.z));
                    mx.x = fmaxf(mx.x, oct_max(uvz.x)); mx.y = fmaxf(mx.y, oct_max(uvz.y)); mx.z = fmaxf(mx.z, oct_max(uvz.z));
                } else
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const f3 uvz = project_pos_to_uvz(extent_corner(c, e, k), mvp);
                    mn.x = fminf(mn.x, uvz.x); mn.y = fminf(mn.y, uvz.y); mn.z = fminf(mn.z, uvz.z);
                    mx.x = fmaxf(mx.x, uvz.x); mx.y = fmaxf(mx.y, uvz.y); mx.z = fmaxf(mx.z, uvz.z);
                }
                const bool zInRange = mx.z < 1.0f && mn.z > 0.0f;
                if (zInRange) {
                    if ((mn.x >= 1.0f || mn.y >= 1.0f) || (mx.x <= 0.0f || mx.y <= 0.0f)) visible = false;
                }
                if (visible && zInRange) {
                    mn.x = saturatef(mn.x); mn.y = saturatef(mn.y);
                    mx.x = saturatef(mx.x); mx.y = saturatef(mx.y);
                    const float W = dv.view.renderDimension[0], H = dv.view.renderDimension[1];
                    int rx = (int)(mn.x * W + 0.5f);
                    int ry = (int)(mn.y * H + 0.5f);
                    int rz = (int)(mx.x * W + -0.5f);
                    int rw = (int)(mx.y * H + -0.5f);
                    rx = max(0, rx); ry = max(0, ry);
                    rz = (int)fminf(W - 1.0f, (float)rz);
                    rw = (int)fminf(H - 1.0f, (float)rw);
                    if (rz < rx || rw < ry) {
                        visible = false;
                    } else {
                        const int mx0 = rx >> 1, my0 = ry >> 1, mz0 = rz >> 1, mw0 = rw >> 1;
                        int lv = max(first_bit_high(mz0 - mx0), first_bit_high(mw0 - my0)) - 1;
                        lv = max(0, lv);
                        if (((mz0 >> lv) - (mx0 >> lv) >= 4) || ((mw0 >> lv) - (my0 >> lv) >= 4)) lv += 1;
                        const int cx = mx0 >> lv, cy = my0 >> lv, cz = mz0 >> lv, cw = mw0 >> lv;
                        const uint32_t mw = max(1u, p.desc.width >> lv);
                        const uint16_t* mip = p.hzbMin + p.desc.mipOffset[lv];
                        float zMin = 10.0f;
                        if (OCT) {
                            // texels (x, y) = (sub >> 1, 2 (sub & 1)) and (sub >> 1, 2 (sub & 1) + 1) of the 4 x 4
                            const int x = (int)(sub >> 1), y0 = (int)((sub & 1u) << 1);
                            const int sx = min(cz, cx + x), sy0 = min(cw, cy + y0), sy1 = min(cw, cy + y0 + 1);
                            float a, b;
                            if (TAIL && lv >= (int)HZB_TAIL_FIRST) {
                                const float* lmip = sTail + sTailOff[lv];
                                a = lmip[(uint32_t)sy0 * mw + (uint32_t)sx]; b = lmip[(uint32_t)sy1 * mw + (uint32_t)sx];
                            } else {
                                a = f16_to_f32(mip[(uint32_t)sy0 * mw + (uint32_t)sx]); b = f16_to_f32(mip[(uint32_t)sy1 * mw + (uint32_t)sx]);
                            }
                            zMin = fminf(zMin, oct_min(fminf(a, b)));
                        } else
                        if (TAIL && lv >= (int)HZB_TAIL_FIRST) {
                            // levels 6.. were reduced by this block into LDS (hzb_tail_to_lds): same values, no launch for them
                            const float* lmip = sTail + sTailOff[lv];
#pragma unroll
                            for (int x = 0; x < 4; x++)
#pragma unroll
                                for (int y = 0; y < 4; y++) {
                                    const int sx = min(cz, cx + x), sy = min(cw, cy + y);
                                    zMin = fminf(zMin, lmip[(uint32_t)sy * mw + (uint32_t)sx]);
                                }
                        } else
#pragma unroll
                        for (int x = 0; x < 4; x++)
#pragma unroll
                            for (int y = 0; y < 4; y++) {
                                const int sx = min(cz, cx + x), sy = min(cw, cy + y);
                                zMin = fminf(zMin, f16_to_f32(mip[(uint32_t)sy * mw + (uint32_t)sx]));
                            }
                        if (zMin > mx.z) visible = false;
                    }
                }
            }
        }
    }
    return visible;
}

template <int PHASE, bool TAIL = false, bool OCT = false>
__device__ __forceinline__ bool hzb_meshlet_visible(const HzbCullParams& p, const DView& dv, const DObjFrame& of, const DMeshlet& m,
                                                    const float* sTail = nullptr, const uint32_t* sTailOff = nullptr, const uint32_t sub = 0u)
{
    if (!(dv.flags & CHORD_FLAG_HZB_CULL)) return true;
    Mat4 mvp;
    const float* src = PHASE == 0 ? of.mvpLast : of.mvp;         // hzb_mainview_culling.hlsl:77-83
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int cc = 0; cc < 4; cc++) mvp.r[r][cc] = src[r * 4 + cc];
    return hzb_bounds_visible<PHASE, TAIL, OCT>(p, dv, mvp, m, sTail, sTailOff, sub);
}

template <int PHASE, bool TAIL = false, bool OCT = false>
__device__ __forceinline__ bool hzb_cmd_visible(const HzbCullParams& p, const DView& dv, const ChordDrawCmd& cmd, uint32_t& tris,
                                                const float* sTail = nullptr, const uint32_t* sTailOff = nullptr, const uint32_t sub = 0u)
{
    const DMeshlet& m = p.meshlets[cmd.meshletId];
    tris = (m.vertexTriangleCount >> 8) & 0xFFu;
    return hzb_meshlet_visible<PHASE, TAIL, OCT>(p, dv, p.objFrame[cmd.objectId], m, sTail, sTailOff, sub);
}

// ---- short scenes on one GPU, frames with a history: instanceCulling AND the phase-0 occlusion cull in ONE kernel -----------------
// (the shapes it replaces: instance_culling.hlsl:47-208 and hzb_mainview_culling.hlsl:35-213 with bFirstStage, three dispatches in the
// reference, three launches -- count, scatter, phase-0 cull: 12 + 5 + 7 us of config 3's 175-us frame -- here until round 6.)
// A short scene's group cull is the four-lane form of group_cull_count_kernel: lane i of a quad holds meshlet i of its group instance,
// i.e. every lane holds at most ONE command -- so the lane can test that command against last frame's HZB right away (its meshlet
// record and its object's frame record are in registers), and what is left of the other two launches is where the commands go:
//   * the full list keeps its deterministic order (slot = position: the visibility ids), so a workgroup needs the number of commands
//     of the workgroups in front of it.  Every workgroup PUBLISHES its counts under the launch's serial in two 64-bit words
//     (relaxed agent-scope atomic stores: the payload and its "ready" stamp are one word, no fence) and adds up its predecessors'
//     words -- thread i polls workgroup i --: one round trip of the memory side instead of a launch.  All workgroups of the grid are
//     resident at once (the host takes this path only for grids of at most one workgroup per CU), so a predecessor always arrives;
//     the spin is bounded anyway (overflow bit 4).
//   * the phase-0 lists (visible / rejected) are placed the same way -- their counts ride in the same words --, which also makes
//     THEIR order deterministic (the three-launch path reserves their space with one atomic per workgroup).
//   * round 3 wrote this look-back once and dropped it because the same kernel zeroes the frame's counters -- among them the list
//     counts and triangle totals it would then write itself, in whatever order the workgroups run.  Those words now sit in ONE 64-byte
//     line of FrameState (triangle totals + list counts) that the bulk zeroing leaves out and the LAST workgroup writes whole, totals
//     where there are any, zeros elsewhere.
//   * levels 6.. of the history chain are the "pending tail" that rides on this very kernel's extra workgroup, so they are not in
//     memory yet: every workgroup reduces them from level 5 into its LDS first (hzb_tail_to_lds: the phase-1 cull does the same with
//     the temporary chain), while its first loads are in flight.
struct FusedCullParams {
    GroupCullParams g;
    HzbCullParams h;                       // history chain (min), lists 1 / 2 and their counts; inCount / inCmds unused
    ChordDrawCmd* outCmds; uint32_t* outCount;      // list 0
    unsigned long long* lookback;          // [cullBlocks][2]
    uint32_t* tailLine;                    // the 64-byte line the last workgroup writes: {trisInstanceCulled, trisHzbVisible0, trisHzbVisible1, pad2} (u64) + listCounts[8]
    uint32_t serial, doHzb, skipVec4First, skipVec4Count, objectCount;
};
#ifndef FUSED_CULL_THREADS
#define FUSED_CULL_THREADS 256u            // a workgroup = 64 group instances, one wave per SIMD of its CU (1024: sixteen waves on one CU, A/B builds)
#endif
#define FUSED_CULL_GROUPS (FUSED_CULL_THREADS / 4u)
#define FUSED_CULL_MAX_BLOCKS 1024u        // look-back words of the buffer

__global__ __launch_bounds__(FUSED_CULL_THREADS) void frame_cull_fused_kernel(FusedCullParams q, const DView dv, DView* __restrict__ dviewOut,
                                                                            DObjFrame* __restrict__ objFrameOut, uint4* __restrict__ zeroBase, uint32_t zeroVec4,
                                                                            uint32_t cullBlocks, FrameTail tail)
{
    constexpr uint32_t BT = FUSED_CULL_THREADS;
    // (profile build: wall-clock ticks of this workgroup's stages behind the look-back words -- tools/fused_cull_profile.py)
#if RASTER_PROFILE
    unsigned long long fc[7];
#define FCLOCK(i) do { fc[i] = wall_clock64(); } while (0)
#else
#define FCLOCK(i) do { } while (0)
#endif
    FCLOCK(0);
    __shared__ float sTail[HZB_TAIL_FLOATS];
    __shared__ uint32_t sTailOff[CHORD_HZB_MAX_MIPS];
    __shared__ uint32_t sWave[BT / 64u];
    __shared__ unsigned long long sSumA[BT / 64u], sSumB[BT / 64u];
    const GroupCullParams& p = q.g;
    const uint32_t qi = threadIdx.x & 3u;
    const uint32_t t = blockIdx.x * FUSED_CULL_GROUPS + (threadIdx.x >> 2);
    // grid: [0, cullBlocks) the tests; then, when there is one, the workgroup of the previous frame's HZB tail; then the object pass
    const bool tailBlock = tail.run && blockIdx.x == cullBlocks;
    if (blockIdx.x >= cullBlocks + tail.run) {
        // ---- the object pass (instanceCullingCS, instance_culling.hlsl:47-131): the per-object records every LATER kernel of the frame
        //      reads.  Workgroups of their own, off everybody's path: the tests below do not wait for them (first version: the block's
        //      objects ahead of its tests, a barrier in between -- 5.8 us of a 20-us workgroup, three dependent round trips and the
        //      matrix products of seven threads with 1 017 waiting) ----
        const uint32_t o = (blockIdx.x - cullBlocks - tail.run) * BT + threadIdx.x;
        if (objFrameOut && o < q.objectCount) object_frame(p.objects, p.objStatic, p.prims, dv, objFrameOut, o);
        return;
    }
    // A quad's four lanes test the four meshlets of ONE group instance of ONE object; what they need of the object -- the rows of V * M
    // for the LOD cut, the camera in local space for the cone test, whether the object is in the frustum at all, last frame's
    // projection for the occlusion test -- they work out among themselves, one piece per lane (the pieces of object_frame, the same
    // arithmetic), and hand round by DPP.  tc: lanes past the list compute on its last entry (all lanes stay active: DPP) and emit nothing.
    const uint32_t tc = min(t, max(p.groupInstances, 1u) - 1u);
    // (level 5 of the history chain, for the tail reduction below: asked for first, it is there when the group's records are)
    HzbTailRegs tailRegs;
    if (q.doHzb && !tailBlock) hzb_tail_fetch(q.h.hzbMin, q.h.desc, BT, tailRegs);
    DGroupRef refQ; DGroup gQ; Mat4 MQ, roleQ; DMeshlet mQ; uint32_t matFlagsQ = 0u;
    float scaleQ = 0.0f, pmnQ[3] = {0.0f, 0.0f, 0.0f}, pmxQ[3] = {0.0f, 0.0f, 0.0f};
    refQ.object = 0u; refQ.group = 0u;
    const bool live = !tailBlock && p.groupInstances != 0u;
    if (live) {
        refQ = p.groupRefs[tc];
        gQ = p.groups[refQ.group & 0x0FFFFFFFu];
        const ChordObject& obj = p.objects[refQ.object];
        MQ = load_mat(obj.basicData.localToTranslatedWorld);
        matFlagsQ = p.objStatic[refQ.object].matFlags;
        mQ = p.meshlets[qi == 0u ? refQ.meshlet[0] : qi == 1u ? refQ.meshlet[1] : qi == 2u ? refQ.meshlet[2] : refQ.meshlet[3]];
        // the lane's piece of the object (below): lane 1 the inverse matrix and the scale, lane 3 last frame's matrix, lane 2 the primitive's bounds
        roleQ = load_mat(qi == 1u ? obj.basicData.translatedWorldToLocal : obj.basicData.localToTranslatedWorldLastFrame);
        scaleQ = obj.basicData.scaleExtractFromMatrix[3];
        if (qi == 2u && (dv.flags & CHORD_FLAG_FRUSTUM_CULL)) {
            const DObjStatic& st = p.objStatic[refQ.object];       // (the primitive's bounds, copied beside the object at upload: no third fetch)
#pragma unroll
            for (int i = 0; i < 3; i++) { pmnQ[i] = st.posMin[i]; pmxQ[i] = st.posMax[i]; }
        }
    }
    if (tailBlock) {
        if (threadIdx.x >= 256u) return;                   // (hzb_tail_block is written for 256 threads; whole waves leave, its barriers count the rest)
        hzb_tail_block(tail.p, 1, 1, (uint32_t)CHORD_TILE_SHIFT); return;
    }
    if (dviewOut && blockIdx.x == 0) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&dv);
        uint32_t* dst = reinterpret_cast<uint32_t*>(dviewOut);
        for (uint32_t i = threadIdx.x; i < sizeof(DView) / 4u; i += BT) dst[i] = src[i];
    }
    // the frame's counters, bin counts, list counts: zero -- but for the line the last workgroup writes below
    for (uint32_t i = blockIdx.x * BT + threadIdx.x; i < zeroVec4; i += cullBlocks * BT)
        if (i - q.skipVec4First >= q.skipVec4Count) zeroBase[i] = make_uint4(0u, 0u, 0u, 0u);
    FCLOCK(1);
    // levels 6.. of the history chain into LDS, while the records above are on their way
    if (q.doHzb) hzb_tail_to_lds(q.h.hzbMin, nullptr, q.h.desc, sTail, sTailOff, BT, &tailRegs);
    __syncthreads();
    FCLOCK(2);

    // ---- the object's pieces, one per lane of the quad ----
    DObjFrame of;                                          // (registers: only what the tests read is ever materialised)
    {
        float own[16];
        uint32_t ownFlag = 0u;
#pragma unroll
        for (int i = 0; i < 16; i++) own[i] = 0.0f;
        if (live) {
            if (qi == 0u) {                                // rows 0..2 of V * M
                const Mat4 l2v = obj_local_to_view(MQ, dv);
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) own[r * 4 + cc] = l2v.r[r][cc];
            } else if (qi == 1u) {                         // camera in local space, largest scale
                const f4 cam = obj_cam_ls(roleQ);
                own[0] = cam.x; own[1] = cam.y; own[2] = cam.z; own[3] = scaleQ;
            } else if (qi == 2u) {                         // VP * M, and the object's own frustum test
                const Mat4 mvp = obj_mvp(MQ, dv);
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) own[r * 4 + cc] = mvp.r[r][cc];
                ownFlag = (obj_visible(MQ, pmnQ, pmxQ, dv, mvp) ? 1u : 0u) | (mvp.r[3][3] == 1.0f ? 2u : 0u);
            } else if (q.doHzb) {                          // VP_last * M_last
                const Mat4 ml = obj_mvp_last(roleQ, dv);
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) own[r * 4 + cc] = ml.r[r][cc];
            }
        }
        // (all lanes active here: whole workgroups are live or not)
#pragma unroll
        for (int i = 0; i < 12; i++) of.localToView[i] = quad_bcast<0>(own[i]);
        of.camLS[0] = quad_bcast<1>(own[0]); of.camLS[1] = quad_bcast<1>(own[1]); of.camLS[2] = quad_bcast<1>(own[2]); of.maxScale = quad_bcast<1>(own[3]);
#pragma unroll
        for (int i = 0; i < 16; i++) of.mvp[i] = quad_bcast<2>(own[i]);
        const uint32_t f2 = quad_bcast_u<2>(ownFlag);
        of.visible = f2 & 1u; of.isOrtho = f2 >> 1;
#pragma unroll
        for (int i = 0; i < 16; i++) of.mvpLast[i] = quad_bcast<3>(own[i]);
    }

    // ---- the tests: group (every lane of the quad), this lane's meshlet, and -- for a meshlet that passed -- last frame's HZB ----
    bool vis = false, hzbVis = false;
    uint32_t tris = 0;
    if (t < p.groupInstances) {
        if (of.visible) {
            const uint32_t cnt = refQ.group >> 28;
            if (cnt != 0u && group_visible(dv.view.lodScale, of.localToView, of.maxScale, gQ)) {     // instance_culling.hlsl:174
                if (qi < cnt && meshlet_visible(dv.flags, &dv.iv.frustumPlanesRS[0][0], of, MQ, (matFlagsQ & CHORD_MATFLAG_TWO_SIDED) != 0, mQ)) {
                    vis = true;
                    tris = (mQ.vertexTriangleCount >> 8) & 0xFFu;
                    hzbVis = !q.doHzb || hzb_meshlet_visible<0, true, false>(q.h, dv, of, mQ, sTail, sTailOff);
                }
            }
        }
    }
    {   // the group's 4-bit mask (what the three-launch path leaves behind for whoever replays the scatter)
        uint32_t mask = vis ? 1u << qi : 0u;
        mask |= quad_xor1(mask); mask |= quad_xor2(mask);
        if (qi == 0u && t < p.groupInstances) p.groupMask[t] = (uint8_t)mask;
    }
    // ---- positions inside the block: thread order = (group instance, meshlet) order = the order of the list ----
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#if RASTER_PROFILE
    __syncthreads();
#endif
    FCLOCK(3);
    const uint32_t mine = (vis ? 1u : 0u) | ((vis && hzbVis) ? 1u << 16 : 0u);
    const uint32_t incl = wave_incl_scan(mine, lane);
    unsigned long long ta = ((unsigned long long)(vis ? tris : 0u)) | ((unsigned long long)((vis && hzbVis) ? tris : 0u) << 32);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ta += __shfl_down(ta, off, 64);
    if (lane == 63u) sWave[wave] = incl;
    if (lane == 0u) sSumB[wave] = ta;
    __syncthreads();
    uint32_t before = 0, all = 0;
    unsigned long long trisAll = 0;
#pragma unroll
    for (uint32_t w = 0; w < BT / 64u; w++) { const uint32_t c = sWave[w]; if (w < wave) before += c; all += c; trisAll += sSumB[w]; }
    const uint32_t excl = before + incl - mine;
    const uint32_t n0 = all & 0xFFFFu, n1 = all >> 16;                      // commands of this block / of which visible in phase 0 (<= 1024)
    const uint32_t t0 = (uint32_t)trisAll, t1 = (uint32_t)(trisAll >> 32); // their triangles (<= 2^17)
    __syncthreads();                                                       // (sSumB is reused below)
    // ---- publish, then add up the workgroups in front of this one ----
    const unsigned long long stampA = (unsigned long long)q.serial << 32, stampB = (unsigned long long)(q.serial & 0xFFFFFFu) << 40;
    FCLOCK(4);
    if (threadIdx.x == 0u) {
        __hip_atomic_store(q.lookback + 2u * blockIdx.x, stampA | n0 | (n1 << 11), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q.lookback + 2u * blockIdx.x + 1u, stampB | ((unsigned long long)t0 << 20) | t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long pa = 0, pb = 0;                                     // (n0 | n1 << 32), (t0 | t1 << 32) of the workgroups threadIdx.x, threadIdx.x + BT, ..
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += BT) {
        unsigned long long a = 0, b = 0;
        uint32_t spins = 0;
        for (;;) {
            a = __hip_atomic_load(q.lookback + 2u * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b = __hip_atomic_load(q.lookback + 2u * j + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((a & 0xFFFFFFFF00000000ull) == stampA && (b & 0xFFFFFF0000000000ull) == stampB) break;
            if (++spins > (1u << 22)) { atomicOr(&q.h.counters->overflow, 16u); a = 0; b = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        pa += (a & 0x7FFull) | (((a >> 11) & 0x7FFull) << 32);
        pb += ((b >> 20) & 0xFFFFFull) | ((b & 0xFFFFFull) << 32);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { pa += __shfl_down(pa, off, 64); pb += __shfl_down(pb, off, 64); }
    if (lane == 0u) { sSumA[wave] = pa; sSumB[wave] = pb; }
    __syncthreads();
    unsigned long long baseA = 0, baseB = 0;
#pragma unroll
    for (uint32_t w = 0; w < BT / 64u; w++) { baseA += sSumA[w]; baseB += sSumB[w]; }
    const uint32_t base0 = (uint32_t)baseA, base1 = (uint32_t)(baseA >> 32), base2 = base0 - base1;
    FCLOCK(5);
    // ---- the commands ----
    if (vis) {
        ChordDrawCmd cmd;
        cmd.objectId = refQ.object;
        cmd.meshletId = qi == 0u ? refQ.meshlet[0] : qi == 1u ? refQ.meshlet[1] : qi == 2u ? refQ.meshlet[2] : refQ.meshlet[3];
        const uint32_t i0 = excl & 0xFFFFu, i1 = excl >> 16;
        cmd.slot = base0 + i0;                                              // instance_culling.hlsl:203-206
        q.outCmds[cmd.slot] = cmd;
        if (q.doHzb) {
            if (hzbVis) q.h.visCmds[base1 + i1] = cmd;
            else q.h.rejCmds[base2 + (i0 - i1)] = cmd;
        }
    }
    // ---- the last workgroup holds the totals: the line of triangle totals and list counts, whole ----
    if (blockIdx.x == cullBlocks - 1u && threadIdx.x == 0u) {
        const uint32_t total0 = base0 + n0, total1 = base1 + n1;
        const unsigned long long tris0 = (uint32_t)baseB + (unsigned long long)t0, tris1 = (baseB >> 32) + (unsigned long long)t1;
        uint4* line = reinterpret_cast<uint4*>(q.tailLine);
        line[0] = make_uint4((uint32_t)tris0, (uint32_t)(tris0 >> 32), q.doHzb ? (uint32_t)tris1 : 0u, q.doHzb ? (uint32_t)(tris1 >> 32) : 0u);   // trisInstanceCulled, trisHzbVisible0
        line[1] = make_uint4(0u, 0u, 0u, 0u);                                                                                                        // trisHzbVisible1, pad2
        line[2] = make_uint4(total0, q.doHzb ? total1 : 0u, q.doHzb ? total0 - total1 : 0u, 0u);                                                     // listCounts[0..3]
        line[3] = make_uint4(0u, 0u, 0u, 0u);                                                                                                        // listCounts[4..7]
    }
#if RASTER_PROFILE
    FCLOCK(6);
    if (threadIdx.x == 0u) for (int i = 0; i < 7; i++) q.lookback[2048u + 8u * blockIdx.x + i] = fc[i];
#endif
#undef FCLOCK
}

// (Round 3: writing the list from the count kernel of short scenes -- every workgroup publishing its counts under a launch
// serial and adding up its predecessors', 5 us of launch for one memory round trip -- was written and not kept: that kernel
// also zeroes the frame's counters, among them the list count and the triangle total it would then write itself, and ordering
// those stores across workgroups needs an agent-scope fence per workgroup that costs what the launch does.)
// PREFIXED: blockCounts already holds exclusive offsets (group_cull_prefix_kernel ran); otherwise every block sums the
// preceding blocks' counts itself (a few hundred blocks: cheaper than one more launch).
// (Tried in round 2: the phase-0 HZB test of every command inside this kernel for short scenes, to save the launch of
// hzb_cull_kernel: 23.8 us against 5 + 7 us -- a thread owns a group's up to four commands and tests them one after
// the other, four dependent chains of loads deep, while the stand-alone kernel has one command per thread.)
// outCmds NULL: only the rank's own list is written (sharded frames inside the library: the full list -- 12 bytes per cluster of the
// whole frame on every rank -- is made when a consumer asks for it, launch_full_list below); countTris 0: a re-run for that purpose.
// (not PREFIXED, since round 6 up to CULL_SELFSUM_MAX_BLOCKS count blocks -- config 4 has 3 076: its prefix launch was 8.3 us of one
// workgroup's work between two kernel boundaries; a block's 12 coalesced loads of counts the L2 holds and one reduction are under a
// microsecond of every block's start.  The triangle total is the count kernel's per-block sums, added up by the last block.)
#define CULL_SELFSUM_MAX_BLOCKS 4096u
template <bool PREFIXED, bool SHARDED>
__global__ __launch_bounds__(256) void group_cull_scatter_kernel(GroupCullParams p, ChordDrawCmd* __restrict__ outCmds,
                                                                 uint32_t* __restrict__ outCount, DeviceCounters* __restrict__ counters, uint32_t countTris)
{
    __shared__ unsigned long long red[4], redTris[4];
    constexpr bool sharded = SHARDED;
    const uint32_t cullBlocks = gridDim.x;
    // (the block's own masks and references are asked for first: the sum of the blocks in front of it is needed last)
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t both = t < p.groupInstances ? p.groupMask[t] : 0u;
    const uint32_t mask = both & 15u, mine = sharded ? both >> 4 : 0u;
    const bool emit = mask && (outCmds || mine);
    DGroupRef ref;
    ref.object = 0u; ref.group = 0u; ref.meshlet[0] = ref.meshlet[1] = ref.meshlet[2] = ref.meshlet[3] = 0u;
    if (!PREFIXED && emit) ref = p.groupRefs[t];
    uint32_t blockBase, mineBase = 0;
    if (PREFIXED) { blockBase = p.blockCounts[blockIdx.x]; if (sharded) mineBase = p.blockCounts[2u * cullBlocks + blockIdx.x]; }
    else {
        // (visible counts in the low half, the rank's own in the high half of one 64-bit sum)
        unsigned long long part = 0;
        for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256u)
            part += (unsigned long long)p.blockCounts[b] | (sharded ? (unsigned long long)p.blockCounts[2u * cullBlocks + b] << 32 : 0ull);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = part;
        __syncthreads();
        const unsigned long long all = (red[0] + red[1]) + (red[2] + red[3]);
        blockBase = (uint32_t)all; if (sharded) mineBase = (uint32_t)(all >> 32);
        __syncthreads();
    }

    uint32_t total;
    const uint32_t offs = block_excl_scan(sharded ? (uint32_t)__popc(mask) | ((uint32_t)__popc(mine) << 16) : (uint32_t)__popc(mask), &total);
    const uint32_t off = offs & 0xFFFFu;
    // (a sharded frame that writes only the rank's own list reads the group's reference -- 24 bytes -- only where the rank has a cluster:
    // an eighth of them on 8 ranks; the slots need nothing but the counts)
    if (emit) {
        if (PREFIXED) ref = p.groupRefs[t];
        uint32_t slot = blockBase + off, mslot = mineBase + (offs >> 16);
        for (uint32_t i = 0; i < CHORD_GROUP_MAX_MESHLETS; i++) {
            if (mask & (1u << i)) {
                ChordDrawCmd cmd;
                cmd.objectId = ref.object;
                cmd.meshletId = ref.meshlet[i];
                cmd.slot = slot;                                            // instance_culling.hlsl:203-206
                if (outCmds) outCmds[slot] = cmd;
                if (sharded && p.mineCmds && (mine & (1u << i))) p.mineCmds[mslot++] = cmd;   // the rank's own list: same order, same slots
                slot++;
            }
        }
    }
    if (PREFIXED) return;
    if (!countTris) return;                                         // (a re-run for the full list: counts and totals are the frame's already)
    if (blockIdx.x != gridDim.x - 1u) return;
    // the last block: the list's length, and the triangles it submits (the Gtri/s unit) -- the count kernel's per-block sums
    if (threadIdx.x == 0) { *outCount = blockBase + (total & 0xFFFFu); if (sharded) *p.mineCount = mineBase + (total >> 16); }
    unsigned long long tris = 0;
    for (uint32_t b = threadIdx.x; b < cullBlocks; b += 256u) tris += p.blockCounts[cullBlocks + b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tris += __shfl_down(tris, off, 64);
    if ((threadIdx.x & 63u) == 0u) redTris[threadIdx.x >> 6] = tris;
    __syncthreads();
    if (threadIdx.x == 0) { const unsigned long long t = (redTris[0] + redTris[1]) + (redTris[2] + redTris[3]); if (t) atomicAdd(&counters->trisInstanceCulled, t); }
}

// -------------------------------------------------------------------------------- HZB stage (kernels) --

// One thread per 4 commands, block-wide scan of the per-thread counts, ONE reservation per list and 1024 commands:
// the reference's one InterlockedAdd per wave and list (hzb_mainview_culling.hlsl:163-185) puts every wave of the
// dispatch on the same two counter words -- 8 120 returning atomics on one 64-byte line at 260 k commands (config 4)
// = 92 us on this GPU (~88/us per line, measured); the kernel took 76 us.  List order is free (cmd.z is carried).
// K = commands per thread: 4 for long lists, 1 for short ones (a short list is latency-bound: one command per thread
// keeps the dependent chain of loads short and still needs only count/256 reservations).
// THREADS: 256, or 1024 for long lists with K = 1 -- the same one reservation per 1 024 commands, but sixteen waves with one
// command each instead of four waves with four dependent chains each (config 4: 260 k commands were 254 workgroups, one per
// CU, i.e. four waves per CU walking 4 x (command -> bounds + matrix -> texels)).
#ifndef HZB_CULL_OCT
#define HZB_CULL_OCT 1
#endif
// OCT: eight lanes per command (hzb_cmd_visible<., ., OCT>), THREADS / 8 commands per workgroup and step -- the form of short lists.
template <int PHASE, uint32_t K, bool TAIL, uint32_t THREADS = 256u, bool OCT = false>
__global__ __launch_bounds__(THREADS) void hzb_cull_kernel(HzbCullParams p)
{
    static_assert(!OCT || K == 1u, "eight lanes per command: one command per octet");
    constexpr uint32_t CPB = OCT ? THREADS / 8u : THREADS * K;      // commands of a workgroup's step
    __shared__ uint32_t sWave[THREADS / 64u], sBase[2];
    __shared__ unsigned long long sTris[THREADS / 64u];
    __shared__ float sTail[TAIL ? HZB_TAIL_FLOATS : 1u];
    __shared__ uint32_t sTailOff[CHORD_HZB_MAX_MIPS];
    // (TAIL: level 5 of the chain for the reduction below, asked for ahead of everything else)
    HzbTailRegs tailRegs;
    if (TAIL && OCT) hzb_tail_fetch(p.hzbMin, p.desc, THREADS, tailRegs);
    const uint32_t count = *p.inCount;
    const DView& dv = *p.dview;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // OCT (short lists: one step per workgroup): the octet's command is asked for TOGETHER with the count, and the two records it names
    // -- the meshlet's bounds, the object's projection -- before the tail reduction below: a chain count -> command -> records ->
    // texels of four dependent round trips behind the reduction's own becomes two beside it.
    ChordDrawCmd cmdF = ChordDrawCmd{0, 0, 0};
    DMeshlet mF;
    Mat4 mvpF;
    if (OCT) {
        const uint32_t i = blockIdx.x * CPB + (threadIdx.x >> 3);
        if (p.inCapacity) cmdF = p.inCmds[min(i, p.inCapacity - 1u)];     // (an entry past the count is whatever an earlier launch left: not used)
        if (i < count) {
            mF = p.meshlets[cmdF.meshletId];
            const float* src = PHASE == 0 ? p.objFrame[cmdF.objectId].mvpLast : p.objFrame[cmdF.objectId].mvp;         // hzb_mainview_culling.hlsl:77-83
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) mvpF.r[r][cc] = src[r * 4 + cc];
        }
    }
    // TAIL: the chain this launch culls against has its levels 0..5 in memory; every block with work reduces the rest
    // itself (block 0 also stores them)
    if (TAIL && (blockIdx.x * CPB < count || blockIdx.x == 0u) && (dv.flags & CHORD_FLAG_HZB_CULL))
        hzb_tail_to_lds(p.hzbMin, blockIdx.x == 0u ? p.hzbTailOut : nullptr, p.desc, sTail, sTailOff, THREADS, (TAIL && OCT) ? &tailRegs : nullptr);
    for (uint32_t base = blockIdx.x * CPB; base < count; base += gridDim.x * CPB) {
        ChordDrawCmd cmd[K];
        uint32_t visBits = 0, rejBits = 0, tris = 0;
        if (OCT) {
            const uint32_t i = base + (threadIdx.x >> 3), sub = threadIdx.x & 7u;
            cmd[0] = ChordDrawCmd{0, 0, 0};
            if (i < count) {                                        // (the same for the eight lanes of an octet)
                if (base != blockIdx.x * CPB) {                     // (a later step of a list longer than the grid: its records now)
                    cmdF = p.inCmds[i];
                    mF = p.meshlets[cmdF.meshletId];
                    const float* src = PHASE == 0 ? p.objFrame[cmdF.objectId].mvpLast : p.objFrame[cmdF.objectId].mvp;
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int cc = 0; cc < 4; cc++) mvpF.r[r][cc] = src[r * 4 + cc];
                }
                cmd[0] = cmdF;
                const uint32_t t = (mF.vertexTriangleCount >> 8) & 0xFFu;
                const bool vis = hzb_bounds_visible<PHASE, TAIL, true>(p, dv, mvpF, mF, sTail, sTailOff, sub);
                if (sub == 0u) { if (vis) { visBits = 1u; tris = t; } else rejBits = 1u; }   // lane 0 of the octet counts and stores the command
            }
        } else
#pragma unroll
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t i = base + k * THREADS + threadIdx.x;
            cmd[k] = ChordDrawCmd{0, 0, 0};
            if (i < count) {
                cmd[k] = p.inCmds[i];
                uint32_t t = 0;
                if (hzb_cmd_visible<PHASE, TAIL>(p, dv, cmd[k], t, sTail, sTailOff)) { visBits |= 1u << k; tris += t; }
                else rejBits |= 1u << k;
            }
        }
        // visible count in the low half, rejected count in the high half (<= 1024 each)
        const uint32_t mine = (uint32_t)__popc(visBits) | ((uint32_t)__popc(rejBits) << 16);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t nb = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= (uint32_t)d) incl += nb; }
        unsigned long long tsum = tris;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tsum += __shfl_down(tsum, off, 64);
        if (lane == 63u) sWave[wave] = incl;
        if (lane == 0u) sTris[wave] = tsum;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < THREADS / 64u; w++) { const uint32_t c = sWave[w]; if (w < wave) before += c; all += c; }
        if (threadIdx.x == 0) {
            const uint32_t nv = all & 0xFFFFu, nr = all >> 16;
            sBase[0] = nv ? atomicAdd(p.visCount, nv) : 0u;
            sBase[1] = (PHASE == 0 && nr) ? atomicAdd(p.rejCount, nr) : 0u;
            unsigned long long t = 0;
#pragma unroll
            for (uint32_t w = 0; w < THREADS / 64u; w++) t += sTris[w];
            if (t) atomicAdd(PHASE == 0 ? &p.counters->trisHzbVisible0 : &p.counters->trisHzbVisible1, t);
        }
        __syncthreads();
        const uint32_t excl = before + incl - mine;
        uint32_t vslot = sBase[0] + (excl & 0xFFFFu), rslot = sBase[1] + (excl >> 16);
#pragma unroll
        for (uint32_t k = 0; k < K; k++) {
            if (visBits & (1u << k)) p.visCmds[vslot++] = cmd[k];
            if (PHASE == 0 && (rejBits & (1u << k))) p.rejCmds[rslot++] = cmd[k];
        }
        __syncthreads();                                             // sWave / sBase are rewritten by the next chunk
    }
}

// ---- one-pass occlusion cull of a (shadow) view -- hzb_culling_generic.hlsl:37-172 --------------------------------------
// Differences from the main-view test: the view is an InstanceCullingViewInfo handed in with the call (instanceViewId /
// instanceViewOffset), the object matrix is moved by the distance between the main camera and the view's camera (:78-81),
// a cluster is only tested when every projected corner lies strictly inside the view volume (:95), the pixel rectangle is
// widened by extentScale (:103), 2 x 2 taps one level lower (:120-144), and there is one output list.  Same block-wide
// reservation as hzb_cull_kernel.
struct HzbCullGenericParams {
    const ChordObject* objects; const DMeshlet* meshlets;
    ChordInstanceCullingView iv;
    float rel[3]; float extentScale; uint32_t flags; uint32_t useLastFrame;
    const uint16_t* hzbMin; ChordHZBDesc desc;
    const uint32_t* inCount; const ChordDrawCmd* inCmds;
    uint32_t* outCount; ChordDrawCmd* outCmds;
};

__global__ __launch_bounds__(256) void hzb_cull_generic_kernel(HzbCullGenericParams p)
{
    __shared__ uint32_t sWave[4], sBase;
    const uint32_t count = *p.inCount;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * 256u; base < count; base += gridDim.x * 256u) {
        const uint32_t i = base + threadIdx.x;
        ChordDrawCmd cmd = ChordDrawCmd{0, 0, 0};
        bool visible = false;
        if (i < count) {
            cmd = p.inCmds[i];
            visible = true;
            if (p.flags & CHORD_FLAG_HZB_CULL) {
                const DMeshlet& m = p.meshlets[cmd.meshletId];
                f3 c, e;
                aabb_center_extent(m.posMin, m.posMax, c, e);
                const ChordObject& obj = p.objects[cmd.objectId];
                Mat4 l2w = load_mat(p.useLastFrame ? obj.basicData.localToTranslatedWorldLastFrame : obj.basicData.localToTranslatedWorld);
                l2w.r[0][3] += p.rel[0]; l2w.r[1][3] += p.rel[1]; l2w.r[2][3] += p.rel[2];
                const Mat4 mvp = mul_mm(load_mat(p.iv.translatedWorldToClip), l2w);
                f3 mx = {-10.0f, -10.0f, -10.0f}, mn = {10.0f, 10.0f, 10.0f};
                bool can = true;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const f3 uvz = project_pos_to_uvz(extent_corner(c, e, k), mvp);
                    mn.x = fminf(mn.x, uvz.x); mn.y = fminf(mn.y, uvz.y); mn.z = fminf(mn.z, uvz.z);
                    mx.x = fmaxf(mx.x, uvz.x); mx.y = fmaxf(mx.y, uvz.y); mx.z = fmaxf(mx.z, uvz.z);
                    can = can && (uvz.x < 1.0f && uvz.y < 1.0f && uvz.z < 1.0f) && (uvz.x > 0.0f && uvz.y > 0.0f && uvz.z > 0.0f);
                }
                if (can) {
                    const float W = p.iv.renderDimension[0], H = p.iv.renderDimension[1];
                    int rx = (int)(mn.x * W + p.extentScale * -0.5f);
                    int ry = (int)(mn.y * H + p.extentScale * -0.5f);
                    int rz = (int)(mx.x * W + p.extentScale * 0.5f);
                    int rw = (int)(mx.y * H + p.extentScale * 0.5f);
                    rx = max(0, rx); ry = max(0, ry);
                    rz = (int)fminf(W - 1.0f, (float)rz);
                    rw = (int)fminf(H - 1.0f, (float)rw);
                    if (rz < rx || rw < ry) visible = false;
                    else {
                        const int mx0 = rx >> 1, my0 = ry >> 1, mz0 = rz >> 1, mw0 = rw >> 1;
                        int lv = max(0, max(first_bit_high(mz0 - mx0), first_bit_high(mw0 - my0)));
                        if (((mz0 >> lv) - (mx0 >> lv) >= 2) || ((mw0 >> lv) - (my0 >> lv) >= 2)) lv += 1;
                        lv = min(lv, (int)p.desc.mipCount - 1);
                        const int cx = mx0 >> lv, cy = my0 >> lv, cz = mz0 >> lv, cw = mw0 >> lv;
                        const uint32_t mw = max(1u, p.desc.width >> lv);
                        const uint16_t* mip = p.hzbMin + p.desc.mipOffset[lv];
                        float zMin = 10.0f;
#pragma unroll
                        for (int x = 0; x < 2; x++)
#pragma unroll
                            for (int y = 0; y < 2; y++) {
                                const int sx = min(cz, cx + x), sy = min(cw, cy + y);
                                zMin = fminf(zMin, f16_to_f32(mip[(uint32_t)sy * mw + (uint32_t)sx]));
                            }
                        if (zMin > mx.z) visible = false;
                    }
                }
            }
        }
        const uint32_t mine = visible ? 1u : 0u;
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t nb = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= (uint32_t)d) incl += nb; }
        if (lane == 63u) sWave[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4u; w++) { const uint32_t cc = sWave[w]; if (w < wave) before += cc; all += cc; }
        if (threadIdx.x == 0) sBase = all ? atomicAdd(p.outCount, all) : 0u;
        __syncthreads();
        if (visible) p.outCmds[sBase + before + incl - mine] = cmd;
        __syncthreads();
    }
}

void launch_hzb_cull_generic(ChordCtx* c, const HzbBuffers& hzb, const ChordInstanceCullingView& iv, const float rel[3], float extentScale,
                             bool useLastFrame, const CmdList& in, const CmdList& out)
{
    HzbCullGenericParams p;
    p.objects = c->dObjects; p.meshlets = c->dMeshlets; p.iv = iv;
    p.rel[0] = rel[0]; p.rel[1] = rel[1]; p.rel[2] = rel[2]; p.extentScale = extentScale; p.flags = c->hView.flags; p.useLastFrame = useLastFrame ? 1u : 0u;
    p.hzbMin = hzb.minTexels; p.desc = hzb.desc;
    p.inCount = in.count; p.inCmds = in.cmds; p.outCount = out.count; p.outCmds = out.cmds;
    uint32_t blocks = (in.capacity + 255u) / 256u;
    blocks = std::max(1u, std::min(blocks, (uint32_t)c->numCUs * 8u));
    CHORD_LAUNCH(c, hzb_cull_generic_kernel, dim3(blocks), dim3(256), 0, c->stream, p);
}

// ---------------------------------------------------------------------------------- launchers --

static HzbCullParams make_hzb_cull_params(ChordCtx* c, const HzbBuffers& hzb, const CmdList& in, const CmdList& outVisible, const CmdList* outRejected)
{
    HzbCullParams p;
    p.objFrame = c->dObjFrame; p.meshlets = c->dMeshlets; p.dview = c->dView;
    p.hzbMin = hzb.minTexels; p.desc = hzb.desc; p.hzbTailOut = hzb.minTexels;
    p.inCount = in.count; p.inCmds = in.cmds; p.inCapacity = in.capacity;
    p.visCount = outVisible.count; p.visCmds = outVisible.cmds;
    p.rejCount = outRejected ? outRejected->count : nullptr;
    p.rejCmds = outRejected ? outRejected->cmds : nullptr;
    p.counters = c->dCounters;
    return p;
}

static GroupCullParams make_group_cull_params(ChordCtx* c)
{
    GroupCullParams p;
    p.objects = c->dObjects; p.objStatic = c->dObjStatic; p.objFrame = c->dObjFrame; p.prims = c->dPrims;
    p.groups = c->dGroups; p.groupIndices = c->dGroupIndices; p.meshlets = c->dMeshlets; p.groupRefs = c->dGroupRefs;
    p.dview = c->dView; p.groupMask = c->dGroupMask; p.blockCounts = c->dBlockCounts; p.groupInstances = c->groupInstances;
    p.shard = c->shard; p.W = (float)c->width; p.H = (float)c->height; p.Wi = (int32_t)c->width; p.Hi = (int32_t)c->height; p.mineCmds = nullptr; p.mineCount = nullptr;
    return p;
}

// The full post-cull list of a frame whose group cull wrote only the rank's own (c->fullListStale): the scatter kernel once
// more over the masks and block counts the cull left behind -- no other rank is asked, every rank culled every group.
void launch_full_list(ChordCtx* c)
{
    if (!c->fullListStale) return;
    GroupCullParams p = make_group_cull_params(c);
    const uint32_t blocks = c->cullBlocks;
    const CmdList& out = c->lists[0];
    // (the masks carry the rank's nibble too; the kernel's SHARDED form scans both halves, mineCmds NULL: nothing of the rank's list is rewritten)
    if (blocks > CULL_SELFSUM_MAX_BLOCKS) CHORD_LAUNCH(c, (group_cull_scatter_kernel<true, false>), dim3(blocks), dim3(256), 0, c->stream, p, out.cmds, out.count, c->dCounters, 0u);
    else                                  CHORD_LAUNCH(c, (group_cull_scatter_kernel<false, false>), dim3(blocks), dim3(256), 0, c->stream, p, out.cmds, out.count, c->dCounters, 0u);
    c->fullListStale = false;
}

bool cull_shardable_config(const ChordCtx* c)
{
    const bool hier = c->cullMode == 1 && c->bvhComplete && c->dBvhNodes;
    return c->sceneLoaded && c->shard.ranks > 1u && c->shard.ranks <= 8u && !hier && c->height && c->shard.ownedRows && c->dTileOwner && !(c->debugFlags & 524288u);
}
bool cull_shardable(const ChordCtx* c)
{
    // (the buffer is made by the set-up calls -- prepare_cull_exchange --: no frame allocates, so no rank can drop out of the exchange
    // on its own)
    const uint32_t N = c->shard.ranks;
    return cull_shardable_config(c) && c->dCullExchange && c->cullExchangeRanks == N && c->cullChunkBlocks == (c->cullBlocks + N - 1u) / N;
}

static FrameTail take_pending_tail(ChordCtx* c)
{
    FrameTail tail;
    std::memset(&tail, 0, sizeof(tail));
    if (c->pendingTailSlot) {                              // the previous frame's buildHZB tail: one extra workgroup
        tail.p = make_hzb_tail_params(c, c->hzb[c->pendingTailSlot]);
        tail.run = 1u;
        c->pendingTailSlot = 0;
    }
    return tail;
}

// Sharded cull, first half (chordvis_frame_phase_cull): the object pass (every object on every rank: a rank's clusters may belong to
// any of them; it also publishes the view, zeroes the frame's counters and carries the previous frame's HZB tail) and the group tests
// of THIS rank's range of count blocks, into the rank's chunk of the exchange buffer.  wholeRange: every rank's chunk (measurement and
// tests on one device: tools/shard_time.py fills the peers' chunks once).
void launch_cull_masks(ChordCtx* c, bool wholeRange)
{
    uint4* zeroBase = nullptr;
    uint32_t zeroVec4 = 0;
    if (c->zeroFrameStateInCull) {
        zeroBase = reinterpret_cast<uint4*>(c->dFrameState);
        zeroVec4 = (uint32_t)((c->frameStateZeroBytes + 15u) / 16u);
        c->zeroFrameStateInCull = false;
    }
    DView* publish = c->viewDirty ? c->dView : (DView*)nullptr;
    const FrameTail tail = take_pending_tail(c);
    CHORD_LAUNCH(c, object_cull_kernel, dim3((c->objectCount + 255u) / 256u + tail.run), dim3(256), 0, c->stream, c->dObjects, c->dObjStatic,
                       c->dPrims, c->hView, publish, c->dObjFrame, c->objectCount, zeroBase, zeroVec4, tail, (uint4*)nullptr, 0u);
    c->viewDirty = false;
    CullMaskParams q;
    q.g = make_group_cull_params(c);
    q.tileOwner = c->dTileOwner; q.tilesX = c->tilesX; q.ranks = c->shard.ranks; q.chunkBlocks = c->cullChunkBlocks;
    const size_t chunkWords = (size_t)c->cullChunkBlocks * 257u;
    for (uint32_t r = wholeRange ? 0u : c->shard.rank; r < (wholeRange ? c->shard.ranks : c->shard.rank + 1u); r++) {
        q.firstBlock = r * c->cullChunkBlocks;
        q.blockCount = q.firstBlock < c->cullBlocks ? std::min(c->cullChunkBlocks, c->cullBlocks - q.firstBlock) : 0u;
        q.out = c->dCullExchange + (size_t)r * chunkWords;
        // (a share of at most 512 count blocks is a short list: four lanes per group instance)
        if (CULL_QUAD && c->cullChunkBlocks <= 512u) CHORD_LAUNCH(c, group_cull_masks_kernel<true>, dim3(c->cullChunkBlocks), dim3(1024), 0, c->stream, q, c->hView);
        else                                        CHORD_LAUNCH(c, group_cull_masks_kernel<false>, dim3(c->cullChunkBlocks), dim3(256), 0, c->stream, q, c->hView);
    }
}

void launch_group_cull(ChordCtx* c, const CmdList& out)
{
    GroupCullParams p = make_group_cull_params(c);
    c->mineValid = false;
    if (c->shard.ranks > 1 && c->height && c->shard.ownedRows) {
        // sharded frame: the rank's own commands leave the cull as a list of their own (no pass over the full list later)
        if (!c->dMineCmds && hipMalloc((void**)&c->dMineCmds, sizeof(ChordDrawCmd) * (size_t)c->cmdCapacity) != hipSuccess) c->dMineCmds = nullptr;
        if (c->dMineCmds) { p.mineCmds = c->dMineCmds; p.mineCount = c->dCounts + 4; c->mineValid = true; }
        else p.shard.ranks = 1;                              // (allocation failed: launch_raster falls back to the rank filter)
    } else p.shard.ranks = 1;
    const uint32_t blocks = c->cullBlocks;
    const bool sh = p.shard.ranks > 1u;
    if (c->cullPhaseDone) {
        // sharded cull, second half: the ranks' words have been all-gathered -- masks and block counts come out of them
        CullUnpackParams u;
        u.words = c->dCullExchange; u.chunkBlocks = c->cullChunkBlocks; u.rank = c->shard.rank;
        u.groupMask = c->dGroupMask; u.blockCounts = c->dBlockCounts; u.cullBlocks = blocks; u.groupInstances = c->groupInstances;
        CHORD_LAUNCH(c, group_mask_unpack_kernel, dim3(blocks), dim3(256), 0, c->stream, u);
        c->cullPhaseDone = false;
    } else {
    uint4* zeroBase = nullptr;
    uint32_t zeroVec4 = 0;
    if (c->zeroFrameStateInCull) {
        zeroBase = reinterpret_cast<uint4*>(c->dFrameState);
        zeroVec4 = (uint32_t)((c->frameStateZeroBytes + 15u) / 16u);
        c->zeroFrameStateInCull = false;
    }
    DView* publish = c->viewDirty ? c->dView : (DView*)nullptr;
    const FrameTail tail = take_pending_tail(c);
    FrameTail none;
    std::memset(&none, 0, sizeof(none));
    const bool hier = c->cullMode == 1 && c->bvhComplete && c->dBvhNodes;
    // The fused short-scene kernel (one GPU, flat cull, inside chordvis_render_frame, a grid the device holds at once: its workgroups
    // wait for each other's counts).  CHORDVIS_CULL_FUSED=0: the three-launch path (A/B runs); the result is the same lists, list 0
    // in the same order -- the visibility ids do not depend on the path.
    static const bool fusedOn = [] { const char* e = getenv("CHORDVIS_CULL_FUSED"); return !e || atoi(e) != 0; }();
    const HzbBuffers* fuseHzb = nullptr;
    bool fused = false;
    // (its own workgroup size: fblocks workgroups of FUSED_CULL_GROUPS group instances; a workgroup waits for those in front of it only,
    // which were dispatched before it -- the bound on the grid is the look-back buffer and what a short scene is, not residency)
    const uint32_t fblocks = (c->groupInstances + FUSED_CULL_GROUPS - 1u) / FUSED_CULL_GROUPS;
    const uint32_t fobj = (c->objectCount + FUSED_CULL_THREADS - 1u) / FUSED_CULL_THREADS;
    if (CULL_FUSED && fusedOn && CULL_QUAD && !sh && !hier && c->inFrame && c->fuseCullFrame && out.cmds == c->lists[0].cmds && fblocks + 1u + fobj <= (uint32_t)c->numCUs * (1024u / FUSED_CULL_THREADS) && fblocks <= FUSED_CULL_MAX_BLOCKS && blocks <= 512u &&
        !(c->debugFlags & ~(32768u | 65536u | 262144u))) {
        uint32_t tailFloats = 0;
        const ChordHZBDesc& hd = c->hzb[0].desc;
        for (uint32_t l = 6; l < hd.mipCount; l++) tailFloats += std::max(1u, hd.width >> l) * std::max(1u, hd.height >> l);
        if (!c->dCullLookback && hipMalloc((void**)&c->dCullLookback, sizeof(unsigned long long) * (2048u + 8u * 1024u)) == hipSuccess)      // (look-back words; behind them the profile build's stage clocks)
            (void)hipMemsetAsync(c->dCullLookback, 0, sizeof(unsigned long long) * (2048u + 8u * 1024u), c->stream);
        if (c->dCullLookback && tailFloats <= HZB_TAIL_FLOATS) { fused = true; fuseHzb = c->fuseCullHzb; }
    }
#define LAUNCH_COUNT(FM, FUSED, grid, ...) do { if (sh) CHORD_LAUNCH(c, (group_cull_count_kernel<FM, true, FUSED>), grid, dim3(256), 0, c->stream, __VA_ARGS__); \
                                                 else    CHORD_LAUNCH(c, (group_cull_count_kernel<FM, false, FUSED>), grid, dim3(256), 0, c->stream, __VA_ARGS__); } while (0)
    if (blocks > 512u || hier) {
        // (the mask array is a multiple of 16 bytes long: dalloc rounds nothing, so the tail is zeroed by the last partial vector
        // only when it exists -- the buffer is allocated with 16 bytes of slack, see upload_scene)
        CHORD_LAUNCH(c, object_cull_kernel, dim3((c->objectCount + 255u) / 256u + tail.run), dim3(256), 0, c->stream, c->dObjects, c->dObjStatic,
                           c->dPrims, c->hView, publish, c->dObjFrame, c->objectCount, zeroBase, zeroVec4, tail,
                           reinterpret_cast<uint4*>(c->dGroupMask), hier ? (c->groupInstances + 15u) / 16u : 0u);
        if (hier) {
            BvhCullParams bp;
            bp.g = p; bp.nodes = c->dBvhNodes; bp.objectCount = c->objectCount;
            const uint32_t bb = std::min((c->objectCount * 9u + 3u) / 4u, (uint32_t)c->numCUs * 8u);
            CHORD_LAUNCH(c, bvh_cull_kernel, dim3(std::max(bb, 1u)), dim3(256), 0, c->stream, bp, c->hView);
            LAUNCH_COUNT(true, false, dim3(blocks), p, c->hView, (DView*)nullptr, (DObjFrame*)nullptr, (uint4*)nullptr, 0u, blocks, none);
        } else {
            LAUNCH_COUNT(false, false, dim3(blocks), p, c->hView, (DView*)nullptr, (DObjFrame*)nullptr, (uint4*)nullptr, 0u, blocks, none);
        }
    } else if (fused) {
        // ... and a grid of at most one workgroup per CU does the whole of instanceCulling -- and, in a frame with a history, the phase-0
        // occlusion cull -- in ONE kernel (frame_cull_fused_kernel: no scatter launch, no hzb_cull_kernel<0> launch)
        FusedCullParams q;
        q.g = p;
        q.h = make_hzb_cull_params(c, fuseHzb ? *fuseHzb : c->hzb[0], out, c->lists[1], &c->lists[2]);
        q.outCmds = out.cmds; q.outCount = out.count;
        q.lookback = c->dCullLookback;
        q.tailLine = reinterpret_cast<uint32_t*>(&c->dFrameState->counters.trisInstanceCulled);
        do { ++c->cullSerial; } while ((c->cullSerial & 0xFFFFFFu) == 0u);          // (a stamp of 0 would match words no launch has written)
        q.serial = c->cullSerial; q.doHzb = fuseHzb ? 1u : 0u;
        q.skipVec4First = (uint32_t)(offsetof(FrameState, counters.trisInstanceCulled) / 16u); q.skipVec4Count = 4u;
        q.objectCount = c->objectCount;
        CHORD_LAUNCH(c, frame_cull_fused_kernel, dim3(fblocks + tail.run + fobj), dim3(FUSED_CULL_THREADS), 0, c->stream,
                     q, c->hView, publish, c->dObjFrame, zeroBase, zeroVec4, fblocks, tail);
        if (fuseHzb) { c->fusedCullDone = true; c->listMine[1] = c->listMine[2] = false; }
        c->viewDirty = false;
        c->fullListStale = false;
        return;
    } else {
        // short scenes on one GPU: four lanes per group instance, one meshlet each (CULL_QUAD) -- a list this short leaves most SIMDs
        // with one wave or none, and a thread that tests a group's four meshlets one after the other is 3 000 dependent VALU
        // instructions long
        if (CULL_QUAD && !sh) CHORD_LAUNCH(c, (group_cull_count_kernel<false, false, true, true>), dim3(blocks + tail.run), dim3(1024), 0, c->stream,
                                            p, c->hView, publish, c->dObjFrame, zeroBase, zeroVec4, blocks, tail);
        else LAUNCH_COUNT(false, true, dim3(blocks + tail.run), p, c->hView, publish, c->dObjFrame, zeroBase, zeroVec4, blocks, tail);
    }
#undef LAUNCH_COUNT
    c->viewDirty = false;
    }
    // sharded frames inside the library (frame_phase_a): the full list is written when a consumer asks (launch_full_list)
    ChordDrawCmd* full = out.cmds;
    c->fullListStale = false;
    if (sh && c->lazyFullList && out.cmds == c->lists[0].cmds) { full = nullptr; c->fullListStale = true; }
    if (blocks > CULL_SELFSUM_MAX_BLOCKS) {
        CHORD_LAUNCH(c, group_cull_prefix_kernel, dim3(1), dim3(1024), 0, c->stream, c->dBlockCounts, blocks, out.count, c->dCounters, p.mineCount);
        if (sh) CHORD_LAUNCH(c, (group_cull_scatter_kernel<true, true>), dim3(blocks), dim3(256), 0, c->stream, p, full, out.count, c->dCounters, 1u);
        else    CHORD_LAUNCH(c, (group_cull_scatter_kernel<true, false>), dim3(blocks), dim3(256), 0, c->stream, p, full, out.count, c->dCounters, 1u);
    } else {
        if (sh) CHORD_LAUNCH(c, (group_cull_scatter_kernel<false, true>), dim3(blocks), dim3(256), 0, c->stream, p, full, out.count, c->dCounters, 1u);
        else    CHORD_LAUNCH(c, (group_cull_scatter_kernel<false, false>), dim3(blocks), dim3(256), 0, c->stream, p, full, out.count, c->dCounters, 1u);
    }
}

// ---- sharded frames: this rank's clusters of a raster pass -----------------------------------------------------
// One thread per command: a cluster whose projected bounds touch none of this rank's screen tiles is another rank's
// work (cluster_touches_rank).  The lists of a frame never come here -- the group cull writes the rank's share itself and the
// occlusion culls start from it; this pass is for lists of unknown origin handed to the stand-alone entry points.  Same
// block-aggregated compaction as hzb_cull_kernel; order is free (cmd.z is carried).
struct RankFilterParams {
    const DObjFrame* objFrame; const DMeshlet* meshlets;
    const uint32_t* inCount; const ChordDrawCmd* inCmds;
    uint32_t* outCount; ChordDrawCmd* outCmds;
    ShardInfo shard; float W, H; int32_t Wi, Hi;
};

__global__ __launch_bounds__(256) void rank_filter_kernel(RankFilterParams p)
{
    __shared__ uint32_t sWave[4], sBase;
    const uint32_t count = *p.inCount;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * 1024u; base < count; base += gridDim.x * 1024u) {
        ChordDrawCmd cmd[4];
        uint32_t keep = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t i = base + k * 256u + threadIdx.x;
            cmd[k] = ChordDrawCmd{0, 0, 0};
            if (i < count) {
                cmd[k] = p.inCmds[i];
                const bool mine = cluster_touches_rank(p.shard, p.objFrame[cmd[k].objectId].mvp, p.meshlets[cmd[k].meshletId], p.W, p.H, p.Wi, p.Hi);
                if (mine) keep |= 1u << k;
            }
        }
        const uint32_t mineCount = (uint32_t)__popc(keep);
        uint32_t incl = mineCount;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t nb = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= (uint32_t)d) incl += nb; }
        if (lane == 63u) sWave[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4u; w++) { const uint32_t c = sWave[w]; if (w < wave) before += c; all += c; }
        if (threadIdx.x == 0) sBase = all ? atomicAdd(p.outCount, all) : 0u;
        __syncthreads();
        uint32_t slot = sBase + before + incl - mineCount;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) if (keep & (1u << k)) p.outCmds[slot++] = cmd[k];
        __syncthreads();
    }
}

void launch_rank_filter(ChordCtx* c, const CmdList& in, const CmdList& out)
{
    RankFilterParams p;
    p.objFrame = c->dObjFrame; p.meshlets = c->dMeshlets;
    p.inCount = in.count; p.inCmds = in.cmds; p.outCount = out.count; p.outCmds = out.cmds;
    p.shard = c->shard; p.W = (float)c->width; p.H = (float)c->height; p.Wi = (int32_t)c->width; p.Hi = (int32_t)c->height;
    uint32_t blocks = (in.capacity + 1023u) / 1024u;
    const uint32_t maxBlocks = (uint32_t)c->numCUs * 8u;
    if (blocks > maxBlocks) blocks = maxBlocks;
    if (blocks < 1) blocks = 1;
    CHORD_LAUNCH(c, rank_filter_kernel, dim3(blocks), dim3(256), 0, c->stream, p);
}

void launch_hzb_cull(ChordCtx* c, const HzbBuffers& hzb, int phase, const CmdList& in, const CmdList& outVisible,
                     const CmdList* outRejected)
{
    const HzbCullParams p = make_hzb_cull_params(c, hzb, in, outVisible, outRejected);
    // long lists: 1 024-thread workgroups, one command per thread, one reservation per workgroup and list
#ifndef HZB_CULL_OCT_LONG
#define HZB_CULL_OCT_LONG 0          // measurement builds: the eight-lane form also for long lists
#endif
    const bool longList = in.capacity > 65536u && !HZB_CULL_OCT_LONG;
    const uint32_t threads = longList ? 1024u : 256u;
    uint32_t blocks = (in.capacity + threads - 1u) / threads;
    const uint32_t maxBlocks = (uint32_t)c->numCUs * (longList ? 2u : 8u);
    if (blocks > maxBlocks) blocks = maxBlocks;
    if (blocks < 1) blocks = 1;
    // short lists: eight lanes per command, 32 commands per 256-thread workgroup (HZB_CULL_OCT) -- one wave per SIMD of a CU: a list of
    // a few thousand commands is a few dozen workgroups, and a wave that shares its SIMD with three others of the same chain issues
    // every 6.6 cycles instead of every 4.6 (tools/microbench/valu_issue)
#ifndef HZB_OCT_THREADS
#define HZB_OCT_THREADS 256u
#endif
    constexpr uint32_t OT = HZB_OCT_THREADS;
    const uint32_t octBlocks = std::max(1u, std::min((in.capacity + OT / 8u - 1u) / (OT / 8u), (uint32_t)c->numCUs * 8u * (1024u / OT)));
    if (phase == 0) { if (longList)          CHORD_LAUNCH(c, (hzb_cull_kernel<0, 1u, false, 1024u>), dim3(blocks), dim3(1024), 0, c->stream, p);
                      else if (HZB_CULL_OCT) CHORD_LAUNCH(c, (hzb_cull_kernel<0, 1u, false, OT, true>), dim3(octBlocks), dim3(OT), 0, c->stream, p);
                      else                   CHORD_LAUNCH(c, (hzb_cull_kernel<0, 1u, false>), dim3(blocks), dim3(256), 0, c->stream, p); }
    else if (c->hzbTailInCull) {
        // (inside chordvis_render_frame: the tile kernel wrote levels 0..5 of this chain; no hzb_tail_kernel ran)
        if (longList)          CHORD_LAUNCH(c, (hzb_cull_kernel<1, 1u, true, 1024u>), dim3(blocks), dim3(1024), 0, c->stream, p);
        else if (HZB_CULL_OCT) CHORD_LAUNCH(c, (hzb_cull_kernel<1, 1u, true, OT, true>), dim3(octBlocks), dim3(OT), 0, c->stream, p);
        else                   CHORD_LAUNCH(c, (hzb_cull_kernel<1, 1u, true>), dim3(blocks), dim3(256), 0, c->stream, p);
    } else          { if (longList)          CHORD_LAUNCH(c, (hzb_cull_kernel<1, 1u, false, 1024u>), dim3(blocks), dim3(1024), 0, c->stream, p);
                      else if (HZB_CULL_OCT) CHORD_LAUNCH(c, (hzb_cull_kernel<1, 1u, false, OT, true>), dim3(octBlocks), dim3(OT), 0, c->stream, p);
                      else                   CHORD_LAUNCH(c, (hzb_cull_kernel<1, 1u, false>), dim3(blocks), dim3(256), 0, c->stream, p); }
}

} // namespace chord
