/*
 * oracle.c — CPU restatement of chord's visibility hot path.  TEST INFRASTRUCTURE
 * (see oracle.h).  PARITY UNPINNED by the reference's own tests; pinned by the
 * analytic known-answer tests in tests/.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -msse2 -mfpmath=sse
 *
 * Canonical arithmetic.  The reference leaves these to the
 * shader compiler / driver / fixed-function rasterizer; this file fixes them:
 *  (1) IEEE-754 binary32, round-to-nearest-even, every + - * / sqrt rounded
 *      separately (no FMA contraction), evaluated in HLSL source order.
 *      mul(M, v)[r]   = ((M[r][0]*v0 + M[r][1]*v1) + M[r][2]*v2) + M[r][3]*v3
 *      mul(A, B)[r][c]= ((A[r][0]*B[0][c] + A[r][1]*B[1][c]) + A[r][2]*B[2][c]) + A[r][3]*B[3][c]
 *      dot(a, b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
 *      normalize(v)   = v / sqrt(dot(v, v))
 *      determinant(float3x3(r0,r1,r2)) = (r0.x*(r1.y*r2.z - r1.z*r2.y)
 *                     - r0.y*(r1.x*r2.z - r1.z*r2.x)) + r0.z*(r1.x*r2.y - r1.y*r2.x)
 *  (2) HLSL round() = round-half-to-even (rintf under the default FP env).
 *  (3) int(float) truncates toward zero; firstbithigh(0) = -1.
 *  (4) HZB texels are binary16, converted round-to-nearest-even.
 *  (5) Raster: vertex (uv * dimension) snapped to 1/256 pixel with rintf, pixel
 *      centres at +0.5, top-left fill rule, edge functions in 64-bit integers,
 *      depth(px) = (d0 + l1*(d1-d0)) + l2*(d2-d0) with l_i = float(E_i) * (1.0f/float(2A)),
 *      float(int64) := (float)(double)v (exact below 2^53, then one rounding),
 *      d_i = clip.z_i / clip.w_i.  Triangles with a vertex outside
 *      {0 <= z <= w, |x| <= G*w, |y| <= G*w} (G = 1024) go through a
 *      Sutherland-Hodgman clipper in clip space (fixed plane order, intersections
 *      always computed from the inside endpoint) and are fan-triangulated.
 *  (6) Packed pixel = (asuint(depth) << 32) | encodeTriangleIdInstanceId(tri, slot); clear 0.
 *  (7) Cluster slots are assigned in (objectId, groupIdx, meshlet-in-group) order.
 *  (8) Depth test GREATER_OR_EQUAL + draw order  ==>  64-bit max of the packed word.
 *  (9) Masked materials (alphaMode 1; mesh_raster.hlsl:34-38,107-112,198-204).  The reference's
 *      baseColorTexture.Sample() takes its level of detail from screen-space derivatives and filters as the
 *      sampler (8x anisotropic, asset_gltf.cpp:394-481) and the hardware see fit; fixed here:
 *      - vertex attributes u/w, v/w, 1/w (fp32 divisions by the clip w); at a pixel, with the depth's l1, l2:
 *        l0 = (1 - l1) - l2, den = (l0*iw0 + l1*iw1) + l2*iw2, u = ((l0*uw0 + l1*uw1) + l2*uw2) / den (v alike);
 *      - ONE level per (clipped piece of a) triangle: ratio = Auv / Apx, Auv = |(u1-u0)(v2-v0) - (u2-u0)(v1-v0)| *
 *        (float(W0) * float(H0)) (doubled uv area in level-0 texels), Apx = float(|2A|) * (1/65536) (doubled pixel
 *        area); ratio >= 1: level = min(mips - 1, floor(log2(ratio)) >> 1) (exponent bits), filter = minFilter;
 *        otherwise level 0, filter = magFilter.  No anisotropy, no blend between levels;
 *      - NEAREST: texel (floor(u*W), floor(v*H)); LINEAR: x = u*W - 0.5, x0 = floor(x), f = x - x0, the four texels
 *        around, a = lerp(lerp(a00,a10,fx), lerp(a01,a11,fx), fy), lerp(a,b,t) = a + (b-a)*t, texel = byte * (1/255);
 *        wrap per axis REPEAT / CLAMP_TO_EDGE / MIRRORED_REPEAT on the integer texel index;
 *      - clip(a * baseColorFactor.w - alphaCutOff): the fragment is dropped iff that is < 0.
 *      Blended materials (alphaMode 2) are in no bucket of renderMesh (mesh_raster.cpp:224) and draw nothing.
 * (10) Depth-only passes (PASS_TYPE_DEPTH, renderMeshDepth mesh_raster.cpp:159-206: shadow views).  Cull mode NONE for every
 *      bucket, no id output: the word is asuint(depth) << 32.  Depth clamp (VkPipelineRasterizationStateCreateInfo::
 *      depthClampEnable): no clipping against the near / far planes -- a vertex is "fast" iff w > 0 and inside the guard
 *      band, the clipper skips planes 0 and 1 -- and the interpolated depth is clamped to [0, 1].  Depth bias
 *      (vkCmdSetDepthBias(const, 0, slope)): o = slope * m + const * r with m = max(|dz/dx|, |dz/dy|) of the snapped
 *      triangle's depth plane (dz/dx = (float(256 a1) * invA) * e1 + (float(256 a2) * invA) * e2, dz/dy with b) and
 *      r = 2^(exponent(max |d_i|) - 23); fixed here as: the three VERTEX depths are biased by o before interpolation.
 */
#define _DEFAULT_SOURCE        /* mmap flags (the all-cores replay) under -std=c11 */
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#define MAT(mat, r, c) ((mat)->m[(c) * 4 + (r)])

typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---------------------------------------------------------------- f16 ---- */

uint16_t orc_f32_to_f16(float f)
{
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007FFFFFu;
    int32_t  exp  = (int32_t)((x >> 23) & 0xFF);
    if (exp == 0xFF) return (uint16_t)(sign | 0x7C00u | (mant ? (0x0200u | (mant >> 13)) : 0u));
    int32_t e = exp - 127 + 15;
    if (e >= 0x1F) return (uint16_t)(sign | 0x7C00u);                 /* overflow -> inf */
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;                              /* underflow -> 0 */
        mant |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - e);                            /* 14..24 */
        uint32_t h = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;             /* may carry into exp / inf: correct */
    return (uint16_t)(sign | h);
}

float orc_f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t mant = h & 0x3FFu;
    if (exp == 0) {
        if (mant == 0) return u2f(sign);
        int e = -1;
        do { e++; mant <<= 1; } while (!(mant & 0x400u));
        mant &= 0x3FFu;
        return u2f(sign | ((uint32_t)(127 - 15 - e) << 23) | (mant << 13));
    }
    if (exp == 0x1F) return u2f(sign | 0x7F800000u | (mant << 13));
    return u2f(sign | ((exp + 127 - 15) << 23) | (mant << 13));
}

/* --------------------------------------------------------- canonical ops -- */

static inline f4 mul_mv(const ChordMat4* M, float v0, float v1, float v2, float v3)
{
    f4 r;
    r.x = ((MAT(M,0,0) * v0 + MAT(M,0,1) * v1) + MAT(M,0,2) * v2) + MAT(M,0,3) * v3;
    r.y = ((MAT(M,1,0) * v0 + MAT(M,1,1) * v1) + MAT(M,1,2) * v2) + MAT(M,1,3) * v3;
    r.z = ((MAT(M,2,0) * v0 + MAT(M,2,1) * v1) + MAT(M,2,2) * v2) + MAT(M,2,3) * v3;
    r.w = ((MAT(M,3,0) * v0 + MAT(M,3,1) * v1) + MAT(M,3,2) * v2) + MAT(M,3,3) * v3;
    return r;
}

static inline void mul_mm(const ChordMat4* A, const ChordMat4* B, ChordMat4* C)
{
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++)
            MAT(C, r, c) = ((MAT(A,r,0) * MAT(B,0,c) + MAT(A,r,1) * MAT(B,1,c)) + MAT(A,r,2) * MAT(B,2,c)) + MAT(A,r,3) * MAT(B,3,c);
}

static inline float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

/* base.hlsli:184-194 */
static const float kExtentApplyFactor[8][3] = {
    { 1,  1,  1}, {-1, -1, -1}, { 1,  1, -1}, { 1, -1,  1},
    {-1,  1,  1}, { 1, -1, -1}, {-1, -1,  1}, {-1,  1, -1},
};

static inline f3 extent_corner(f3 c, f3 e, int k)
{
    f3 p;
    p.x = c.x + e.x * kExtentApplyFactor[k][0];
    p.y = c.y + e.y * kExtentApplyFactor[k][1];
    p.z = c.z + e.z * kExtentApplyFactor[k][2];
    return p;
}

/* base.hlsli:161-171 */
static inline f3 project_pos_to_uvz(f3 pos, const ChordMat4* proj)
{
    f4 h = mul_mv(proj, pos.x, pos.y, pos.z, 1.0f);
    f3 r;
    r.x = h.x / h.w; r.y = h.y / h.w; r.z = h.z / h.w;
    r.x = r.x * 0.5f + 0.5f;
    r.y = r.y * -0.5f + 0.5f;
    return r;
}

/* base.hlsli:243-246 */
static inline int is_ortho_projection(const ChordMat4* M) { return MAT(M,3,3) == 1.0f; }

/* base.hlsli:251-272 */
static int ortho_frustum_culling(f3 c, f3 e, const ChordMat4* localToClip)
{
    f3 mn = {10.0f, 10.0f, 10.0f}, mx = {-10.0f, -10.0f, -10.0f};
    for (int k = 0; k < 8; k++) {
        f3 uvz = project_pos_to_uvz(extent_corner(c, e, k), localToClip);
        mn.x = fminf(mn.x, uvz.x); mn.y = fminf(mn.y, uvz.y); mn.z = fminf(mn.z, uvz.z);
        mx.x = fmaxf(mx.x, uvz.x); mx.y = fmaxf(mx.y, uvz.y); mx.z = fmaxf(mx.z, uvz.z);
    }
    return (mn.x >= 1.0f || mn.y >= 1.0f) || (mx.x <= 0.0f || mx.y <= 0.0f);
}

/* base.hlsli:275-305 */
static int frustum_culling(const float planes[6][4], f3 c, f3 e, const ChordMat4* localToTranslatedWorld)
{
    f3 p[8];
    for (int k = 0; k < 8; k++) {
        f3 q = extent_corner(c, e, k);
        f4 h = mul_mv(localToTranslatedWorld, q.x, q.y, q.z, 1.0f);
        p[k].x = h.x; p[k].y = h.y; p[k].z = h.z;
    }
    for (int i = 0; i < 6; i++) {
        f3 n = {planes[i][0], planes[i][1], planes[i][2]};
        int allBack = 1;
        for (int j = 0; j < 8; j++) {
            if (dot3(n, p[j]) > -planes[i][3]) { allBack = 0; break; }
        }
        if (allBack) return 1;
    }
    return 0;
}

static inline void aabb_center_extent(const float mn[3], const float mx[3], int meshletOrder, f3* c, f3* e)
{
    /* instance_culling.hlsl:73-76 writes (posMin + posMax) * 0.5, nanite_shared.hlsli:74-77
     * and hzb_mainview_culling.hlsl:65-68 write 0.5 * (...) / (...) * 0.5: same value. */
    (void)meshletOrder;
    c->x = (mn[0] + mx[0]) * 0.5f; c->y = (mn[1] + mx[1]) * 0.5f; c->z = (mn[2] + mx[2]) * 0.5f;
    e->x = mx[0] - c->x; e->y = mx[1] - c->y; e->z = mx[2] - c->z;
}

/* ------------------------------------------------------------ object cull -- */

static int object_visible(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv, uint32_t flags, uint32_t o)
{
    /* instance_culling.hlsl:66-90 */
    const ChordObject* obj = &scene->objects[o];
    const ChordPrimitive* prim = &scene->primitives[obj->GLTFPrimitiveDetail];
    if (!(flags & CHORD_FLAG_FRUSTUM_CULL)) return 1;
    ChordMat4 localToClip;
    mul_mm(&iv->translatedWorldToClip, &obj->basicData.localToTranslatedWorld, &localToClip);
    f3 c, e;
    aabb_center_extent(prim->posMin, prim->posMax, 0, &c, &e);
    if (is_ortho_projection(&localToClip)) return !ortho_frustum_culling(c, e, &localToClip);
    return !frustum_culling(iv->frustumPlanesRS, c, e, &obj->basicData.localToTranslatedWorld);
}

void orc_object_cull(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv, uint32_t flags, uint8_t* visible)
{
    for (uint32_t o = 0; o < scene->objectCount; o++) visible[o] = (uint8_t)object_visible(scene, iv, flags, o);
}

/* -------------------------------------------------------- group LOD + cull -- */

/* base.hlsli:233-241 + :503-518; K = (h * 0.5) / tan(fovy / 2) is host-precomputed (view->lodScale). */
static float projected_error_px(const ChordCameraView* view, const ChordMat4* localToView, float maxScaleAbs,
                                const float center[3], float radius)
{
    f4 q = mul_mv(localToView, center[0], center[1], center[2], 1.0f);
    float R = maxScaleAbs * radius;
    f3 q3 = {q.x, q.y, q.z};
    float d2 = dot3(q3, q3);
    float r2 = R * R;
    if (d2 <= r2) return -1.0f;
    return view->lodScale * R / sqrtf(d2 - r2);
}

int orc_group_visible(const ChordCameraView* view, const ChordObject* obj, const ChordMeshletGroup* g)
{
    /* nanite_shared.hlsli:15-49; localToView from instance_culling.hlsl:170 (always the main view) */
    ChordMat4 localToView;
    mul_mm(&view->translatedWorldToView, &obj->basicData.localToTranslatedWorld, &localToView);
    float s = obj->basicData.scaleExtractFromMatrix[3];
    int finalLod = g->parentError > CHORD_ERROR_RADIUS_ROOT;
    int firstLod = g->error < -0.5f;
    if (!finalLod) {
        float pe = projected_error_px(view, &localToView, s, g->parentPosCenter, g->parentError);
        if (pe > 0.0f && pe <= CHORD_ERROR_PIXEL_THRESHOLD) return 0;
    }
    if (!firstLod) {
        float er = projected_error_px(view, &localToView, s, g->clusterPosCenter, g->error);
        if (er < 0.0f || er > CHORD_ERROR_PIXEL_THRESHOLD) return 0;
    }
    return 1;
}

int orc_meshlet_visible(uint32_t flags, const ChordInstanceCullingView* iv, const ChordObject* obj,
                        const ChordMeshlet* m, const ChordMaterial* mat)
{
    /* nanite_shared.hlsli:51-91 */
    if (mat->bTwoSided == 0 && (flags & CHORD_FLAG_CONE_CULL)) {
        f4 cam = mul_mv(&obj->basicData.translatedWorldToLocal, 0.0f, 0.0f, 0.0f, 1.0f);
        f3 v = {m->coneApex[0] - cam.x, m->coneApex[1] - cam.y, m->coneApex[2] - cam.z};
        float len = sqrtf(dot3(v, v));
        f3 n = {v.x / len, v.y / len, v.z / len};
        f3 axis = {m->coneAxis[0], m->coneAxis[1], m->coneAxis[2]};
        if (dot3(n, axis) >= m->coneCutOff) return 0;
    }
    if (flags & CHORD_FLAG_FRUSTUM_CULL) {
        f3 c, e;
        aabb_center_extent(m->posMin, m->posMax, 1, &c, &e);
        ChordMat4 localToClip;
        mul_mm(&iv->translatedWorldToClip, &obj->basicData.localToTranslatedWorld, &localToClip);
        if (is_ortho_projection(&localToClip)) return !ortho_frustum_culling(c, e, &localToClip);
        return !frustum_culling(iv->frustumPlanesRS, c, e, &obj->basicData.localToTranslatedWorld);
    }
    return 1;
}

/* objects [o0, o1): commands appended at outCmds[n...] (those past `cap` are counted, not stored), slots numbered from n */
static uint32_t instance_culling_range(const ChordSceneDesc* scene, const ChordCameraView* view,
                                       const ChordInstanceCullingView* iv, uint32_t flags, uint32_t o0, uint32_t o1,
                                       ChordDrawCmd* outCmds, uint32_t cap, uint32_t n)
{
    for (uint32_t o = o0; o < o1; o++) {
        if (!object_visible(scene, iv, flags, o)) continue;
        const ChordObject* obj = &scene->objects[o];
        const ChordPrimitive* prim = &scene->primitives[obj->GLTFPrimitiveDetail];
        const ChordMaterial* mat = &scene->materials[obj->GLTFMaterialData];
        const ChordAssetDesc* as = &scene->assets[prim->primitiveDatasBufferId];
        for (uint32_t gi = 0; gi < prim->meshletGroupCount; gi++) {
            /* instance_culling.hlsl:163-164 */
            const ChordMeshletGroup* g = &as->meshletGroups[prim->meshletGroupOffset + gi];
            if (!orc_group_visible(view, obj, g)) continue;
            for (uint32_t i = 0; i < g->meshletCount; i++) {
                /* instance_culling.hlsl:178-180 */
                uint32_t loadId = i + g->meshletOffset + prim->meshletGroupIndicesOffset;
                uint32_t meshletIndex = prim->meshletOffset + as->meshletGroupIndices[loadId];
                if (orc_meshlet_visible(flags, iv, obj, &as->meshlets[meshletIndex], mat)) {
                    if (n < cap) { outCmds[n].objectId = o; outCmds[n].meshletId = meshletIndex; outCmds[n].slot = n; }
                    n++;
                }
            }
        }
    }
    return n;
}

uint32_t orc_instance_culling(const ChordSceneDesc* scene, const ChordCameraView* view,
                              const ChordInstanceCullingView* iv, uint32_t flags,
                              ChordDrawCmd* outCmds, uint32_t cap)
{
    return instance_culling_range(scene, view, iv, flags, 0, scene->objectCount, outCmds, cap, 0);
}

/* -------------------------------------------------------------------- HZB -- */

static uint32_t next_pot(uint32_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++; return v; }

void orc_hzb_desc(uint32_t srcW, uint32_t srcH, ChordHZBDesc* out)
{
    /* hzb.cpp:49-63 */
    uint32_t w = next_pot(srcW) / 2, h = next_pot(srcH) / 2;
    if (w == srcW) w /= 2;
    if (h == srcH) h /= 2;
    if (w < 1) w = 1;
    if (h < 1) h = 1;
    uint32_t mx = w > h ? w : h, mips = 0;
    while (mx) { mips++; mx >>= 1; }                 /* floor(log2(max)) + 1 */
    if (mips > CHORD_HZB_MAX_MIPS) mips = CHORD_HZB_MAX_MIPS;
    memset(out, 0, sizeof(*out));
    out->srcWidth = srcW; out->srcHeight = srcH; out->width = w; out->height = h; out->mipCount = mips;
    uint32_t off = 0;
    for (uint32_t l = 0; l < mips; l++) {
        uint32_t mw = w >> l, mh = h >> l;
        if (mw < 1) mw = 1;
        if (mh < 1) mh = 1;
        out->mipOffset[l] = off;
        off += mw * mh;
    }
    out->totalTexels = off;
}

static inline uint32_t mip_w(const ChordHZBDesc* d, uint32_t l) { uint32_t w = d->width >> l; return w ? w : 1; }
static inline uint32_t mip_h(const ChordHZBDesc* d, uint32_t l) { uint32_t h = d->height >> l; return h ? h : 1; }

uint32_t orc_hzb_valid_w(const ChordHZBDesc* d, uint32_t l)
{
    uint32_t v = (((d->srcWidth - 1) >> 1) >> l) + 1, w = mip_w(d, l);
    return v < w ? v : w;
}
uint32_t orc_hzb_valid_h(const ChordHZBDesc* d, uint32_t l)
{
    uint32_t v = (((d->srcHeight - 1) >> 1) >> l) + 1, h = mip_h(d, l);
    return v < h ? v : h;
}

/* hzb_one.hlsl:126-372 / hzb.hlsl:127-389 restated per SURVEY Appendix A4:
 * texel (x,y) of mip l = reduce over the 2^(l+1) square of edge-clamped
 * source depth; converted to binary16 on store.  min/max commute with the
 * monotone f32->f16 rounding, so reducing level by level from the stored
 * halves (what the shader does from mip 5 up) gives the same bits.  Max
 * variant: mips >= 5 carry +1 ulp (hzb.hlsl:67-71).  Only the sampled
 * (valid) extent of each mip is defined; the rest is written as 0.
 * Rows [y0, y1) of a level (the all-cores replay hands each thread a range; texels do not depend on each other). */
static void hzb_mip0_rows(const uint64_t* vis, uint32_t W, uint32_t H, const ChordHZBDesc* desc,
                          uint16_t* hzbMin, uint16_t* hzbMax, uint32_t y0, uint32_t y1, uint32_t* vminIo, uint32_t* vmaxIo)
{
    uint32_t vmin = *vminIo, vmax = *vmaxIo;
    const uint32_t vw = orc_hzb_valid_w(desc, 0), mw = mip_w(desc, 0);
    for (uint32_t y = y0; y < y1; y++) for (uint32_t x = 0; x < vw; x++) {
        float mn = 0, mx = 0;
        for (int j = 0; j < 2; j++) for (int i = 0; i < 2; i++) {
            uint32_t sx = 2 * x + i, sy = 2 * y + j;
            if (sx > W - 1) sx = W - 1;
            if (sy > H - 1) sy = H - 1;
            float d = u2f((uint32_t)(vis[(size_t)sy * W + sx] >> 32));
            if (i == 0 && j == 0) { mn = d; mx = d; } else { mn = fminf(mn, d); mx = fmaxf(mx, d); }
            if (d > 0.0f) {                      /* hzb.hlsl:163-166 */
                uint32_t b = f2u(d);
                if (d < 1.0f && b < vmin) vmin = b;   /* :173 guard, see note in DESIGN.md */
                if (b > vmax) vmax = b;
            }
        }
        hzbMin[desc->mipOffset[0] + y * mw + x] = orc_f32_to_f16(mn);
        if (hzbMax) hzbMax[desc->mipOffset[0] + y * mw + x] = orc_f32_to_f16(mx);
    }
    *vminIo = vmin; *vmaxIo = vmax;
}

static void hzb_mip_rows(const ChordHZBDesc* desc, uint32_t l, uint16_t* hzbMin, uint16_t* hzbMax, uint32_t y0, uint32_t y1)
{
    const uint32_t vw = orc_hzb_valid_w(desc, l), mw = mip_w(desc, l);
    const uint32_t pw = orc_hzb_valid_w(desc, l - 1), ph = orc_hzb_valid_h(desc, l - 1), pmw = mip_w(desc, l - 1);
    for (uint32_t y = y0; y < y1; y++) for (uint32_t x = 0; x < vw; x++) {
        float mn = 0, mx = 0;
        for (int j = 0; j < 2; j++) for (int i = 0; i < 2; i++) {
            uint32_t cx = 2 * x + i, cy = 2 * y + j;
            if (cx > pw - 1) cx = pw - 1;        /* children past the valid edge are clamped duplicates */
            if (cy > ph - 1) cy = ph - 1;
            size_t idx = desc->mipOffset[l - 1] + (size_t)cy * pmw + cx;
            float a = orc_f16_to_f32(hzbMin[idx]);
            float b = hzbMax ? orc_f16_to_f32(hzbMax[idx]) : 0.0f;
            if (i == 0 && j == 0) { mn = a; mx = b; } else { mn = fminf(mn, a); mx = fmaxf(mx, b); }
        }
        size_t o = desc->mipOffset[l] + (size_t)y * mw + x;
        hzbMin[o] = orc_f32_to_f16(mn);
        if (hzbMax) {
            uint16_t h = orc_f32_to_f16(mx);
            if (l == 5) h = (uint16_t)(h + 1);   /* storeHZBMip5: f32tof16(depth) + 1 */
            hzbMax[o] = h;
        }
    }
}

void orc_hzb_build(const uint64_t* vis, uint32_t W, uint32_t H, const ChordHZBDesc* desc,
                   uint16_t* hzbMin, uint16_t* hzbMax, uint32_t validRange[2])
{
    memset(hzbMin, 0, sizeof(uint16_t) * desc->totalTexels);
    if (hzbMax) memset(hzbMax, 0, sizeof(uint16_t) * desc->totalTexels);
    uint32_t vmin = 0xFFFFFFFFu, vmax = 0u;
    hzb_mip0_rows(vis, W, H, desc, hzbMin, hzbMax, 0, orc_hzb_valid_h(desc, 0), &vmin, &vmax);   /* mip 0 from the source */
    for (uint32_t l = 1; l < desc->mipCount; l++) hzb_mip_rows(desc, l, hzbMin, hzbMax, 0, orc_hzb_valid_h(desc, l));
    if (validRange) { validRange[0] = vmin; validRange[1] = vmax; }
}

static inline int first_bit_high(int32_t v)
{
    if (v <= 0) return -1;     /* arguments here are >= 0; firstbithigh(0) = -1 */
    int b = 31;
    while (!((uint32_t)v >> b)) b--;
    return b;
}

static inline float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

static int hzb_visible(const ChordSceneDesc* scene, const ChordCameraView* view, uint32_t flags, int phase,
                       const ChordHZBDesc* hzb, const uint16_t* hzbMin, const ChordDrawCmd* cmd)
{
    /* hzb_mainview_culling.hlsl:56-161 */
    const ChordObject* obj = &scene->objects[cmd->objectId];
    const ChordPrimitive* prim = &scene->primitives[obj->GLTFPrimitiveDetail];
    const ChordAssetDesc* as = &scene->assets[prim->primitiveDatasBufferId];
    const ChordMeshlet* m = &as->meshlets[cmd->meshletId];
    if (!(flags & CHORD_FLAG_HZB_CULL)) return 1;

    f3 c, e;
    aabb_center_extent(m->posMin, m->posMax, 1, &c, &e);
    ChordMat4 mvp;
    if (phase == 0) mul_mm(&view->translatedWorldToClipLastFrame, &obj->basicData.localToTranslatedWorldLastFrame, &mvp);
    else            mul_mm(&view->translatedWorldToClip, &obj->basicData.localToTranslatedWorld, &mvp);

    f3 mx = {-10.0f, -10.0f, -10.0f}, mn = {10.0f, 10.0f, 10.0f};
    for (int i = 0; i < 8; i++) {
        f3 uvz = project_pos_to_uvz(extent_corner(c, e, i), &mvp);
        mn.x = fminf(mn.x, uvz.x); mn.y = fminf(mn.y, uvz.y); mn.z = fminf(mn.z, uvz.z);
        mx.x = fmaxf(mx.x, uvz.x); mx.y = fmaxf(mx.y, uvz.y); mx.z = fmaxf(mx.z, uvz.z);
    }
    int zInRange = mx.z < 1.0f && mn.z > 0.0f;
    int visible = 1;
    if (zInRange) {
        if ((mn.x >= 1.0f || mn.y >= 1.0f) || (mx.x <= 0.0f || mx.y <= 0.0f)) visible = 0;
    }
    if (visible && zInRange) {
        mn.x = saturatef(mn.x); mn.y = saturatef(mn.y);
        mx.x = saturatef(mx.x); mx.y = saturatef(mx.y);
        const float W = view->renderDimension[0], H = view->renderDimension[1];
        int32_t rx = (int32_t)(mn.x * W + 0.5f);
        int32_t ry = (int32_t)(mn.y * H + 0.5f);
        int32_t rz = (int32_t)(mx.x * W + -0.5f);
        int32_t rw = (int32_t)(mx.y * H + -0.5f);
        if (rx < 0) rx = 0;
        if (ry < 0) ry = 0;
        /* min(renderDimension.xy - 1, pixelRect.zw) in float, then truncated */
        rz = (int32_t)fminf(W - 1.0f, (float)rz);
        rw = (int32_t)fminf(H - 1.0f, (float)rw);
        if (rz < rx || rw < ry) {
            visible = 0;
        } else {
            int32_t mx0 = rx >> 1, my0 = ry >> 1, mz0 = rz >> 1, mw0 = rw >> 1;
            int lx = first_bit_high(mz0 - mx0), ly = first_bit_high(mw0 - my0);
            int lv = (lx > ly ? lx : ly) - 1;
            if (lv < 0) lv = 0;
            if (((mz0 >> lv) - (mx0 >> lv) >= 4) || ((mw0 >> lv) - (my0 >> lv) >= 4)) lv += 1;
            int32_t cx = mx0 >> lv, cy = my0 >> lv, cz = mz0 >> lv, cw = mw0 >> lv;
            float zMin = 10.0f;
            uint32_t mw = mip_w(hzb, (uint32_t)lv);
            for (int x = 0; x < 4; x++) for (int y = 0; y < 4; y++) {
                int32_t sx = cx + x < cz ? cx + x : cz;
                int32_t sy = cy + y < cw ? cy + y : cw;
                zMin = fminf(zMin, orc_f16_to_f32(hzbMin[hzb->mipOffset[lv] + (size_t)sy * mw + (size_t)sx]));
            }
            if (zMin > mx.z) visible = 0;
        }
    }
    return visible;
}

void orc_hzb_culling(const ChordSceneDesc* scene, const ChordCameraView* view, uint32_t flags, int phase,
                     const ChordHZBDesc* hzb, const uint16_t* hzbMin,
                     const ChordDrawCmd* inCmds, uint32_t inCount,
                     ChordDrawCmd* outVisible, uint32_t* outVisibleCount,
                     ChordDrawCmd* outRejected, uint32_t* outRejectedCount)
{
    uint32_t nv = 0, nr = 0;
    for (uint32_t i = 0; i < inCount; i++) {
        if (hzb_visible(scene, view, flags, phase, hzb, hzbMin, &inCmds[i])) outVisible[nv++] = inCmds[i];
        else if (phase == 0 && outRejected) outRejected[nr++] = inCmds[i];
    }
    *outVisibleCount = nv;
    if (outRejectedCount) *outRejectedCount = nr;
}

/* ----------------------------------------------------------------- raster -- */

static inline int owns_pixel(const OrcShard* s, uint32_t x, uint32_t y)
{
    if (!s || s->ranks <= 1 || !s->owners) return 1;
    return s->owners[(size_t)(y / 64u) * s->tilesX + x / 64u] == s->rank;
}

/* `atomic` names how a raster call reaches the image: 0 plain row-major, 1 row-major shared by threads (compare-and-swap),
 * 2 a thread's PRIVATE image stored tile by tile (64 x 64 words each, so that only the pages of touched tiles are ever
 * mapped) with the touched tiles recorded in t_dirty -- the all-cores replay merges those by max (orc_frame_mt). */
#define ORC_PRIV_TILE 64u
static __thread uint8_t* t_dirty = NULL;

static inline void vis_max(uint64_t* p, uint64_t v, int atomic)
{
    if (atomic != 1) { if (v > *p) *p = v; return; }
    uint64_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > cur) {
        if (__atomic_compare_exchange_n(p, &cur, v, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
    }
}

static inline int64_t floor_shift8(int64_t v) { return v >= 0 ? (v >> 8) : -((-v + 255) >> 8); }

/* ---- masked materials: canonical texture fetch (header item 9) ---- */
typedef struct {
    float uw[3], vw[3], iw[3];
    const ChordTexture* tex;        /* NULL: white fallback (alpha 1) */
    ChordSampler smp;
    uint32_t level;
    int linear;
    float alphaFactor, alphaCutOff;
} MaskCtx;

static inline int wrap_index(int64_t i, int64_t n, uint32_t mode)
{
    if (mode == CHORD_WRAP_CLAMP_TO_EDGE) return (int)(i < 0 ? 0 : (i > n - 1 ? n - 1 : i));
    if (mode == CHORD_WRAP_MIRRORED_REPEAT) {
        int64_t m = i % (2 * n);
        if (m < 0) m += 2 * n;
        return (int)(m < n ? m : 2 * n - 1 - m);
    }
    int64_t m = i % n;                                   /* REPEAT (and anything unknown) */
    if (m < 0) m += n;
    return (int)m;
}

static inline int64_t texel_floor(float x)
{
    if (!(fabsf(x) < 1.0e9f)) return 0;                  /* NaN / out of any texture: texel 0 */
    return (int64_t)floorf(x);
}

float orc_sample_alpha(const ChordTexture* tex, const ChordSampler* smp, uint32_t level, int linear, float u, float v)
{
    if (!tex || !tex->rgba8) return 1.0f;
    if (level >= tex->mipCount) level = tex->mipCount - 1u;
    size_t off = 0;
    for (uint32_t l = 0; l < level; l++) {
        uint32_t w = tex->width >> l, h = tex->height >> l;
        off += (size_t)(w ? w : 1u) * (h ? h : 1u);
    }
    const int64_t W = (tex->width >> level) ? (tex->width >> level) : 1, H = (tex->height >> level) ? (tex->height >> level) : 1;
    const uint8_t* base = tex->rgba8 + off * 4u;
#define TEXEL_A(ix, iy) ((float)base[((size_t)(iy) * (size_t)W + (size_t)(ix)) * 4u + 3u] * (1.0f / 255.0f))
    if (!linear) {
        const int ix = wrap_index(texel_floor(u * (float)W), W, smp->wrapS), iy = wrap_index(texel_floor(v * (float)H), H, smp->wrapT);
        return TEXEL_A(ix, iy);
    }
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const int64_t x0 = texel_floor(x), y0 = texel_floor(y);
    float fx = x - (float)x0, fy = y - (float)y0;
    if (!(fabsf(x) < 1.0e9f)) fx = 0.0f;
    if (!(fabsf(y) < 1.0e9f)) fy = 0.0f;
    const int ix0 = wrap_index(x0, W, smp->wrapS), ix1 = wrap_index(x0 + 1, W, smp->wrapS);
    const int iy0 = wrap_index(y0, H, smp->wrapT), iy1 = wrap_index(y0 + 1, H, smp->wrapT);
    const float a00 = TEXEL_A(ix0, iy0), a10 = TEXEL_A(ix1, iy0), a01 = TEXEL_A(ix0, iy1), a11 = TEXEL_A(ix1, iy1);
#undef TEXEL_A
    const float top = a00 + (a10 - a00) * fx, bot = a01 + (a11 - a01) * fx;
    return top + (bot - top) * fy;
}

static inline int filter_is_linear(uint32_t f)
{
    return f == CHORD_FILTER_LINEAR || f == CHORD_FILTER_LINEAR_MIPMAP_NEAREST || f == CHORD_FILTER_LINEAR_MIPMAP_LINEAR;
}

/* per (piece of a) triangle: attributes over w, the level and the filter */
static void mask_setup(MaskCtx* mk, const ChordSceneDesc* scene, const ChordMaterial* mat, int64_t absArea2,
                       const float u[3], const float v[3], const float w[3])
{
    for (int i = 0; i < 3; i++) { mk->iw[i] = 1.0f / w[i]; mk->uw[i] = u[i] * mk->iw[i]; mk->vw[i] = v[i] * mk->iw[i]; }
    mk->tex = (scene->textures && mat->baseColorId < scene->textureCount) ? &scene->textures[mat->baseColorId] : NULL;
    if (scene->samplers && mat->baseColorSampler < scene->samplerCount) mk->smp = scene->samplers[mat->baseColorSampler];
    else { mk->smp.minFilter = mk->smp.magFilter = CHORD_FILTER_NEAREST; mk->smp.wrapS = mk->smp.wrapT = CHORD_WRAP_REPEAT; }
    mk->alphaFactor = mat->baseColorFactor[3]; mk->alphaCutOff = mat->alphaCutOff;
    mk->level = 0; mk->linear = filter_is_linear(mk->smp.magFilter);
    if (mk->tex) {
        const float texels = (float)mk->tex->width * (float)mk->tex->height;
        const float auv = fabsf((u[1] - u[0]) * (v[2] - v[0]) - (u[2] - u[0]) * (v[1] - v[0])) * texels;
        const float apx = (float)(double)absArea2 * (1.0f / 65536.0f);
        const float ratio = auv / apx;
        if (ratio >= 1.0f) {
            const int32_t e = (int32_t)((f2u(ratio) >> 23) & 0xFFu) - 127;      /* floor(log2(ratio)), 128 for inf */
            uint32_t level = (uint32_t)(e >> 1);
            if (level > mk->tex->mipCount - 1u) level = mk->tex->mipCount - 1u;
            mk->level = level; mk->linear = filter_is_linear(mk->smp.minFilter);
        }
    }
}

uint32_t orc_mask_level(const ChordSceneDesc* scene, const ChordMaterial* mat, int64_t absArea2, const float u[3], const float v[3], int* linear)
{
    MaskCtx mk; const float w[3] = {1.0f, 1.0f, 1.0f};
    mask_setup(&mk, scene, mat, absArea2, u, v, w);
    if (linear) *linear = mk.linear;
    return mk.level;
}

typedef struct { int atomic; } RasterCtx;

static void raster_snapped(const int32_t X[3], const int32_t Y[3], const float d[3], int twoSided, uint32_t payload,
                           uint32_t W, uint32_t H, const OrcShard* shard, uint64_t* vis, OrcRasterStats* st, int atomic,
                           MaskCtx* mk, const ChordSceneDesc* scene, const ChordMaterial* mat, const float tu[3], const float tv[3], const float tw[3],
                           int depthClamp)
{
    /* signed doubled area in y-down screen space; reference front faces
     * (det > 0 in clip (x,y,w), mesh_raster.hlsl:143-149; CCW front + Y-flipped
     * viewport, helper.h:304-324,395-400) have area2 < 0 here. */
    int64_t area2 = (int64_t)(X[1] - X[0]) * (int64_t)(Y[2] - Y[0]) - (int64_t)(X[2] - X[0]) * (int64_t)(Y[1] - Y[0]);
    if (area2 == 0) return;
    if (!twoSided && area2 > 0) return;             /* VK_CULL_MODE_BACK_BIT, mesh_raster.cpp:235 */
    int64_t s = area2 < 0 ? -1 : 1;
    int64_t A = area2 * s;

    int32_t minX = X[0] < X[1] ? X[0] : X[1]; if (X[2] < minX) minX = X[2];
    int32_t maxX = X[0] > X[1] ? X[0] : X[1]; if (X[2] > maxX) maxX = X[2];
    int32_t minY = Y[0] < Y[1] ? Y[0] : Y[1]; if (Y[2] < minY) minY = Y[2];
    int32_t maxY = Y[0] > Y[1] ? Y[0] : Y[1]; if (Y[2] > maxY) maxY = Y[2];
    int64_t px0 = floor_shift8((int64_t)minX + 127), px1 = floor_shift8((int64_t)maxX - 128);
    int64_t py0 = floor_shift8((int64_t)minY + 127), py1 = floor_shift8((int64_t)maxY - 128);
    if (px0 < 0) px0 = 0;
    if (py0 < 0) py0 = 0;
    if (px1 > (int64_t)W - 1) px1 = (int64_t)W - 1;
    if (py1 > (int64_t)H - 1) py1 = (int64_t)H - 1;
    if (px1 < px0 || py1 < py0) return;
    if (st) st->trianglesRastered++;
    const uint32_t tilesX = (W + ORC_PRIV_TILE - 1u) / ORC_PRIV_TILE;
    if (atomic == 2 && t_dirty)
        for (int64_t ty = py0 / ORC_PRIV_TILE; ty <= py1 / ORC_PRIV_TILE; ty++)
            for (int64_t tx = px0 / ORC_PRIV_TILE; tx <= px1 / ORC_PRIV_TILE; tx++) t_dirty[ty * tilesX + tx] = 1;

    /* edge i is opposite vertex i: E0 = orient(V1,V2,P), E1 = orient(V2,V0,P), E2 = orient(V0,V1,P) */
    static const int ea[3] = {1, 2, 0}, eb[3] = {2, 0, 1};
    int64_t a[3], b[3], bias[3];
    for (int i = 0; i < 3; i++) {
        /* F(P) = s * ((B.x-A.x)*(P.y-A.y) - (B.y-A.y)*(P.x-A.x)); dF/dx = a, dF/dy = b */
        a[i] = -s * (int64_t)(Y[eb[i]] - Y[ea[i]]);
        b[i] =  s * (int64_t)(X[eb[i]] - X[ea[i]]);
        int topLeft = (a[i] > 0) || (a[i] == 0 && b[i] > 0);
        bias[i] = topLeft ? 0 : -1;
    }
    const float invA = 1.0f / (float)(double)A;
    const float e1 = d[1] - d[0], e2 = d[2] - d[0];
    if (mk) mask_setup(mk, scene, mat, A, tu, tv, tw);

    for (int64_t py = py0; py <= py1; py++) {
        for (int64_t px = px0; px <= px1; px++) {
            if (!owns_pixel(shard, (uint32_t)px, (uint32_t)py)) continue;
            int64_t cx = px * 256 + 128, cy = py * 256 + 128;
            int64_t E[3];
            int inside = 1;
            for (int i = 0; i < 3; i++) {
                E[i] = s * ((int64_t)(X[eb[i]] - X[ea[i]]) * (cy - Y[ea[i]]) - (int64_t)(Y[eb[i]] - Y[ea[i]]) * (cx - X[ea[i]]));
                if (E[i] + bias[i] < 0) { inside = 0; break; }
            }
            if (!inside) continue;
            float l1 = (float)(double)E[1] * invA, l2 = (float)(double)E[2] * invA;
            if (mk) {                                    /* mesh_raster.hlsl:198-204 */
                const float l0 = (1.0f - l1) - l2;
                const float den = (l0 * mk->iw[0] + l1 * mk->iw[1]) + l2 * mk->iw[2];
                const float tu_ = ((l0 * mk->uw[0] + l1 * mk->uw[1]) + l2 * mk->uw[2]) / den;
                const float tv_ = ((l0 * mk->vw[0] + l1 * mk->vw[1]) + l2 * mk->vw[2]) / den;
                const float a = orc_sample_alpha(mk->tex, &mk->smp, mk->level, mk->linear, tu_, tv_);
                if (a * mk->alphaFactor - mk->alphaCutOff < 0.0f) { if (st) st->fragmentsClipped++; continue; }
            }
            float z = (d[0] + l1 * e1) + l2 * e2;
            if (depthClamp) z = fminf(fmaxf(z, 0.0f), 1.0f);
            uint64_t packed = ((uint64_t)f2u(z) << 32) | payload;
            const size_t at = atomic == 2 ? ((((size_t)(py / ORC_PRIV_TILE) * tilesX + (size_t)(px / ORC_PRIV_TILE)) * ORC_PRIV_TILE + (size_t)(py % ORC_PRIV_TILE)) * ORC_PRIV_TILE + (size_t)(px % ORC_PRIV_TILE))
                                          : (size_t)py * W + (size_t)px;
            vis_max(&vis[at], packed, atomic);
            if (st) st->fragments++;
        }
    }
}

void orc_raster_snapped_triangle(const int32_t X[3], const int32_t Y[3], const float d[3],
                                 int twoSided, uint32_t payload, uint32_t W, uint32_t H,
                                 const OrcShard* shard, uint64_t* vis, OrcRasterStats* stats)
{
    raster_snapped(X, Y, d, twoSided, payload, W, H, shard, vis, stats, 0, NULL, NULL, NULL, NULL, NULL, NULL, 0);
}

#define ORC_GUARD 1024.0f

/* signed distance of clip-space vertex to plane k (inside >= 0) */
static inline float clip_dist(const f4* v, int k)
{
    switch (k) {
    case 0: return v->w - v->z;                 /* near (reverse-Z: depth <= 1) */
    case 1: return v->z;                        /* far  (depth >= 0)            */
    case 2: return ORC_GUARD * v->w + v->x;
    case 3: return ORC_GUARD * v->w - v->x;
    case 4: return ORC_GUARD * v->w + v->y;
    default: return ORC_GUARD * v->w - v->y;
    }
}

static inline int vertex_in_fast_volume(const f4* v)
{
    if (!(v->w > 0.0f)) return 0;
    for (int k = 0; k < 6; k++) if (!(clip_dist(v, k) >= 0.0f)) return 0;
    return 1;
}

/* depth clamp: the near / far planes do not clip */
static inline int vertex_fast(const f4* v, int depthClamp)
{
    if (!depthClamp) return vertex_in_fast_volume(v);
    if (!(v->w > 0.0f)) return 0;
    for (int k = 2; k < 6; k++) if (!(clip_dist(v, k) >= 0.0f)) return 0;
    return 1;
}

static inline f4 clip_intersect(const f4* in, const f4* out, float din, float dout)
{
    float t = din / (din - dout);
    f4 r;
    r.x = in->x + (out->x - in->x) * t;
    r.y = in->y + (out->y - in->y) * t;
    r.z = in->z + (out->z - in->z) * t;
    r.w = in->w + (out->w - in->w) * t;
    return r;
}

static inline void snap_vertex(const f4* h, float W, float H, int32_t* X, int32_t* Y, float* d)
{
    /* same expression the cull tests use (mesh_raster.hlsl:159-161) with w > 0 */
    float u = h->x / fabsf(h->w) * 0.5f + 0.5f;
    float v = h->y / fabsf(h->w) * -0.5f + 0.5f;
    *X = (int32_t)rintf((u * W) * 256.0f);
    *Y = (int32_t)rintf((v * H) * 256.0f);
    *d = h->z / h->w;
}

typedef struct { int depthOnly, depthClamp; float biasConst, biasSlope; } PassMode;
static const PassMode kClusterPass = {0, 0, 0.0f, 0.0f};

/* depth bias of a snapped triangle (header item 10) */
static float depth_bias(const PassMode* pm, const int32_t X[3], const int32_t Y[3], const float d[3])
{
    if (pm->biasConst == 0.0f && pm->biasSlope == 0.0f) return 0.0f;
    int64_t area2 = (int64_t)(X[1] - X[0]) * (int64_t)(Y[2] - Y[0]) - (int64_t)(X[2] - X[0]) * (int64_t)(Y[1] - Y[0]);
    if (area2 == 0) return 0.0f;
    const int64_t s = area2 < 0 ? -1 : 1;
    const float invA = 1.0f / (float)(double)(area2 * s);
    /* edge i opposite vertex i: E1 = orient(V2, V0, P), E2 = orient(V0, V1, P); dE/dx = a = -s * dy, dE/dy = b = s * dx (per sub-pixel) */
    const int64_t a1 = -s * (int64_t)(Y[0] - Y[2]), b1 = s * (int64_t)(X[0] - X[2]);
    const int64_t a2 = -s * (int64_t)(Y[1] - Y[0]), b2 = s * (int64_t)(X[1] - X[0]);
    const float e1 = d[1] - d[0], e2 = d[2] - d[0];
    const float dzdx = ((float)(double)(a1 * 256) * invA) * e1 + ((float)(double)(a2 * 256) * invA) * e2;
    const float dzdy = ((float)(double)(b1 * 256) * invA) * e1 + ((float)(double)(b2 * 256) * invA) * e2;
    const float m = fmaxf(fabsf(dzdx), fabsf(dzdy));
    const float mz = fmaxf(fabsf(d[0]), fmaxf(fabsf(d[1]), fabsf(d[2])));
    const int32_t e = (int32_t)((f2u(mz) >> 23) & 0xFFu) - 23;
    const float r = (e > 0 && e < 255) ? u2f((uint32_t)e << 23) : 0.0f;
    return pm->biasSlope * m + pm->biasConst * r;
}

static void raster_cluster_pass(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv, const ChordDrawCmd* cmd,
                                const OrcShard* shard, uint64_t* vis, OrcRasterStats* st, int atomic, const PassMode* pm);

static void raster_cluster(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv, const ChordDrawCmd* cmd,
                           const OrcShard* shard, uint64_t* vis, OrcRasterStats* st, int atomic)
{
    raster_cluster_pass(scene, iv, cmd, shard, vis, st, atomic, &kClusterPass);
}

static void raster_cluster_pass(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv, const ChordDrawCmd* cmd,
                                const OrcShard* shard, uint64_t* vis, OrcRasterStats* st, int atomic, const PassMode* pm)
{
    /* mesh_raster.hlsl:66-185 */
    const ChordObject* obj = &scene->objects[cmd->objectId];
    const ChordPrimitive* prim = &scene->primitives[obj->GLTFPrimitiveDetail];
    const ChordMaterial* mat = &scene->materials[obj->GLTFMaterialData];
    const ChordAssetDesc* as = &scene->assets[prim->primitiveDatasBufferId];
    const ChordMeshlet* m = &as->meshlets[cmd->meshletId];
    const uint32_t V = m->vertexTriangleCount & 0xFFu, T = (m->vertexTriangleCount >> 8) & 0xFFu;
    const int twoSided = mat->bTwoSided != 0 || pm->depthOnly;   /* mesh_raster.cpp:224-235: DIM_TWO_SIDED bucket; depth passes: always (:188-190) */
    const int masked = mat->alphaMode == CHORD_ALPHA_MASK;   /* DIM_MASKED_MATERIAL bucket */
    MaskCtx mkStore; MaskCtx* mk = masked ? &mkStore : NULL;
    const float W = iv->renderDimension[0], H = iv->renderDimension[1];
    const uint32_t Wi = (uint32_t)W, Hi = (uint32_t)H;

    ChordMat4 mvp;
    mul_mm(&iv->translatedWorldToClip, &obj->basicData.localToTranslatedWorld, &mvp);

    f4 hs[CHORD_MESHLET_MAX_VERTICES + 1];
    float tus[CHORD_MESHLET_MAX_VERTICES + 1], tvs[CHORD_MESHLET_MAX_VERTICES + 1];
    for (uint32_t i = 0; i < V; i++) {
        uint32_t vi = prim->vertexOffset + as->meshletData[m->dataOffset + i];
        const float* p = &as->positions[(size_t)vi * 3];
        hs[i] = mul_mv(&mvp, p[0], p[1], p[2], 1.0f);
        tus[i] = tvs[i] = 0.0f;
        if (masked && as->texcoord0 && vi < as->texcoord0Count) { tus[i] = as->texcoord0[(size_t)vi * 2]; tvs[i] = as->texcoord0[(size_t)vi * 2 + 1]; }   /* mesh_raster.hlsl:109 */
    }
    if (st) { st->clusters++; st->trianglesSubmitted += T; }
    if (mat->alphaMode >= CHORD_ALPHA_BLEND) return;     /* neither bucket of renderMesh draws it (mesh_raster.cpp:224) */

    for (uint32_t t = 0; t < T; t++) {
        uint32_t packedIdx = as->meshletData[m->dataOffset + V + t];
        uint32_t idx[3] = {packedIdx & 0xFFu, (packedIdx >> 8) & 0xFFu, (packedIdx >> 16) & 0xFFu};
        f4 h[3] = {hs[idx[0]], hs[idx[1]], hs[idx[2]]};

        /* #0 back face (homogeneous determinant on x,y,w) */
        if (!twoSided) {
            float det = (h[0].x * (h[1].y * h[2].w - h[1].w * h[2].y)
                       - h[0].y * (h[1].x * h[2].w - h[1].w * h[2].x))
                       + h[0].w * (h[1].x * h[2].y - h[1].y * h[2].x);
            if (det <= 0.0f) { if (st) st->trianglesBackface++; continue; }
        }
        /* #1 all behind */
        if (h[0].w <= 0.0f && h[1].w <= 0.0f && h[2].w <= 0.0f) { if (st) st->trianglesNear++; continue; }
        float u[3], v[3];
        for (int i = 0; i < 3; i++) {
            u[i] = h[i].x / fabsf(h[i].w) * 0.5f + 0.5f;
            v[i] = h[i].y / fabsf(h[i].w) * -0.5f + 0.5f;
        }
        float maxU = fmaxf(u[0], fmaxf(u[1], u[2])), maxV = fmaxf(v[0], fmaxf(v[1], v[2]));
        float minU = fminf(u[0], fminf(u[1], u[2])), minV = fminf(v[0], fminf(v[1], v[2]));
        /* #2 off screen */
        if ((minU >= 1.0f || minV >= 1.0f) || (maxU <= 0.0f || maxV <= 0.0f)) { if (st) st->trianglesOffscreen++; continue; }
        /* #3 small primitive */
        if (rintf(minU * W) == rintf(maxU * W) || rintf(minV * H) == rintf(maxV * H)) { if (st) st->trianglesSmall++; continue; }

        uint32_t payload = pm->depthOnly ? 0u : chord_encode_triangle_instance(t, cmd->slot);
        if (vertex_fast(&h[0], pm->depthClamp) && vertex_fast(&h[1], pm->depthClamp) && vertex_fast(&h[2], pm->depthClamp)) {
            int32_t X[3], Y[3]; float d[3];
            for (int i = 0; i < 3; i++) {
                X[i] = (int32_t)rintf((u[i] * W) * 256.0f);
                Y[i] = (int32_t)rintf((v[i] * H) * 256.0f);
                d[i] = h[i].z / h[i].w;
            }
            const float tu3[3] = {tus[idx[0]], tus[idx[1]], tus[idx[2]]}, tv3[3] = {tvs[idx[0]], tvs[idx[1]], tvs[idx[2]]};
            const float tw3[3] = {h[0].w, h[1].w, h[2].w};
            if (pm->depthOnly) { const float o = depth_bias(pm, X, Y, d); d[0] += o; d[1] += o; d[2] += o; }
            raster_snapped(X, Y, d, twoSided, payload, Wi, Hi, shard, vis, st, atomic, mk, scene, mat, tu3, tv3, tw3, pm->depthClamp);
        } else {
            if (st) st->trianglesClipped++;
            f4 poly[2][12];
            float pu[2][12], pv[2][12];                  /* texture coordinates ride along (same t, from the inside end) */
            int n = 3, cur = 0;
            poly[0][0] = h[0]; poly[0][1] = h[1]; poly[0][2] = h[2];
            for (int i = 0; i < 3; i++) { pu[0][i] = tus[idx[i]]; pv[0][i] = tvs[idx[i]]; }
            for (int k = pm->depthClamp ? 2 : 0; k < 6 && n >= 3; k++) {
                int m2 = 0;
                f4* in = poly[cur]; f4* out = poly[cur ^ 1];
                for (int i = 0; i < n; i++) {
                    const int j = (i + 1) % n;
                    const f4* P = &in[i]; const f4* Q = &in[j];
                    float dp = clip_dist(P, k), dq = clip_dist(Q, k);
                    int pin = dp >= 0.0f, qin = dq >= 0.0f;
                    if (pin) { pu[cur ^ 1][m2] = pu[cur][i]; pv[cur ^ 1][m2] = pv[cur][i]; out[m2++] = *P; }
                    if (pin && !qin) {
                        const float t = dp / (dp - dq);
                        pu[cur ^ 1][m2] = pu[cur][i] + (pu[cur][j] - pu[cur][i]) * t; pv[cur ^ 1][m2] = pv[cur][i] + (pv[cur][j] - pv[cur][i]) * t;
                        out[m2++] = clip_intersect(P, Q, dp, dq);
                    } else if (!pin && qin) {
                        const float t = dq / (dq - dp);
                        pu[cur ^ 1][m2] = pu[cur][j] + (pu[cur][i] - pu[cur][j]) * t; pv[cur ^ 1][m2] = pv[cur][j] + (pv[cur][i] - pv[cur][j]) * t;
                        out[m2++] = clip_intersect(Q, P, dq, dp);
                    }
                }
                n = m2; cur ^= 1;
            }
            if (n < 3) continue;
            int ok = 1;
            int32_t PX[12], PY[12]; float PD[12];
            for (int i = 0; i < n; i++) {
                if (!(poly[cur][i].w > 0.0f)) { ok = 0; break; }
                snap_vertex(&poly[cur][i], W, H, &PX[i], &PY[i], &PD[i]);
            }
            if (!ok) continue;
            for (int i = 1; i + 1 < n; i++) {
                int32_t X[3] = {PX[0], PX[i], PX[i + 1]}, Y[3] = {PY[0], PY[i], PY[i + 1]};
                float d[3] = {PD[0], PD[i], PD[i + 1]};
                const float tu3[3] = {pu[cur][0], pu[cur][i], pu[cur][i + 1]}, tv3[3] = {pv[cur][0], pv[cur][i], pv[cur][i + 1]};
                const float tw3[3] = {poly[cur][0].w, poly[cur][i].w, poly[cur][i + 1].w};
                if (pm->depthOnly) { const float o = depth_bias(pm, X, Y, d); d[0] += o; d[1] += o; d[2] += o; }
                raster_snapped(X, Y, d, twoSided, payload, Wi, Hi, shard, vis, st, atomic, mk, scene, mat, tu3, tv3, tw3, pm->depthClamp);
            }
        }
    }
}

void orc_raster(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv,
                const ChordDrawCmd* cmds, uint32_t count, const OrcShard* shard,
                uint64_t* vis, OrcRasterStats* stats)
{
    for (uint32_t i = 0; i < count; i++) raster_cluster(scene, iv, &cmds[i], shard, vis, stats, 0);
}

typedef struct {
    const ChordSceneDesc* scene; const ChordInstanceCullingView* iv; const ChordDrawCmd* cmds;
    uint32_t count; uint32_t* next; uint64_t* vis; OrcRasterStats stats;
} MtJob;

static void* mt_worker(void* p)
{
    MtJob* j = (MtJob*)p;
    for (;;) {
        uint32_t i = __atomic_fetch_add(j->next, 64u, __ATOMIC_RELAXED);
        if (i >= j->count) break;
        uint32_t e = i + 64u < j->count ? i + 64u : j->count;
        for (; i < e; i++) raster_cluster(j->scene, j->iv, &j->cmds[i], NULL, j->vis, &j->stats, 1);
    }
    return NULL;
}

void orc_raster_mt(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv,
                   const ChordDrawCmd* cmds, uint32_t count, uint32_t threads,
                   uint64_t* vis, OrcRasterStats* stats)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; MtJob jobs[256]; uint32_t next = 0;
    for (uint32_t t = 0; t < threads; t++) {
        memset(&jobs[t], 0, sizeof(MtJob));
        jobs[t].scene = scene; jobs[t].iv = iv; jobs[t].cmds = cmds; jobs[t].count = count; jobs[t].next = &next; jobs[t].vis = vis;
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    for (uint32_t t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        if (stats) {
            uint64_t* a = (uint64_t*)stats; const uint64_t* b = (const uint64_t*)&jobs[t].stats;
            for (size_t k = 0; k < sizeof(OrcRasterStats) / 8; k++) a[k] += b[k];
        }
    }
}

/* renderMeshDepth (mesh_raster.cpp:159-206) into a cleared 64-bit buffer whose high words are the D32 image */
void orc_raster_depth(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv, const ChordDrawCmd* cmds, uint32_t count,
                      int depthClamp, float biasConst, float biasSlope, uint64_t* vis, OrcRasterStats* stats)
{
    PassMode pm = {1, depthClamp, biasConst, biasSlope};
    for (uint32_t i = 0; i < count; i++) raster_cluster_pass(scene, iv, &cmds[i], NULL, vis, stats, 0, &pm);
}

/* hzb_culling_generic.hlsl:37-172 -- one-pass occlusion cull of a (shadow) view against an HZB of that view's depth */
uint32_t orc_hzb_culling_generic(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv, const double mainCameraWorldPos[3],
                                 uint32_t flags, float extentScale, int bObjectUseLastFrameProject,
                                 const ChordHZBDesc* hzb, const uint16_t* hzbMin,
                                 const ChordDrawCmd* inCmds, uint32_t inCount, ChordDrawCmd* outCmds)
{
    uint32_t nv = 0;
    double ivCam[3];
    memcpy(ivCam, iv->cameraWorldPos, sizeof(ivCam));               /* GPUStorageDouble4: the first three doubles */
    const float rel[3] = {(float)(mainCameraWorldPos[0] - ivCam[0]), (float)(mainCameraWorldPos[1] - ivCam[1]), (float)(mainCameraWorldPos[2] - ivCam[2])};   /* :78 */
    for (uint32_t i = 0; i < inCount; i++) {
        const ChordDrawCmd* cmd = &inCmds[i];
        const ChordObject* obj = &scene->objects[cmd->objectId];
        const ChordPrimitive* prim = &scene->primitives[obj->GLTFPrimitiveDetail];
        const ChordAssetDesc* as = &scene->assets[prim->primitiveDatasBufferId];
        const ChordMeshlet* m = &as->meshlets[cmd->meshletId];
        int visible = 1;
        if (flags & CHORD_FLAG_HZB_CULL) {
            f3 c, e;
            aabb_center_extent(m->posMin, m->posMax, 1, &c, &e);
            ChordMat4 l2w = bObjectUseLastFrameProject ? obj->basicData.localToTranslatedWorldLastFrame : obj->basicData.localToTranslatedWorld;
            MAT(&l2w, 0, 3) += rel[0]; MAT(&l2w, 1, 3) += rel[1]; MAT(&l2w, 2, 3) += rel[2];     /* :79-81 */
            ChordMat4 mvp;
            mul_mm(&iv->translatedWorldToClip, &l2w, &mvp);
            f3 mx = {-10.0f, -10.0f, -10.0f}, mn = {10.0f, 10.0f, 10.0f};
            int can = 1;
            for (int k = 0; k < 8; k++) {
                f3 uvz = project_pos_to_uvz(extent_corner(c, e, k), &mvp);
                mn.x = fminf(mn.x, uvz.x); mn.y = fminf(mn.y, uvz.y); mn.z = fminf(mn.z, uvz.z);
                mx.x = fmaxf(mx.x, uvz.x); mx.y = fmaxf(mx.y, uvz.y); mx.z = fmaxf(mx.z, uvz.z);
                can = can && (uvz.x < 1.0f && uvz.y < 1.0f && uvz.z < 1.0f) && (uvz.x > 0.0f && uvz.y > 0.0f && uvz.z > 0.0f);   /* :95 */
            }
            if (can) {
                const float W = iv->renderDimension[0], H = iv->renderDimension[1];
                int32_t rx = (int32_t)(mn.x * W + extentScale * -0.5f);           /* :103 */
                int32_t ry = (int32_t)(mn.y * H + extentScale * -0.5f);
                int32_t rz = (int32_t)(mx.x * W + extentScale * 0.5f);
                int32_t rw = (int32_t)(mx.y * H + extentScale * 0.5f);
                if (rx < 0) rx = 0;
                if (ry < 0) ry = 0;
                rz = (int32_t)fminf(W - 1.0f, (float)rz);
                rw = (int32_t)fminf(H - 1.0f, (float)rw);
                if (rz < rx || rw < ry) visible = 0;
                else {
                    const int32_t mx0 = rx >> 1, my0 = ry >> 1, mz0 = rz >> 1, mw0 = rw >> 1;
                    const int lx = first_bit_high(mz0 - mx0), ly = first_bit_high(mw0 - my0);
                    int lv = lx > ly ? lx : ly;
                    if (lv < 0) lv = 0;
                    if (((mz0 >> lv) - (mx0 >> lv) >= 2) || ((mw0 >> lv) - (my0 >> lv) >= 2)) lv += 1;   /* :125-126 */
                    if (lv > (int)hzb->mipCount - 1) lv = (int)hzb->mipCount - 1;   /* (a Load beyond the last mip returns 0: never reached at these sizes) */
                    const int32_t cx = mx0 >> lv, cy = my0 >> lv, cz = mz0 >> lv, cw = mw0 >> lv;
                    float zMin = 10.0f;
                    const uint32_t mw = mip_w(hzb, (uint32_t)lv);
                    for (int x = 0; x < 2; x++) for (int y = 0; y < 2; y++) {
                        const int32_t sx = cx + x < cz ? cx + x : cz, sy = cy + y < cw ? cy + y : cw;
                        zMin = fminf(zMin, orc_f16_to_f32(hzbMin[hzb->mipOffset[lv] + (size_t)sy * mw + (size_t)sx]));
                    }
                    if (zMin > mx.z) visible = 0;
                }
            }
        }
        if (visible) outCmds[nv++] = *cmd;
    }
    return nv;
}

/* ------------------------------------------------------------------ frame -- */

void orc_frame(const ChordSceneDesc* scene, const ChordCameraView* view, const ChordInstanceCullingView* iv,
               uint32_t flags, const uint16_t* prevHzbMin, const OrcShard* shard,
               uint64_t* vis, ChordDrawCmd* outCmds, uint32_t cmdCap, uint32_t counts[4],
               uint16_t* outHzbMin, uint16_t* outHzbMax, uint32_t outValidRange[2],
               OrcRasterStats* stats)
{
    const uint32_t W = (uint32_t)iv->renderDimension[0], H = (uint32_t)iv->renderDimension[1];
    ChordHZBDesc hd;
    orc_hzb_desc(W, H, &hd);
    memset(vis, 0, sizeof(uint64_t) * (size_t)W * H);           /* render_textures.cpp:81-85,98-100 */
    if (counts) memset(counts, 0, sizeof(uint32_t) * 4);

    uint32_t n = orc_instance_culling(scene, view, iv, flags, outCmds, cmdCap);   /* renderer.cpp:321 */
    if (n > cmdCap) n = cmdCap;
    if (counts) counts[0] = n;

    if (prevHzbMin && (flags & CHORD_FLAG_HZB_CULL)) {                           /* mesh_raster.cpp:293 */
        ChordDrawCmd* visL = (ChordDrawCmd*)malloc(sizeof(ChordDrawCmd) * (n ? n : 1));
        ChordDrawCmd* rejL = (ChordDrawCmd*)malloc(sizeof(ChordDrawCmd) * (n ? n : 1));
        uint16_t* tmpHzb = (uint16_t*)malloc(sizeof(uint16_t) * hd.totalTexels);
        uint32_t nv = 0, nr = 0, nv1 = 0;
        orc_hzb_culling(scene, view, flags, 0, &hd, prevHzbMin, outCmds, n, visL, &nv, rejL, &nr);
        orc_raster(scene, iv, visL, nv, shard, vis, stats);                       /* stage 0 */
        orc_hzb_build(vis, W, H, &hd, tmpHzb, NULL, NULL);                        /* renderer.cpp:334 */
        orc_hzb_culling(scene, view, flags, 1, &hd, tmpHzb, rejL, nr, visL, &nv1, NULL, NULL);
        orc_raster(scene, iv, visL, nv1, shard, vis, stats);                      /* stage 1 */
        if (counts) { counts[1] = nv; counts[2] = nr; counts[3] = nv1; }
        free(visL); free(rejL); free(tmpHzb);
    } else {
        orc_raster(scene, iv, outCmds, n, shard, vis, stats);                     /* mesh_raster.cpp:307 */
        if (counts) counts[1] = n;
    }
    if (outHzbMin) orc_hzb_build(vis, W, H, &hd, outHzbMin, outHzbMax, outValidRange);  /* renderer.cpp:343 */
}

/* ================================================================================================
 * All-cores replay of orc_frame (SURVEY 8d "CPU baseline": threads over clusters, per-thread tile-private images merged by
 * max; culling and the HZB builds over ranges).  Same functions, same results as orc_frame -- the merge is a max, the lists
 * are concatenated in range order -- checked by tests/test_oracle_kat.py.  Not part of any product path.
 * ================================================================================================ */
typedef void (*ParFn)(void* ctx, uint32_t t, uint32_t T);

/* A frame is ~20 parallel sections; creating and joining `threads` threads for each costs more than most of them do (64 threads:
 * 1 300 pthread_create per frame).  Workers are created once, sleep on a condition variable between sections and are told apart
 * by a generation counter; worker k takes part in a section iff k < T.  One replay at a time (the callers hold g_privLock or are
 * single-threaded tests); the workers are never joined -- the process ends with them. */
static pthread_mutex_t g_poolLock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_poolGo = PTHREAD_COND_INITIALIZER, g_poolDone = PTHREAD_COND_INITIALIZER;
static pthread_mutex_t g_poolRun = PTHREAD_MUTEX_INITIALIZER;          /* serialises par_run callers */
static uint32_t g_poolThreads = 0, g_poolGen = 0, g_poolT = 0, g_poolLeft = 0;
static ParFn g_poolFn = NULL;
static void* g_poolCtx = NULL;

static void* pool_worker(void* arg)
{
    const uint32_t k = (uint32_t)(uintptr_t)arg;
    uint32_t seen = 0;
    pthread_mutex_lock(&g_poolLock);
    for (;;) {
        while (g_poolGen == seen) pthread_cond_wait(&g_poolGo, &g_poolLock);
        seen = g_poolGen;
        if (k >= g_poolT) continue;
        ParFn fn = g_poolFn; void* ctx = g_poolCtx; const uint32_t T = g_poolT;
        pthread_mutex_unlock(&g_poolLock);
        fn(ctx, k, T);
        pthread_mutex_lock(&g_poolLock);
        if (--g_poolLeft == 0) pthread_cond_signal(&g_poolDone);
    }
    return NULL;
}

static void par_run(uint32_t T, ParFn fn, void* ctx)
{
    if (T < 1) T = 1;
    if (T > 256) T = 256;
    if (T == 1) { fn(ctx, 0, 1); return; }
    pthread_mutex_lock(&g_poolRun);
    pthread_mutex_lock(&g_poolLock);
    while (g_poolThreads + 1 < T) {                                /* workers 1 .. T-1 (the caller is thread 0) */
        pthread_t th;
        g_poolThreads++;
        pthread_create(&th, NULL, pool_worker, (void*)(uintptr_t)g_poolThreads);
        pthread_detach(th);
    }
    g_poolFn = fn; g_poolCtx = ctx; g_poolT = T; g_poolLeft = T - 1; g_poolGen++;
    pthread_cond_broadcast(&g_poolGo);
    pthread_mutex_unlock(&g_poolLock);
    fn(ctx, 0, T);
    pthread_mutex_lock(&g_poolLock);
    while (g_poolLeft != 0) pthread_cond_wait(&g_poolDone, &g_poolLock);
    pthread_mutex_unlock(&g_poolLock);
    pthread_mutex_unlock(&g_poolRun);
}
static inline uint32_t part_lo(uint32_t n, uint32_t t, uint32_t T) { return (uint32_t)((uint64_t)n * t / T); }

/* ---- instance culling over object ranges ---- */
typedef struct {
    const ChordSceneDesc* scene; const ChordCameraView* view; const ChordInstanceCullingView* iv; uint32_t flags;
    ChordDrawCmd** part; uint32_t* partCount; ChordDrawCmd* out; uint32_t cap; uint32_t* base;
} CullMt;
static void cull_mt_count(void* c, uint32_t t, uint32_t T)
{
    CullMt* m = (CullMt*)c;
    const uint32_t o0 = part_lo(m->scene->objectCount, t, T), o1 = part_lo(m->scene->objectCount, t + 1, T);
    /* first the count (nothing stored), then the commands into a buffer of exactly that size */
    const uint32_t n = instance_culling_range(m->scene, m->view, m->iv, m->flags, o0, o1, NULL, 0, 0);
    m->part[t] = (ChordDrawCmd*)malloc(sizeof(ChordDrawCmd) * (n ? n : 1));
    m->partCount[t] = instance_culling_range(m->scene, m->view, m->iv, m->flags, o0, o1, m->part[t], n, 0);
}
static void cull_mt_copy(void* c, uint32_t t, uint32_t T)
{
    CullMt* m = (CullMt*)c;
    (void)T;
    for (uint32_t i = 0; i < m->partCount[t]; i++) {
        const uint32_t k = m->base[t] + i;
        if (k >= m->cap) break;
        m->out[k] = m->part[t][i]; m->out[k].slot = k;                      /* slot = index in the whole list */
    }
    free(m->part[t]);
}
static uint32_t instance_culling_mt(const ChordSceneDesc* scene, const ChordCameraView* view, const ChordInstanceCullingView* iv,
                                    uint32_t flags, ChordDrawCmd* outCmds, uint32_t cap, uint32_t T)
{
    ChordDrawCmd* part[256]; uint32_t cnt[256], base[257];
    CullMt m = {scene, view, iv, flags, part, cnt, outCmds, cap, base};
    par_run(T, cull_mt_count, &m);
    base[0] = 0;
    for (uint32_t t = 0; t < T; t++) base[t + 1] = base[t] + cnt[t];
    par_run(T, cull_mt_copy, &m);
    return base[T];
}

/* ---- HZB culling over command ranges ---- */
typedef struct {
    const ChordSceneDesc* scene; const ChordCameraView* view; uint32_t flags; int phase; const ChordHZBDesc* hzb; const uint16_t* hzbMin;
    const ChordDrawCmd* in; uint32_t inCount; uint8_t* verdict; uint32_t* nv; uint32_t* nr; uint32_t* bv; uint32_t* br;
    ChordDrawCmd* outV; ChordDrawCmd* outR;
} HzbMt;
static void hzb_mt_test(void* c, uint32_t t, uint32_t T)
{
    HzbMt* m = (HzbMt*)c;
    const uint32_t i0 = part_lo(m->inCount, t, T), i1 = part_lo(m->inCount, t + 1, T);
    uint32_t nv = 0, nr = 0;
    for (uint32_t i = i0; i < i1; i++) {
        const int v = hzb_visible(m->scene, m->view, m->flags, m->phase, m->hzb, m->hzbMin, &m->in[i]);
        m->verdict[i] = (uint8_t)v;
        if (v) nv++; else nr++;
    }
    m->nv[t] = nv; m->nr[t] = nr;
}
static void hzb_mt_write(void* c, uint32_t t, uint32_t T)
{
    HzbMt* m = (HzbMt*)c;
    const uint32_t i0 = part_lo(m->inCount, t, T), i1 = part_lo(m->inCount, t + 1, T);
    uint32_t kv = m->bv[t], kr = m->br[t];
    for (uint32_t i = i0; i < i1; i++) {
        if (m->verdict[i]) m->outV[kv++] = m->in[i];
        else if (m->phase == 0 && m->outR) m->outR[kr++] = m->in[i];
    }
}
static void hzb_culling_mt(const ChordSceneDesc* scene, const ChordCameraView* view, uint32_t flags, int phase,
                           const ChordHZBDesc* hzb, const uint16_t* hzbMin, const ChordDrawCmd* inCmds, uint32_t inCount,
                           ChordDrawCmd* outVisible, uint32_t* outVisibleCount, ChordDrawCmd* outRejected, uint32_t* outRejectedCount, uint32_t T)
{
    uint32_t nv[256], nr[256], bv[257], br[257];
    uint8_t* verdict = (uint8_t*)malloc(inCount ? inCount : 1);
    HzbMt m = {scene, view, flags, phase, hzb, hzbMin, inCmds, inCount, verdict, nv, nr, bv, br, outVisible, outRejected};
    par_run(T, hzb_mt_test, &m);
    bv[0] = br[0] = 0;
    for (uint32_t t = 0; t < T; t++) { bv[t + 1] = bv[t] + nv[t]; br[t + 1] = br[t] + nr[t]; }
    par_run(T, hzb_mt_write, &m);
    *outVisibleCount = bv[T];
    if (outRejectedCount) *outRejectedCount = br[T];
    free(verdict);
}

/* ---- raster: clusters handed out in chunks, every thread into its own tile-linear image; then the touched tiles are merged
 *      into the frame's image by max, tile ranges in parallel ---- */
typedef struct {
    const ChordSceneDesc* scene; const ChordInstanceCullingView* iv; const ChordDrawCmd* cmds; uint32_t count; uint32_t* next;
    uint64_t** priv; uint8_t** dirty; OrcRasterStats* stats; uint32_t W, H, tilesX, tilesY; uint64_t* vis;
} RasterMt;
static void raster_mt_draw(void* c, uint32_t t, uint32_t T)
{
    RasterMt* m = (RasterMt*)c;
    (void)T;
    t_dirty = m->dirty[t];
    for (;;) {
        /* (8 clusters per grab: a cluster next to the camera can hold more fragments than a thousand distant ones) */
        uint32_t i = __atomic_fetch_add(m->next, 8u, __ATOMIC_RELAXED);
        if (i >= m->count) break;
        const uint32_t e = i + 8u < m->count ? i + 8u : m->count;
        for (; i < e; i++) raster_cluster(m->scene, m->iv, &m->cmds[i], NULL, m->priv[t], &m->stats[t], 2);
    }
    t_dirty = NULL;
}
static void raster_mt_merge(void* c, uint32_t t, uint32_t T)
{
    RasterMt* m = (RasterMt*)c;
    const uint32_t tiles = m->tilesX * m->tilesY, k0 = part_lo(tiles, t, T), k1 = part_lo(tiles, t + 1, T);
    for (uint32_t k = k0; k < k1; k++)
        for (uint32_t q = 0; q < T; q++) {
            if (!m->dirty[q][k]) continue;
            const uint32_t ox = (k % m->tilesX) * ORC_PRIV_TILE, oy = (k / m->tilesX) * ORC_PRIV_TILE;
            uint64_t* src = m->priv[q] + (size_t)k * ORC_PRIV_TILE * ORC_PRIV_TILE;
            for (uint32_t y = 0; y < ORC_PRIV_TILE && oy + y < m->H; y++)
                for (uint32_t x = 0; x < ORC_PRIV_TILE && ox + x < m->W; x++) {
                    const uint64_t v = src[y * ORC_PRIV_TILE + x];
                    uint64_t* dst = &m->vis[(size_t)(oy + y) * m->W + ox + x];
                    if (v > *dst) *dst = v;
                }
            /* the private images live across calls (below): a merged tile goes back to all zero */
            memset(src, 0, sizeof(uint64_t) * ORC_PRIV_TILE * ORC_PRIV_TILE);
            m->dirty[q][k] = 0;
        }
}

/* The threads' private images are kept between calls: mapping and zero-filling them anew for every raster leg costs more than
 * the leg (page faults of one process serialise), and after a merge they are all zero again.  One replay at a time. */
static pthread_mutex_t g_privLock = PTHREAD_MUTEX_INITIALIZER;
static uint64_t* g_priv[256];
static uint8_t* g_dirty[256];
static size_t g_privBytes = 0;
static uint32_t g_privTiles = 0;

static void raster_private_mt(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv, const ChordDrawCmd* cmds, uint32_t count,
                              uint32_t T, uint32_t W, uint32_t H, uint64_t* vis, OrcRasterStats* stats)
{
    OrcRasterStats st[256]; uint32_t next = 0;
    const uint32_t tilesX = (W + ORC_PRIV_TILE - 1u) / ORC_PRIV_TILE, tilesY = (H + ORC_PRIV_TILE - 1u) / ORC_PRIV_TILE;
    const size_t bytes = (size_t)tilesX * tilesY * ORC_PRIV_TILE * ORC_PRIV_TILE * sizeof(uint64_t);
    if (T > 256) T = 256;
    pthread_mutex_lock(&g_privLock);
    if (bytes != g_privBytes) {                                   /* another image size: start over */
        for (uint32_t t = 0; t < 256; t++) {
            if (g_priv[t]) { munmap(g_priv[t], g_privBytes); g_priv[t] = NULL; }
            free(g_dirty[t]); g_dirty[t] = NULL;
        }
        g_privBytes = bytes; g_privTiles = tilesX * tilesY;
    }
    for (uint32_t t = 0; t < T; t++) {
        /* anonymous mapping: zero pages, mapped on first touch -- a thread pays for the tiles it draws into, not for the screen */
        if (!g_priv[t]) g_priv[t] = (uint64_t*)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (!g_dirty[t]) g_dirty[t] = (uint8_t*)calloc(g_privTiles, 1);
        memset(&st[t], 0, sizeof(OrcRasterStats));
    }
    RasterMt m = {scene, iv, cmds, count, &next, g_priv, g_dirty, st, W, H, tilesX, tilesY, vis};
    par_run(T, raster_mt_draw, &m);
    par_run(T, raster_mt_merge, &m);
    pthread_mutex_unlock(&g_privLock);
    for (uint32_t t = 0; t < T; t++)
        if (stats) {
            uint64_t* a = (uint64_t*)stats; const uint64_t* b = (const uint64_t*)&st[t];
            for (size_t k = 0; k < sizeof(OrcRasterStats) / 8; k++) a[k] += b[k];
        }
}

/* ---- HZB build over row ranges ---- */
typedef struct {
    const uint64_t* vis; uint32_t W, H; const ChordHZBDesc* desc; uint16_t* hzbMin; uint16_t* hzbMax; uint32_t level; uint32_t* vmin; uint32_t* vmax;
} HzbBuildMt;
static void hzb_build_mt_rows(void* c, uint32_t t, uint32_t T)
{
    HzbBuildMt* m = (HzbBuildMt*)c;
    const uint32_t vh = orc_hzb_valid_h(m->desc, m->level), y0 = part_lo(vh, t, T), y1 = part_lo(vh, t + 1, T);
    if (m->level == 0) { m->vmin[t] = 0xFFFFFFFFu; m->vmax[t] = 0u; hzb_mip0_rows(m->vis, m->W, m->H, m->desc, m->hzbMin, m->hzbMax, y0, y1, &m->vmin[t], &m->vmax[t]); }
    else hzb_mip_rows(m->desc, m->level, m->hzbMin, m->hzbMax, y0, y1);
}
static void hzb_build_mt(const uint64_t* vis, uint32_t W, uint32_t H, const ChordHZBDesc* desc,
                         uint16_t* hzbMin, uint16_t* hzbMax, uint32_t validRange[2], uint32_t T)
{
    uint32_t vmin[256], vmax[256];
    memset(hzbMin, 0, sizeof(uint16_t) * desc->totalTexels);
    if (hzbMax) memset(hzbMax, 0, sizeof(uint16_t) * desc->totalTexels);
    HzbBuildMt m = {vis, W, H, desc, hzbMin, hzbMax, 0, vmin, vmax};
    for (uint32_t l = 0; l < desc->mipCount; l++) {
        m.level = l;
        const uint32_t vh = orc_hzb_valid_h(desc, l);
        par_run(vh >= 4u * T ? T : 1u, hzb_build_mt_rows, &m);            /* (a level is complete before the next one reads it) */
        if (l == 0 && validRange) {
            uint32_t lo = 0xFFFFFFFFu, hi = 0u;
            for (uint32_t t = 0; t < (vh >= 4u * T ? T : 1u); t++) { if (vmin[t] < lo) lo = vmin[t]; if (vmax[t] > hi) hi = vmax[t]; }
            validRange[0] = lo; validRange[1] = hi;
        }
    }
}

typedef struct { uint64_t* vis; size_t words; } ClearMt;
static void clear_mt_part(void* c, uint32_t t, uint32_t T)
{
    ClearMt* m = (ClearMt*)c;
    const size_t a = m->words * t / T, b = m->words * (t + 1) / T;
    memset(m->vis + a, 0, sizeof(uint64_t) * (b - a));
}

void orc_frame_mt(const ChordSceneDesc* scene, const ChordCameraView* view, const ChordInstanceCullingView* iv,
                  uint32_t flags, const uint16_t* prevHzbMin, uint32_t threads,
                  uint64_t* vis, ChordDrawCmd* outCmds, uint32_t cmdCap, uint32_t counts[4],
                  uint16_t* outHzbMin, uint16_t* outHzbMax, uint32_t outValidRange[2],
                  OrcRasterStats* stats)
{
    const uint32_t W = (uint32_t)iv->renderDimension[0], H = (uint32_t)iv->renderDimension[1];
    const uint32_t T = threads < 1 ? 1 : threads > 256 ? 256 : threads;
    ChordHZBDesc hd;
    orc_hzb_desc(W, H, &hd);
    ClearMt cl = {vis, (size_t)W * H};
    par_run(T, clear_mt_part, &cl);
    if (counts) memset(counts, 0, sizeof(uint32_t) * 4);

    uint32_t n = instance_culling_mt(scene, view, iv, flags, outCmds, cmdCap, T);
    if (n > cmdCap) n = cmdCap;
    if (counts) counts[0] = n;

    if (prevHzbMin && (flags & CHORD_FLAG_HZB_CULL)) {
        ChordDrawCmd* visL = (ChordDrawCmd*)malloc(sizeof(ChordDrawCmd) * (n ? n : 1));
        ChordDrawCmd* rejL = (ChordDrawCmd*)malloc(sizeof(ChordDrawCmd) * (n ? n : 1));
        uint16_t* tmpHzb = (uint16_t*)malloc(sizeof(uint16_t) * hd.totalTexels);
        uint32_t nv = 0, nr = 0, nv1 = 0;
        hzb_culling_mt(scene, view, flags, 0, &hd, prevHzbMin, outCmds, n, visL, &nv, rejL, &nr, T);
        raster_private_mt(scene, iv, visL, nv, T, W, H, vis, stats);
        hzb_build_mt(vis, W, H, &hd, tmpHzb, NULL, NULL, T);
        hzb_culling_mt(scene, view, flags, 1, &hd, tmpHzb, rejL, nr, visL, &nv1, NULL, NULL, T);
        raster_private_mt(scene, iv, visL, nv1, T, W, H, vis, stats);
        if (counts) { counts[1] = nv; counts[2] = nr; counts[3] = nv1; }
        free(visL); free(rejL); free(tmpHzb);
    } else {
        raster_private_mt(scene, iv, outCmds, n, T, W, H, vis, stats);
        if (counts) counts[1] = n;
    }
    if (outHzbMin) hzb_build_mt(vis, W, H, &hd, outHzbMin, outHzbMax, outValidRange, T);
}

/* ================================================================================================
 * Visibility tile marker and shading tile lists — visibility_tile.hlsl, visibility_tile.cpp
 * ================================================================================================ */

/* remap8x8 — base.hlsli:363-366 */
static void remap8x8(uint32_t tid, uint32_t* x, uint32_t* y)
{
    *x = (((tid >> 2) & 0x7u) & 0xFFFEu) | (tid & 0x1u);
    *y = ((tid >> 1) & 0x3u) | (((tid >> 3) & 0x7u) & 0xFFFCu);
}

/* getShadingType — visibility_tile.hlsl:39-63.  packID = the R32_UINT visibility texel = low word here. */
static uint32_t shading_type(const ChordSceneDesc* scene, uint32_t packID, const ChordDrawCmd* cmds, uint32_t cmdCount)
{
    if (packID == 0u) return 0u;                                   /* kLightingType_None, base.h:422 */
    const uint32_t instanceId = ((packID >> 8) & CHORD_MAX_INSTANCE_ID) - 1u;   /* base.hlsli:443-447 */
    if (instanceId >= cmdCount) return 0u;                         /* (the shader would read out of bounds) */
    const ChordDrawCmd cmd = cmds[instanceId];                     /* check(drawCmd.z == instanceId), :54 */
    const ChordObject* obj = &scene->objects[cmd.objectId];
    return scene->materials[obj->GLTFMaterialData].materialType;   /* :56-60 */
}

void orc_visibility_mark(const ChordSceneDesc* scene, const uint64_t* vis, uint32_t W, uint32_t H,
                         const ChordDrawCmd* cmds, uint32_t cmdCount, uint32_t* marker)
{
    const uint32_t mW = (W + 7u) / 8u, mH = (H + 7u) / 8u;        /* visibility_tile.cpp:31 */
    const uint32_t gW = (mW + 3u) / 4u, gH = (mH + 3u) / 4u;      /* dispatch, visibility_tile.cpp:54 */
    for (uint32_t gy = 0; gy < gH; gy++)
        for (uint32_t gx = 0; gx < gW; gx++) {
            uint32_t s[8][8][4];                                   /* sTileMarkerR/G/B/A[x][y], :34-37 */
            for (uint32_t tid = 0; tid < 64u; tid++) {
                uint32_t rx, ry;
                remap8x8(tid, &rx, &ry);
                uint32_t m[4] = {0, 0, 0, 0};
                for (uint32_t y = 0; y < 2u; y++)
                    for (uint32_t x = 0; x < 2u; x++) {
                        /* Gather at uv = (gatherPos + 1) * texelSize with a point/clamp-to-edge sampler: the 2x2
                         * texels gatherPos + {0,1}^2, coordinates clamped to the image (:83-87) */
                        const uint32_t px = gx * 32u + 4u * rx + 2u * x, py = gy * 32u + 4u * ry + 2u * y;
                        for (uint32_t j = 0; j < 2u; j++)
                            for (uint32_t i = 0; i < 2u; i++) {
                                const uint32_t cx = px + i < W ? px + i : W - 1u, cy = py + j < H ? py + j : H - 1u;
                                const uint32_t t = shading_type(scene, (uint32_t)(vis[(size_t)cy * W + cx] & 0xFFFFFFFFull), cmds, cmdCount);
                                m[(t / 32u) & 3u] |= 1u << (t % 32u);            /* :92-93 (types < 128) */
                            }
                    }
                for (int k = 0; k < 4; k++) s[rx][ry][k] = m[k];
            }
            for (uint32_t rx = 0; rx < 8u; rx += 2u)               /* :104-110 */
                for (uint32_t ry = 0; ry < 8u; ry++)
                    for (int k = 0; k < 4; k++) s[rx][ry][k] |= s[rx + 1u][ry][k];
            for (uint32_t rx = 0; rx < 8u; rx += 2u)               /* :112-118 (only even x is read afterwards) */
                for (uint32_t ry = 0; ry < 8u; ry += 2u)
                    for (int k = 0; k < 4; k++) s[rx][ry][k] |= s[rx][ry + 1u][k];
            for (uint32_t rx = 0; rx < 8u; rx += 2u)               /* :120-131 */
                for (uint32_t ry = 0; ry < 8u; ry += 2u) {
                    const uint32_t sx = gx * 4u + rx / 2u, sy = gy * 4u + ry / 2u;
                    if (sx < mW && sy < mH)                        /* out-of-range image stores are dropped */
                        for (int k = 0; k < 4; k++) marker[((size_t)sy * mW + sx) * 4u + (uint32_t)k] = s[rx][ry][k];
                }
        }
}

uint32_t orc_shading_tiles(const uint32_t* marker, uint32_t mW, uint32_t mH, uint32_t shadingType,
                           uint32_t* tiles, uint32_t dispatchArgs[4])
{
    const uint32_t index = shadingType / 32u, bit = 1u << (shadingType % 32u);   /* visibility_tile.cpp:76-77 */
    const uint32_t gW = (mW + 15u) / 16u, gH = (mH + 15u) / 16u;                 /* :83 */
    uint32_t count = 0;
    for (uint32_t gy = 0; gy < gH; gy++)
        for (uint32_t gx = 0; gx < gW; gx++)
            for (uint32_t tid = 0; tid < 64u; tid++) {
                uint32_t rx, ry;
                remap8x8(tid, &rx, &ry);
                for (uint32_t y = 0; y < 2u; y++)
                    for (uint32_t x = 0; x < 2u; x++) {
                        const uint32_t sx = gx * 16u + rx + x * 8u, sy = gy * 16u + ry + y * 8u;   /* :141,152 */
                        if (sx >= mW || sy >= mH) continue;                                  /* bAllInRange :170 */
                        if (!(marker[((size_t)sy * mW + sx) * 4u + (index & 3u)] & bit)) continue;   /* :169 */
                        tiles[2u * count] = sx * 8u; tiles[2u * count + 1u] = sy * 8u;       /* :174 */
                        count++;
                    }
            }
    dispatchArgs[0] = (count + 3u) / 4u; dispatchArgs[1] = 1u; dispatchArgs[2] = 1u; dispatchArgs[3] = 1u;   /* :211-216 */
    return count;
}
