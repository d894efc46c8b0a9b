// Canonical fp32 arithmetic shared by the cull / raster kernels (gfx950 device code).
//
// Source-order evaluation, no FMA contraction (the translation units are built
// with -ffp-contract=off), IEEE divide / sqrt (HIP's default
// -fhip-fp32-correctly-rounded-divide-sqrt).  Every helper states the HLSL
// expression of the reference it evaluates; SURVEY.md §8c pins the op order.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "chordvis_types.h"

namespace chord {

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

// Row-major copy of a glm column-major matrix: r[i][j] = M[row i][col j].
struct Mat4 { float r[4][4]; };

__device__ __forceinline__ Mat4 load_mat(const ChordMat4& M)
{
    Mat4 o;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) o.r[r][c] = M.m[c * 4 + r];
    return o;
}

// mul(M, float4(v, w)) — HLSL mul(matrix, column vector)
__device__ __forceinline__ f4 mul_mv(const Mat4& M, float v0, float v1, float v2, float v3)
{
    f4 o;
    o.x = ((M.r[0][0] * v0 + M.r[0][1] * v1) + M.r[0][2] * v2) + M.r[0][3] * v3;
    o.y = ((M.r[1][0] * v0 + M.r[1][1] * v1) + M.r[1][2] * v2) + M.r[1][3] * v3;
    o.z = ((M.r[2][0] * v0 + M.r[2][1] * v1) + M.r[2][2] * v2) + M.r[2][3] * v3;
    o.w = ((M.r[3][0] * v0 + M.r[3][1] * v1) + M.r[3][2] * v2) + M.r[3][3] * v3;
    return o;
}

// mul(A, B)
__device__ __forceinline__ Mat4 mul_mm(const Mat4& A, const Mat4& B)
{
    Mat4 C;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            C.r[r][c] = ((A.r[r][0] * B.r[0][c] + A.r[r][1] * B.r[1][c]) + A.r[r][2] * B.r[2][c]) + A.r[r][3] * B.r[3][c];
    return C;
}

__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// kExtentApplyFactor — base.hlsli:184-194
__device__ __forceinline__ f3 extent_corner(f3 c, f3 e, int k)
{
    const float sx = (k == 1 || k == 4 || k == 6 || k == 7) ? -1.0f : 1.0f;
    const float sy = (k == 1 || k == 3 || k == 5 || k == 6) ? -1.0f : 1.0f;
    const float sz = (k == 1 || k == 2 || k == 5 || k == 7) ? -1.0f : 1.0f;
    f3 p;
    p.x = c.x + e.x * sx;
    p.y = c.y + e.y * sy;
    p.z = c.z + e.z * sz;
    return p;
}

// projectPosToUVz — base.hlsli:161-171
__device__ __forceinline__ f3 project_pos_to_uvz(f3 pos, const Mat4& proj)
{
    f4 h = mul_mv(proj, pos.x, pos.y, pos.z, 1.0f);
    f3 r;
    r.x = h.x / h.w; r.y = h.y / h.w; r.z = h.z / h.w;
    r.x = r.x * 0.5f + 0.5f;
    r.y = r.y * -0.5f + 0.5f;
    return r;
}

// orthoFrustumCulling — base.hlsli:251-272 (true = culled)
__device__ __forceinline__ bool ortho_frustum_culling(f3 c, f3 e, const Mat4& localToClip)
{
    f3 mn = {10.0f, 10.0f, 10.0f}, mx = {-10.0f, -10.0f, -10.0f};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        f3 uvz = project_pos_to_uvz(extent_corner(c, e, k), localToClip);
        mn.x = fminf(mn.x, uvz.x); mn.y = fminf(mn.y, uvz.y); mn.z = fminf(mn.z, uvz.z);
        mx.x = fmaxf(mx.x, uvz.x); mx.y = fmaxf(mx.y, uvz.y); mx.z = fmaxf(mx.z, uvz.z);
    }
    return (mn.x >= 1.0f || mn.y >= 1.0f) || (mx.x <= 0.0f || mx.y <= 0.0f);
}

// frustumCulling — base.hlsli:275-305 (true = culled). planes: 6 x float4.
__device__ __forceinline__ bool frustum_culling(const float* __restrict__ planes, f3 c, f3 e, const Mat4& localToTranslatedWorld)
{
    f3 p[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        f3 q = extent_corner(c, e, k);
        f4 h = mul_mv(localToTranslatedWorld, q.x, q.y, q.z, 1.0f);
        p[k].x = h.x; p[k].y = h.y; p[k].z = h.z;
    }
    bool culled = false;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const f3 n = {planes[i * 4 + 0], planes[i * 4 + 1], planes[i * 4 + 2]};
        const float negw = -planes[i * 4 + 3];
        bool allBack = true;
#pragma unroll
        for (int j = 0; j < 8; j++) allBack = allBack && !(dot3(n, p[j]) > negw);
        culled = culled || allBack;
    }
    return culled;
}

__device__ __forceinline__ void aabb_center_extent(const float* mn, const float* mx, f3& c, f3& e)
{
    c.x = (mn[0] + mx[0]) * 0.5f; c.y = (mn[1] + mx[1]) * 0.5f; c.z = (mn[2] + mx[2]) * 0.5f;
    e.x = mx[0] - c.x; e.y = mx[1] - c.y; e.z = mx[2] - c.z;
}

__device__ __forceinline__ float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

__device__ __forceinline__ int first_bit_high(int v) { return v <= 0 ? -1 : 31 - __clz(v); }

// encodeTriangleIdInstanceId — base.hlsli:437-441
__device__ __forceinline__ uint32_t encode_triangle_instance(uint32_t triangleId, uint32_t instanceId)
{
    return (((instanceId + 1u) & CHORD_MAX_INSTANCE_ID) << 8) | (triangleId & 0xFFu);
}

// binary16 <-> binary32, round-to-nearest-even (v_cvt_f16_f32 / v_cvt_f32_f16)
__device__ __forceinline__ uint16_t f32_to_f16(float f) { return __half_as_ushort(__float2half_rn(f)); }
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return __half2float(__ushort_as_half(h)); }

} // namespace chord
