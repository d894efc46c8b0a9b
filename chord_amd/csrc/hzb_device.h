// Device code shared by the HZB kernels and the kernels that carry the HZB tail as an extra workgroup.
#pragma once

#include "device_layer.h"
#include "device_math.h"

namespace chord {

struct HzbParams {
    const unsigned long long* vis; int32_t W, H;   // row-major visibility words (a sharded context: the resolved copy)
    ChordHZBDesc desc;
    uint16_t* hzbMin; uint16_t* hzbMax; uint32_t* validRange;
    uint32_t* rangePartials;      // per mip-0 block {min bits, max bits}; reduced by the tail kernel
    uint32_t rangePartialCount;
};

#ifdef __HIPCC__
__device__ __forceinline__ uint32_t valid_w(const ChordHZBDesc& d, uint32_t l)
{
    const uint32_t w = max(1u, d.width >> l);
    return min(w, (((d.srcWidth - 1u) >> 1) >> l) + 1u);
}
__device__ __forceinline__ uint32_t valid_h(const ChordHZBDesc& d, uint32_t l)
{
    const uint32_t h = max(1u, d.height >> l);
    return min(h, (((d.srcHeight - 1u) >> 1) >> l) + 1u);
}
#endif

#ifdef __HIPCC__
// One block: mips 6..mipCount-1 from the stored mip 5, and the valid-range reduction.
__device__ __forceinline__ void hzb_tail_block(const HzbParams& p, int wantMax, int wantRange, uint32_t firstLevel)
{
    const ChordHZBDesc& d = p.desc;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    __shared__ uint32_t sRng[8];
    if (wantRange) {
        uint32_t mn = 0xFFFFFFFFu, mx = 0u;
        for (uint32_t i = threadIdx.x; i < p.rangePartialCount; i += 256u) {
            mn = min(mn, p.rangePartials[2u * i]);
            mx = max(mx, p.rangePartials[2u * i + 1u]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn = min(mn, (uint32_t)__shfl_down((int)mn, off, 64));
            mx = max(mx, (uint32_t)__shfl_down((int)mx, off, 64));
        }
        if (lane == 0u) { sRng[2u * wave] = mn; sRng[2u * wave + 1u] = mx; }
    }
    // The first level is reduced from memory (its four source texels per output are independent loads: one round
    // trip); every further level is reduced LDS -> LDS and stored, so the chain costs one memory round trip instead of
    // one per level.  (binary16 -> f32 -> min/max -> binary16 is exact, so reducing the kept f32 copies of the STORED
    // values gives the bits a re-read would.)
    __shared__ float sMinA[1024], sMaxA[1024], sMinB[256], sMaxB[256];
    uint32_t pw = 0, ph = 0;
    bool inLds = false, curA = true;
    for (uint32_t l = firstLevel; l < d.mipCount; l++) {
        const uint32_t vw = valid_w(d, l), vh = valid_h(d, l), mw = max(1u, d.width >> l);
        const uint32_t gw = valid_w(d, l - 1), gh = valid_h(d, l - 1), pmw = max(1u, d.width >> (l - 1));
        const bool keep = vw * vh <= (inLds ? (curA ? 256u : 1024u) : 1024u);      // fits the buffer this level is written to
        const float* rdMin = curA ? sMinA : sMinB; const float* rdMax = curA ? sMaxA : sMaxB;
        float* wrMin = inLds ? (curA ? sMinB : sMinA) : sMinA; float* wrMax = inLds ? (curA ? sMaxB : sMaxA) : sMaxA;
        for (uint32_t i = threadIdx.x; i < vw * vh; i += 256u) {
            const uint32_t x = i % vw, y = i / vw;
            float mn = 0.0f, mx = 0.0f;
#pragma unroll
            for (int jj = 0; jj < 2; jj++)
#pragma unroll
                for (int ii = 0; ii < 2; ii++) {
                    float a, b;
                    if (inLds) {
                        const uint32_t cx = min(2u * x + ii, pw - 1u), cy = min(2u * y + jj, ph - 1u);
                        a = rdMin[cy * pw + cx]; b = rdMax[cy * pw + cx];
                    } else {
                        const uint32_t cx = min(2u * x + ii, gw - 1u), cy = min(2u * y + jj, gh - 1u);
                        a = f16_to_f32(p.hzbMin[d.mipOffset[l - 1] + cy * pmw + cx]);
                        b = wantMax ? f16_to_f32(p.hzbMax[d.mipOffset[l - 1] + cy * pmw + cx]) : 0.0f;
                    }
                    if (ii == 0 && jj == 0) { mn = a; mx = b; } else { mn = fminf(mn, a); mx = fmaxf(mx, b); }
                }
            const uint16_t hmn = f32_to_f16(mn);
            const uint16_t hmx = (uint16_t)(f32_to_f16(mx) + (l == 5u ? 1u : 0u));   // storeHZBMip5
            p.hzbMin[d.mipOffset[l] + y * mw + x] = hmn;
            if (wantMax) p.hzbMax[d.mipOffset[l] + y * mw + x] = hmx;
            if (keep) { wrMin[i] = f16_to_f32(hmn); wrMax[i] = f16_to_f32(hmx); }
        }
        __syncthreads();       // level l complete (in LDS, or in memory and visible to this block) before level l+1
        if (keep) { if (inLds) curA = !curA; else { inLds = true; curA = true; } pw = vw; ph = vh; }
        else inLds = false;
    }
    if (wantRange) {
        __syncthreads();
        if (threadIdx.x == 0) {   // init {~0u, 0u}: hzb.cpp:108-109
            p.validRange[0] = min(min(sRng[0], sRng[2]), min(sRng[4], sRng[6]));
            p.validRange[1] = max(max(sRng[1], sRng[3]), max(sRng[5], sRng[7]));
        }
    }
}

#endif

HzbParams make_hzb_tail_params(ChordCtx* c, HzbBuffers& out);     // kernels_hzb.hip: the tail of a chain whose mips 0..5 the tile kernel wrote

} // namespace chord
