// Micro-benchmark: 64-bit max into a 3840x2160 buffer on MI355X.
//   mode 0: device-scope atomicMax (what a multi-XCD coherent visbuffer needs)
//   mode 1: workgroup-scope atomicMax (executes in the XCD-local L2; only valid if one XCD owns the line)
//   mode 2: plain store
//   mode 3: LDS atomic max into a 64x64 tile per workgroup (ds_max_u64)
// pattern 0: every lane its own random pixel; pattern 1: 8x8 pixel footprint per wave (row-coalesced)
// build: hipcc --offload-arch=gfx950 -O3 atomics.hip -o atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ uint32_t pcg(uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u; return (w >> 22) ^ w; }

template <int MODE, int PATTERN>
__global__ __launch_bounds__(256) void k(unsigned long long* buf, uint32_t W, uint32_t H, uint32_t iters)
{
    __shared__ unsigned long long tile[64 * 64];
    const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (MODE == 3) { for (uint32_t i = threadIdx.x; i < 4096; i += 256) tile[i] = 0; __syncthreads(); }
    for (uint32_t it = 0; it < iters; it++) {
        uint32_t r = pcg(gw * 7919u + it * 104729u + (PATTERN == 0 ? lane * 31u : 0u));
        uint32_t x, y;
        if (PATTERN == 0) { x = r % W; y = (r >> 12) % H; }
        else { x = ((r % (W / 8)) * 8) + (lane & 7); y = (((r >> 12) % (H / 8)) * 8) + (lane >> 3); }
        unsigned long long v = ((unsigned long long)(r | 1u) << 32) | lane;
        if (MODE == 0) atomicMax(&buf[(size_t)y * W + x], v);
        else if (MODE == 1) __hip_atomic_fetch_max(&buf[(size_t)y * W + x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) buf[(size_t)y * W + x] = v;
        else atomicMax(&tile[(y & 63) * 64 + (x & 63)], v);
    }
    if (MODE == 3) { __syncthreads(); if (threadIdx.x == 0) buf[blockIdx.x] = tile[0]; }
}

template <int MODE, int PATTERN>
static void run(const char* name, unsigned long long* d, uint32_t W, uint32_t H)
{
    const uint32_t blocks = 256 * 8, iters = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(blocks), dim3(256), 0, 0, d, W, H, 16u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(blocks), dim3(256), 0, 0, d, W, H, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 256 * iters;
    std::printf("%-46s %8.3f ms  %8.2f Gop/s  %8.1f GB/s(8B/op)\n", name, ms, ops / ms / 1e6, ops * 8 / ms / 1e6);
}

int main()
{
    const uint32_t W = 3840, H = 2160;
    unsigned long long* d; hipMalloc(&d, (size_t)W * H * 8); hipMemset(d, 0, (size_t)W * H * 8);
    run<0, 0>("device-scope atomicMax, scattered", d, W, H);
    run<1, 0>("workgroup-scope atomicMax, scattered", d, W, H);
    run<2, 0>("plain store, scattered", d, W, H);
    run<0, 1>("device-scope atomicMax, 8x8 tile per wave", d, W, H);
    run<1, 1>("workgroup-scope atomicMax, 8x8 tile per wave", d, W, H);
    run<2, 1>("plain store, 8x8 tile per wave", d, W, H);
    run<3, 0>("LDS atomicMax u64, scattered in 64x64 tile", d, W, H);
    run<3, 1>("LDS atomicMax u64, 8x8 footprint", d, W, H);
    return 0;
}
