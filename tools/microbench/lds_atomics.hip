// Micro-benchmark: what an LDS wave-instruction of the tile kernel's row loop costs on MI355X.
// 512-thread workgroups, 2 per CU (the tile kernel's residency: 16 waves per CU), each lane walks a pixel row of a
// 64 x 65-word LDS tile (lane <-> row, the tile kernel's unit pattern) and issues ONE LDS operation per step:
//   mode 0  ds_max_u64 (no return), every lane a real value         mode 1  the same, half of the lanes merge 0
//   mode 2  ds_max_u64 under an exec mask (half of the lanes off)   mode 3  ds_write_b64
//   mode 4  ds_max_u32                                              mode 5  ds_max_rtn_u64 (value used)
//   mode 6  no LDS operation (the loop's ALU alone)                 mode 7  ds_max_u64, all lanes of a wave in ONE row (32 consecutive words: no conflicts at all)
//   mode 8  ds_max_u64, random rows per lane (bank conflicts as they come)
// Prints ns and shader cycles (2.4 GHz nominal) per wave-instruction per CU.
// build: hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define PITCH 65
template <int MODE>
__global__ __launch_bounds__(512, 4) void k(unsigned long long* out, uint32_t steps, uint32_t reps)
{
    __shared__ unsigned long long tile[64 * PITCH];
    for (uint32_t i = threadIdx.x; i < 64 * PITCH; i += 512) tile[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t row = lane;                                            // lane <-> row
    if (MODE == 7) row = wave * 8u;
    if (MODE == 8) row = (lane * 2654435761u >> 7) & 63u;
    unsigned long long acc = 0;
    unsigned long long v = ((unsigned long long)(0x3F000000u + threadIdx.x) << 32) | threadIdx.x;
    for (uint32_t r = 0; r < reps; r++) {
        unsigned long long* px = tile + row * PITCH + (MODE == 7 ? lane & 31u : (wave & 3u) * 16u);
        uint32_t* px32 = reinterpret_cast<uint32_t*>(px);
        for (uint32_t s = 0; s < steps; s++) {
            v += 0x100000000ull;                                    // (keeps the value live and changing: one VALU op per step)
            if (MODE == 0 || MODE == 7 || MODE == 8) atomicMax(px, v);
            else if (MODE == 1) atomicMax(px, (lane & 1u) ? v : 0ull);
            else if (MODE == 2) { if (lane & 1u) atomicMax(px, v); }
            else if (MODE == 3) *reinterpret_cast<volatile unsigned long long*>(px) = v;
            else if (MODE == 4) atomicMax(px32, (uint32_t)(v >> 32));
            else if (MODE == 5) acc += atomicMax(px, v);
            else acc += v;
            if (MODE != 7) px++; else px += 0;
            px32 += 2;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tile[1] + acc;
    if (MODE >= 5 && acc == 0x1234567ull) out[blockIdx.x + 1] = acc;
}

template <int MODE>
static void run(const char* name, unsigned long long* d, int cus)
{
    const uint32_t blocks = (uint32_t)cus * 2u, steps = 16, reps = 2048;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, steps, 16u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, steps, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waveInstrPerCU = 16.0 * steps * reps;             // 16 waves per CU
    const double ns = ms * 1e6 / waveInstrPerCU;
    std::printf("%-64s %8.3f ms  %7.2f ns = %6.1f cycles per wave-instruction per CU  (%.0f G lane-ops/s chip)\n", name, ms, ns, ns * 2.4,
                (double)blocks * 512 * steps * reps / ms / 1e6);
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned long long* d; hipMalloc(&d, 1 << 20);
    std::printf("%s, %d CUs, 2 x 512-thread workgroups per CU\n", prop.name, cus);
    run<6>("no LDS op (loop ALU only)", d, cus);
    run<0>("ds_max_u64, lane <-> row, all lanes real values", d, cus);
    run<1>("ds_max_u64, half of the lanes merge 0", d, cus);
    run<2>("ds_max_u64, half of the lanes masked off (exec)", d, cus);
    run<3>("ds_write_b64", d, cus);
    run<4>("ds_max_u32", d, cus);
    run<5>("ds_max_rtn_u64 (returned value consumed)", d, cus);
    run<7>("ds_max_u64, a wave's lanes in one row (consecutive words)", d, cus);
    run<8>("ds_max_u64, random row per lane", d, cus);
    return 0;
}
