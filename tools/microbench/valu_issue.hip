// Micro-benchmark: what a wave64 VALU instruction costs a SIMD of MI355X in issue time, by instruction class and by the number of
// waves resident on the SIMD.  The block kernel of BASELINE config 5 is bound by VALU issue: its roofline is
// SIMDs x clock / (cycles per wave-instruction), and that divisor is what this program measures instead of assuming it.
//   every wave runs LOOPS x 256 instructions of one class on 8 independent registers (no dependent back-to-back pair closer
//   than 8 instructions), times itself with s_memtime (shader clock) and wall_clock64 (100 MHz constant), and the host reports
//   cycles per wave-instruction per SIMD = (cycles of the slowest wave) x ... / (instructions x waves per SIMD)
// grid = 256 CUs x 4 SIMDs x W waves: blocks of 64 x W x 4 threads, one block per CU (launch bounds keep it at one).
// build: hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define REP256(X) REP32(X) REP32(X) REP32(X) REP32(X) REP32(X) REP32(X) REP32(X) REP32(X)

enum { ADD_U32, AND_B32, LSHL_ADD, MUL_F32, FMA_F32, MAD_U24, MUL_LO_U32, CVT_F32_I32, CNDMASK, RCP_F32, MIN_I32, ADD_F32_ALT, DEP_ADD_U32, DEP_FMA_F32, CMP_CNDMASK, NKINDS };
static const char* kNames[NKINDS] = {"v_add_u32", "v_and_b32", "v_lshl_add_u32", "v_mul_f32", "v_fma_f32", "v_mad_u32_u24", "v_mul_lo_u32",
                                     "v_cvt_f32_i32", "v_cndmask_b32 (vcc)", "v_rcp_f32", "v_min_i32", "v_add_f32 / v_add_u32 alternating",
                                     "v_add_u32, ONE dependent chain", "v_fma_f32, ONE dependent chain", "v_cmp_lt_u32 + v_cndmask_b32 pairs"};

template <int KIND>
__global__ __launch_bounds__(1024, 2) void k_issue(uint32_t loops, unsigned long long* out, uint32_t seed)
{
    uint32_t r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = seed + threadIdx.x * 8u + i;
    const uint32_t c = seed | 3u;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (uint32_t l = 0; l < loops; l++) {
#define ONE(i) \
        if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c)); \
        else if (KIND == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(c)); \
        else if (KIND == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(c)); \
        else if (KIND == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c)); \
        else if (KIND == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c)); \
        else if (KIND == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c)); \
        else if (KIND == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c)); \
        else if (KIND == CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[i])); \
        else if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(c)); \
        else if (KIND == DEP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[0]) : "v"(c)); \
        else if (KIND == DEP_FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[0]) : "v"(c)); \
        else if (KIND == CMP_CNDMASK) { if ((i) & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(c)); else asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r[i]), "v"(c) : "vcc"); } \
        else if (KIND == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i])); \
        else if (KIND == MIN_I32) asm volatile("v_min_i32 %0, %0, %1" : "+v"(r[i]) : "v"(c)); \
        else if (KIND == ADD_F32_ALT) { if ((i) & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c)); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c)); }
        REP256(ONE)
#undef ONE
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc ^= r[i];
    if ((threadIdx.x & 63u) == 0u) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[w * 3 + 0] = t1 - t0; out[w * 3 + 1] = w1 - w0; out[w * 3 + 2] = acc;
    }
}

template <int KIND>
static void run(hipStream_t s, unsigned long long* dOut, int cus, double wallHz)
{
    const uint32_t loops = 400;
    for (int wavesPerSimd : {1, 2, 4, 8}) {
        // (8 waves per SIMD: two 1024-thread blocks per CU -- the dispatcher places one on every CU before it doubles up)
        const int blocksPerCu = wavesPerSimd == 8 ? 2 : 1;
        const int threads = 64 * 4 * wavesPerSimd / blocksPerCu, waves = cus * 4 * wavesPerSimd;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(k_issue<KIND>, dim3(cus * blocksPerCu), dim3(threads), 0, s, loops, dOut, 12345u);
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(a, s);
        hipLaunchKernelGGL(k_issue<KIND>, dim3(cus * blocksPerCu), dim3(threads), 0, s, loops, dOut, 12345u);
        (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h((size_t)waves * 3);
        (void)hipMemcpy(h.data(), dOut, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int w = 0; w < waves; w++) { cyc += (double)h[(size_t)w * 3]; wall += (double)h[(size_t)w * 3 + 1]; }
        cyc /= waves; wall /= waves;
        const double insts = (double)loops * 256.0;
        // a wave's own loop: cycles per instruction of ITS stream; per SIMD the W waves share the issue port
        std::printf("%-36s %d wave(s) per SIMD: %6.2f shader-clock cycles per wave-instruction per SIMD (a wave's loop: %.0f cycles, %.1f us; clock %.2f GHz); kernel %.1f us\n",
                    kNames[KIND], wavesPerSimd, cyc / (insts * wavesPerSimd), cyc, wall / wallHz * 1e6, cyc / (wall / wallHz) * 1e-9, ms * 1e3);
    }
}

int main()
{
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    int wallKHz = 0; (void)hipDeviceGetAttribute(&wallKHz, hipDeviceAttributeWallClockRate, 0);
    const int cus = prop.multiProcessorCount;
    std::printf("%s: %d CUs, clockRate %d kHz, wall clock %d kHz\n", prop.name, cus, prop.clockRate, wallKHz);
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long* dOut; (void)hipMalloc(&dOut, (size_t)cus * 32 * 3 * 8);
    const double wallHz = wallKHz > 0 ? wallKHz * 1e3 : 1e8;
    run<ADD_U32>(s, dOut, cus, wallHz); run<AND_B32>(s, dOut, cus, wallHz); run<LSHL_ADD>(s, dOut, cus, wallHz); run<MIN_I32>(s, dOut, cus, wallHz);
    run<MUL_F32>(s, dOut, cus, wallHz); run<FMA_F32>(s, dOut, cus, wallHz); run<ADD_F32_ALT>(s, dOut, cus, wallHz);
    run<MAD_U24>(s, dOut, cus, wallHz); run<CVT_F32_I32>(s, dOut, cus, wallHz); run<CNDMASK>(s, dOut, cus, wallHz);
    run<MUL_LO_U32>(s, dOut, cus, wallHz); run<RCP_F32>(s, dOut, cus, wallHz);
    run<DEP_ADD_U32>(s, dOut, cus, wallHz); run<DEP_FMA_F32>(s, dOut, cus, wallHz); run<CMP_CNDMASK>(s, dOut, cus, wallHz);
    return 0;
}
