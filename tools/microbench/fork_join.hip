// Micro-benchmark: is there anything to win by letting the NEXT frame's group cull run beside the LAST kernel of this frame
// (two streams, fork after the tile schedule of the second raster pass, join in front of the first occlusion cull)?
// The frame is modelled with kernels that spin for a fixed time per workgroup, grids and durations like config 3's
// (profiles/r05_config3_timeline.txt):
//     C  group cull count     90 x 256, 14 us        S  scatter              90 x 256, 4 us
//     H0 occlusion cull       64 x 256, 6 us         P0 setup              1024 x 256, 2 x 9 us
//     B0 clip + large         256 x 256, 3 us        O0 tile schedule         1 x 1024, 3 us
//     T0 tile kernel       2040 x 512, 4 x 16 us     H1 occlusion cull       64 x 256, 9 us
//     P1 setup             1024 x 256, 2 x 4 us      B1, O1                   as above
//     T1 tile kernel       2040 x 512, 4 x 5 us
//   V0  serial, one stream                                   V3  serial + an event recorded between O1 and T1 (what the fork point costs)
//   V1  forked with events: B waits for the fork event of frame i-1 (recorded behind O1), runs C, S, records "cull done";
//       A waits for it in front of H0
//   V2  the same with stream memory operations (hipStreamWriteValue32 behind O1 / behind S, hipStreamWaitValue32) if the runtime has them
//   V4  forked, but the fork event sits behind T1 (no window: the overhead of the two dependencies alone)
//   V5  one stream, the cull's first kernel launched with hipExtAnyOrderLaunch (no barrier bit in its packet): it starts beside the T1 in front of it
// build: hipcc --offload-arch=gfx950 -O3 fork_join.hip -o fork_join
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_spin(uint32_t ticks, uint32_t* sink)      // ticks of the 100 MHz constant clock per workgroup
{
    const uint64_t t0 = wall_clock64();
    uint32_t x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0xDEADBEEFu && sink) sink[0] = x;
}

struct K { uint32_t grid, block, ticks, lds; };     // lds: dynamic LDS bytes (the tile kernel's 64 KB keep it at two workgroups per CU)
static const K C{90, 256, 1400, 0}, S{90, 256, 400, 0}, H0{64, 256, 600, 0}, P0{1024, 256, 900, 16384}, B0{256, 256, 300, 0}, O0{1, 1024, 300, 0},
               T0{2040, 512, 1600, 65536}, H1{64, 256, 900, 0}, P1{1024, 256, 400, 16384}, T1{2040, 512, 500, 65536};

static void go(hipStream_t s, const K& k, uint32_t* sink) { hipLaunchKernelGGL(k_spin, dim3(k.grid), dim3(k.block), k.lds, s, k.ticks, sink); }

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t A, B;
    (void)hipStreamCreateWithFlags(&A, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
    uint32_t* sink; (void)hipMalloc(&sink, 256);
    (void)hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int RING = 8, N = 400;
    hipEvent_t evFork[RING], evDone[RING], t0, t1;
    for (int i = 0; i < RING; i++) { (void)hipEventCreateWithFlags(&evFork[i], hipEventDisableTiming); (void)hipEventCreateWithFlags(&evDone[i], hipEventDisableTiming); }
    (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
    uint32_t* sig = nullptr;
    const bool haveSig = hipExtMallocWithFlags((void**)&sig, 256, hipMallocSignalMemory) == hipSuccess && sig;
    if (haveSig) (void)hipMemset(sig, 0, 256);

    auto rest = [&](bool recordFork, int i, int mode) {      // everything behind the cull on stream A
        go(A, H0, sink); go(A, P0, sink); go(A, B0, sink); go(A, O0, sink); go(A, T0, sink);
        go(A, H1, sink); go(A, P1, sink); go(A, B0, sink); go(A, O0, sink);
        if (recordFork && mode == 1) (void)hipEventRecord(evFork[i % RING], A);
        if (recordFork && mode == 2) (void)hipStreamWriteValue32(A, sig, (uint32_t)(i + 1), 0);
        go(A, T1, sink);
        if (recordFork && mode == 4) (void)hipEventRecord(evFork[i % RING], A);
    };
    auto run = [&](int mode) -> float {
        // (frame numbers keep growing across the warm-up and the timed region: the signal words only ever increase)
        static int serial = 0;
        float ms = 0;
        for (int pass = 0; pass < 2; pass++) {
            const int n = pass == 0 ? 50 : N;
            if (pass == 1) { (void)hipDeviceSynchronize(); (void)hipEventRecord(t0, A); }
            for (int k = 0; k < n; k++, serial++) {
                const int i = serial;
                if (mode == 5) {
                    // one stream; the cull's first kernel goes out WITHOUT the barrier bit (hipExtAnyOrderLaunch): it may start while
                    // the kernel in front of it -- the last frame's T1 -- runs; S, with the bit, waits for both
                    hipExtLaunchKernelGGL(k_spin, dim3(C.grid), dim3(C.block), C.lds, A, nullptr, nullptr, k > 0 ? 1u : 0u, C.ticks, sink);
                    go(A, S, sink);
                    rest(false, i, 0);
                } else if (mode == 0 || mode == 3) {
                    go(A, C, sink); go(A, S, sink);
                    rest(mode == 3, i, mode == 3 ? 1 : 0);
                } else if (mode == 1 || mode == 4) {
                    if (k > 0) (void)hipStreamWaitEvent(B, evFork[(i - 1) % RING], 0);
                    go(B, C, sink); go(B, S, sink);
                    (void)hipEventRecord(evDone[i % RING], B);
                    (void)hipStreamWaitEvent(A, evDone[i % RING], 0);
                    rest(true, i, mode);
                } else {
                    if (k > 0) (void)hipStreamWaitValue32(B, sig, (uint32_t)i, hipStreamWaitValueGte, 0xFFFFFFFFu);
                    go(B, C, sink); go(B, S, sink);
                    (void)hipStreamWriteValue32(B, sig + 16, (uint32_t)(i + 1), 0);
                    (void)hipStreamWaitValue32(A, sig + 16, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFu);
                    rest(true, i, 2);
                }
            }
            if (pass == 1) { (void)hipStreamSynchronize(B); (void)hipEventRecord(t1, A); (void)hipEventSynchronize(t1); (void)hipEventElapsedTime(&ms, t0, t1); }
        }
        (void)hipDeviceSynchronize();
        return ms * 1e3f / N;
    };
    const float sum = (C.ticks + S.ticks + H0.ticks + 2 * P0.ticks + 2 * B0.ticks + 2 * O0.ticks + 4 * T0.ticks + H1.ticks + 2 * P1.ticks + 4 * T1.ticks) / 100.0f;
    std::printf("modelled frame: 12 kernels, %.0f us of spinning if every grid ran at 2 workgroups per CU (C + S = %.0f us of it)\n", sum, (C.ticks + S.ticks) / 100.0f);
    for (int rep = 0; rep < 2; rep++) {
        std::printf("V0 serial, one stream                                  %7.2f us per frame\n", run(0));
        std::printf("V3 serial + event record in front of the last kernel   %7.2f us per frame\n", run(3));
        std::printf("V1 forked behind the last tile schedule (events)       %7.2f us per frame\n", run(1));
        std::printf("V4 forked behind the last kernel (events, no window)   %7.2f us per frame\n", run(4));
        std::printf("V5 one stream, C launched without the barrier bit      %7.2f us per frame\n", run(5));
        if (haveSig) std::printf("V2 forked, stream memory operations                    %7.2f us per frame\n", run(2));
        else if (rep == 0) std::printf("V2 (no signal memory: hipExtMallocWithFlags(hipMallocSignalMemory) failed)\n");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) std::printf("last error: %s\n", hipGetErrorString(e));
    return 0;
}
