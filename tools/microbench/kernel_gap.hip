// Micro-benchmark: what does the boundary between two DEPENDENT kernels cost as a function of the bytes the first one wrote, and of
// HOW it wrote them?  (The frame's timeline shows ~5.5 us between most dependent launches and 14.5 us behind the first pass's tile
// kernel, which writes the 66 MB image; an empty kernel follows an empty kernel after 2.6 us.  MI355X has one L2 per XCD, not
// coherent with each other: a kernel's dirty lines are written back when it ends.)
//   W<MODE>  2048 blocks x 256 threads (all resident); a block stores its share of SIZE bytes in eight slices with 2 us of spinning
//            behind each slice -- a kernel that is not bound by its stores, like the raster kernels.
//            MODE 0 plain 16-byte stores | 1 __builtin_nontemporal_store | 2 relaxed atomic stores, system scope (write-through: sc0 sc1)
//            | 3 relaxed atomic stores, agent scope | 4 plain stores + __builtin_amdgcn_global_wb-like fence at the end of every block (release, agent scope)
//   R        64 blocks x 256 threads, reads 64 KB of what W wrote
// Printed: us per (W, R) pair over 300 pairs on one stream, and us per W back to back.
// build: hipcc --offload-arch=gfx950 -O3 kernel_gap.hip -o kernel_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k_write(unsigned long long* buf, uint32_t wordsPerBlockSlice, uint32_t spinTicks, uint32_t seed)
{
    // words: 8-byte words; a thread stores pairs (16 bytes) at consecutive addresses across the block
    unsigned long long* base = buf + (size_t)blockIdx.x * wordsPerBlockSlice * 8u;
    for (uint32_t s = 0; s < 8u; s++) {
        unsigned long long* p = base + (size_t)s * wordsPerBlockSlice;
        for (uint32_t i = threadIdx.x * 2u; i + 1u < wordsPerBlockSlice; i += 512u) {
            const unsigned long long a = ((unsigned long long)seed << 32) | i, b = a + 1ull;
            if (MODE == 0 || MODE == 4) { *reinterpret_cast<ulonglong2*>(p + i) = make_ulonglong2(a, b); }
            else if (MODE == 1) { __builtin_nontemporal_store(a, p + i); __builtin_nontemporal_store(b, p + i + 1); }
            else if (MODE == 2) { __hip_atomic_store(p + i, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(p + i + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            else { __hip_atomic_store(p + i, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(p + i + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        const uint64_t t0 = wall_clock64();
        while (wall_clock64() - t0 < spinTicks) { }
    }
    if (MODE == 4) __atomic_thread_fence(__ATOMIC_RELEASE);     // (agent scope by default in HIP: s_waitcnt + buffer_wbl2 sc1)
}

__global__ __launch_bounds__(256) void k_read(const unsigned long long* buf, unsigned long long* out)
{
    const unsigned long long v = buf[(size_t)blockIdx.x * 128u + (threadIdx.x & 127u)];
    if (v == 0x123456789ull) out[0] = v;
}

template <int MODE>
static void measure(hipStream_t s, unsigned long long* buf, unsigned long long* out, size_t bytes, const char* name)
{
    const uint32_t blocks = 2048;
    const uint32_t wordsPerSlice = (uint32_t)(bytes / 8 / blocks / 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float pair = 0, solo = 0;
    for (int form = 0; form < 2; form++) {
        const int N = 300;
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1) { (void)hipStreamSynchronize(s); (void)hipEventRecord(a, s); }
            for (int i = 0; i < (pass ? N : 30); i++) {
                hipLaunchKernelGGL(k_write<MODE>, dim3(blocks), dim3(256), 0, s, buf, wordsPerSlice, 200u, (uint32_t)i);
                if (form == 0) hipLaunchKernelGGL(k_read, dim3(64), dim3(256), 0, s, buf, out);
            }
            if (pass == 1) { (void)hipEventRecord(b, s); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); (form == 0 ? pair : solo) = ms * 1e3f / N; }
        }
    }
    std::printf("  %-44s %8.2f us per (W, R) pair   %8.2f us per W back to back\n", name, pair, solo);
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long *buf, *out;
    (void)hipMalloc(&buf, (size_t)256 << 20); (void)hipMalloc(&out, 256);
    (void)hipMemset(buf, 0, (size_t)256 << 20);
    std::printf("W spins 8 x 2 us per block (16 us without stores); an empty kernel behind an empty kernel: ~2.6 us\n");
    for (size_t mb : {0, 1, 4, 16, 32, 64, 128}) {
        std::printf("W writes %zu MB:\n", mb);
        const size_t bytes = mb << 20;
        measure<0>(s, buf, out, bytes, "plain 16-byte stores");
        if (!mb) continue;
        measure<1>(s, buf, out, bytes, "nontemporal stores");
        measure<2>(s, buf, out, bytes, "atomic stores, system scope (write-through)");
        measure<3>(s, buf, out, bytes, "atomic stores, agent scope");
        measure<4>(s, buf, out, bytes, "plain stores + release fence per block");
    }
    return 0;
}
