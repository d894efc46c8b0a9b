// Micro-benchmark: what a kernel boundary costs on MI355X, against the two ways of avoiding one.
//   A  back-to-back launches of a 1-block kernel that does (almost) nothing          -> stream time per launch
//   B  the same with a 256-block x 256-thread kernel                                   -> ... with a real grid
//   C  "last block finishes the job": B blocks each draw a ticket with one returning atomicAdd on ONE word; the last
//      one runs the epilogue.  Cost = kernel time with the ticket - kernel time without, for B = 64 ... 2048
//   D  the same ticket preceded by an agent-scope release fence (__threadfence) in every block
// Each kernel writes 256 B per block first, so that there is something in flight for a fence to order.
// build: hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_empty(uint32_t* out) { if (threadIdx.x == 0 && blockIdx.x == 0xFFFFFFu) out[0] = 1; }

template <int MODE>   // 0: no ticket, 1: ticket, 2: fence + ticket
__global__ __launch_bounds__(256) void k_ticket(uint32_t* buf, uint32_t* ticket, uint32_t* result)
{
    buf[(size_t)blockIdx.x * 64 + (threadIdx.x & 63)] = threadIdx.x + blockIdx.x;
    if (MODE == 0) return;
    __shared__ uint32_t last;
    if (MODE == 2) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1u;
    __syncthreads();
    if (last) {
        if (threadIdx.x == 0) { *ticket = 0; result[0] += 1; }
    }
}

static float time_launches(hipStream_t s, int n, void (*launch)(hipStream_t, void*), void* ctx)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 50; i++) launch(s, ctx);
    (void)hipStreamSynchronize(s);
    (void)hipEventRecord(a, s);
    for (int i = 0; i < n; i++) launch(s, ctx);
    (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / n;
}

struct Ctx { uint32_t* buf; uint32_t* ticket; uint32_t* result; uint32_t blocks; };

int main()
{
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    Ctx c; c.blocks = 1;
    (void)hipMalloc(&c.buf, (size_t)4096 * 256); (void)hipMalloc(&c.ticket, 256); (void)hipMalloc(&c.result, 256);
    (void)hipMemset(c.ticket, 0, 256); (void)hipMemset(c.result, 0, 256);
    const int N = 2000;
    float us = time_launches(s, N, [](hipStream_t st, void* p) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, ((Ctx*)p)->buf); }, &c);
    std::printf("A  1 block x 64 threads, empty, back to back          %6.2f us per launch\n", us);
    us = time_launches(s, N, [](hipStream_t st, void* p) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, ((Ctx*)p)->buf); }, &c);
    std::printf("B  256 blocks x 256 threads, empty, back to back      %6.2f us per launch\n", us);
    us = time_launches(s, N, [](hipStream_t st, void* p) { hipLaunchKernelGGL(k_empty, dim3(2040), dim3(512), 0, st, ((Ctx*)p)->buf); }, &c);
    std::printf("B' 2040 blocks x 512 threads, empty, back to back     %6.2f us per launch\n", us);
    for (uint32_t blocks : {64u, 128u, 256u, 512u, 1088u, 2048u}) {
        c.blocks = blocks;
        const float t0 = time_launches(s, N, [](hipStream_t st, void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(k_ticket<0>, dim3(c->blocks), dim3(256), 0, st, c->buf, c->ticket, c->result); }, &c);
        const float t1 = time_launches(s, N, [](hipStream_t st, void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(k_ticket<1>, dim3(c->blocks), dim3(256), 0, st, c->buf, c->ticket, c->result); }, &c);
        const float t2 = time_launches(s, N, [](hipStream_t st, void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(k_ticket<2>, dim3(c->blocks), dim3(256), 0, st, c->buf, c->ticket, c->result); }, &c);
        std::printf("C/D %4u blocks: plain %6.2f us   + ticket %6.2f us (%+.2f)   + fence and ticket %6.2f us (%+.2f)\n", blocks, t0, t1, t1 - t0, t2, t2 - t0);
    }
    return 0;
}
