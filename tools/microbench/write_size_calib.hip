// Calibration of rocprofv3's WRITE_SIZE (and FETCH_SIZE) on gfx950 against KNOWN byte counts, in the store patterns the
// raster kernels use (MI355X_MICROARCH.md, HBM: "WRITE_SIZE is uncalibrated for partial-line stores: calibrate on a known
// byte count in your own access pattern").  Every kernel writes exactly BYTES useful bytes into a 2 GiB buffer (beyond the
// 256 MiB Infinity Cache), in a different shape:
//   k_full16     16 B per lane, wave-contiguous (1 KiB per wave instruction): the tile kernel's tile-out
//   k_full8      8 B per lane, wave-contiguous (512 B per wave instruction)
//   k_block      the setup kernel's pixel-block stores (round 2 form): per wave instruction 4 runs of 10 lanes x 8 B, the runs
//                contiguous in memory (320 B), the block's first word 8 B past a 16-byte granule; 10 x 10-word blocks
//   k_block16    the same blocks written as 16-byte stores of consecutive word pairs (round 3 form)
//   k_rec32      32-byte records, one per lane: two 16-byte stores per lane, lane stride 32 B (write_record_c)
//   k_rec48      48-byte records: three 16-byte stores per lane, lane stride 48 B (write_record)
//   k_scatter4   one 4-byte store per lane to a pseudo-random word of the buffer (bin entries)
//   k_read16     reads BYTES with 16 B per lane (FETCH_SIZE cross-check: the guide says x2)
// Run under the profiler in two passes:
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out_w -o r -- ./write_size_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_f -o r -- ./write_size_calib
// and compare Counter_Value (KiB) per kernel with the "useful KiB" this program prints (tools/summarize_calib.py).
// build: hipcc --offload-arch=gfx950 -O3 write_size_calib.hip -o write_size_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

static const size_t BUF_BYTES = 2ull << 30;

__global__ __launch_bounds__(256) void k_full16(uint4* buf, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) buf[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ __launch_bounds__(256) void k_full8(uint2* buf, size_t n8)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) buf[i] = make_uint2((uint32_t)i, 1u);
}
// one wave per block of W x H words (W = H = 10: what a cluster of ~8x8 px + guard becomes); block b lives at granule
// offset b * 51 (header + 100 words = 101 words -> 51 granules), payload from word 1
__global__ __launch_bounds__(256) void k_block(unsigned long long* buf, uint32_t blocks)
{
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6, waves = gridDim.x * 4u;
    for (uint32_t b = wave; b < blocks; b += waves) {
        unsigned long long* blk = buf + (size_t)b * 102u;
        if (lane == 0u) blk[0] = 0x1234ull;
        for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t idx = lane + 64u * k, x = idx & 15u, y = idx >> 4;
            if (x < 10u && y < 10u) blk[1u + y * 10u + x] = ((unsigned long long)idx << 32) | b;
        }
    }
}
__global__ __launch_bounds__(256) void k_block16(unsigned long long* buf, uint32_t blocks)
{
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6, waves = gridDim.x * 4u;
    for (uint32_t b = wave; b < blocks; b += waves) {
        ulonglong2* blk = reinterpret_cast<ulonglong2*>(buf + (size_t)b * 102u);
        if (lane < 51u) blk[lane] = make_ulonglong2(((unsigned long long)lane << 32) | b, 7ull);   // {header, w0}, {w1, w2}, ...
    }
}
__global__ __launch_bounds__(256) void k_rec32(uint4* buf, size_t recs)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < recs; i += (size_t)gridDim.x * 256) {
        buf[2 * i] = make_uint4((uint32_t)i, 1u, 2u, 3u); buf[2 * i + 1] = make_uint4(4u, 5u, 6u, 7u);
    }
}
__global__ __launch_bounds__(256) void k_rec48(uint4* buf, size_t recs)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < recs; i += (size_t)gridDim.x * 256) {
        buf[3 * i] = make_uint4((uint32_t)i, 1u, 2u, 3u); buf[3 * i + 1] = make_uint4(4u, 5u, 6u, 7u); buf[3 * i + 2] = make_uint4(8u, 9u, 10u, 11u);
    }
}
__global__ __launch_bounds__(256) void k_scatter4(uint32_t* buf, size_t stores, size_t words)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < stores; i += (size_t)gridDim.x * 256) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        buf[h % words] = (uint32_t)i;
    }
}
__global__ __launch_bounds__(256) void k_read16(const uint4* buf, size_t n16, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = buf[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    void* buf; uint32_t* sink;
    if (hipMalloc(&buf, BUF_BYTES) != hipSuccess || hipMalloc((void**)&sink, 256) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, BUF_BYTES);
    (void)hipDeviceSynchronize();
    const size_t BYTES = 1ull << 30;                       // useful bytes per kernel (the block kernels: 20 M blocks x 808 B)
    const uint32_t blocks = 1u << 20, nblk = 1300000u;     // 1.3 M blocks x 816 B stride = 1.06 GB span
    hipLaunchKernelGGL(k_full16, dim3(blocks >> 6), dim3(256), 0, 0, (uint4*)buf, BYTES / 16);
    hipLaunchKernelGGL(k_full8, dim3(blocks >> 6), dim3(256), 0, 0, (uint2*)buf, BYTES / 8);
    hipLaunchKernelGGL(k_block, dim3(4096), dim3(256), 0, 0, (unsigned long long*)buf, nblk);
    hipLaunchKernelGGL(k_block16, dim3(4096), dim3(256), 0, 0, (unsigned long long*)buf, nblk);
    hipLaunchKernelGGL(k_rec32, dim3(blocks >> 6), dim3(256), 0, 0, (uint4*)buf, BYTES / 32);
    hipLaunchKernelGGL(k_rec48, dim3(blocks >> 6), dim3(256), 0, 0, (uint4*)buf, BYTES / 48);
    hipLaunchKernelGGL(k_scatter4, dim3(blocks >> 6), dim3(256), 0, 0, (uint32_t*)buf, (size_t)(16u << 20), BUF_BYTES / 4);
    hipLaunchKernelGGL(k_read16, dim3(blocks >> 6), dim3(256), 0, 0, (const uint4*)buf, BYTES / 16, sink);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 1; }
    std::printf("useful KiB written / read per kernel:\n");
    std::printf("k_full16   %zu\n", BYTES / 1024);
    std::printf("k_full8    %zu\n", BYTES / 1024);
    std::printf("k_block    %zu\n", (size_t)nblk * 808 / 1024);
    std::printf("k_block16  %zu\n", (size_t)nblk * 816 / 1024);
    std::printf("k_rec32    %zu\n", BYTES / 1024);
    std::printf("k_rec48    %zu\n", (BYTES / 48) * 48 / 1024);
    std::printf("k_scatter4 %zu\n", (size_t)(16u << 20) * 4 / 1024);
    std::printf("k_read16   %zu (read)\n", BYTES / 1024);
    return 0;
}
