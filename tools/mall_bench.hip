// Does a partition pass whose output is consumed chunk by chunk stay inside the 256 MiB Infinity Cache?  (Ground truth for DESIGN.md
// §4.3: whether a chunk-blocked GROUP BY -- scatter a chunk, aggregate the chunk, reuse the same scratch -- can beat the three-pass HBM
// floor.)  The traffic of the scatter and aggregate passes is modelled by a coalesced copy (read input chunk, write scratch) followed by a
// streaming reduce of the scratch; "streaming" moves the scratch window through a buffer as large as the input (every byte goes to HBM),
// "blocked" reuses ONE scratch window of the chunk's size.
// build: hipcc -O3 --offload-arch=gfx950 tools/mall_bench.hip -o tools/mall_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned long long u64;

__global__ __launch_bounds__(256) void k_copy(const uint4 * __restrict__ src, uint4 * __restrict__ dst, u64 n16)
{
    const u64 stride = (u64)gridDim.x * 256 * 4;
    for (u64 i = (u64)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride)
    {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256 < n16) v[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256 < n16) dst[i + u * 256] = v[u];
    }
}

__global__ __launch_bounds__(256) void k_reduce(const uint4 * __restrict__ src, u64 n16, u64 * out)
{
    const u64 stride = (u64)gridDim.x * 256 * 4;
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride)
    {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256 < n16)
            {
                const uint4 v = src[i + u * 256];
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
    }
    if (acc == 0x123456789abcull) out[0] = acc;
}

int main(int argc, char ** argv)
{
    const u64 total = argc > 1 ? strtoull(argv[1], nullptr, 10) : 12000000000ull;  // bytes of "rows" (C3: 1e9 rows x 12 B)
    const u64 n16 = total / 16;
    uint4 *src, *tmp;
    u64 * out;
    if (hipMalloc(&src, n16 * 16) != hipSuccess || hipMalloc(&tmp, n16 * 16) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, n16 * 16);
    hipMemset(tmp, 0, n16 * 16);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("total %.1f GB, %d CUs\n", total / 1e9, cus);
    const u64 chunks_mb[] = {12, 24, 48, 96, 192, 384, 1536, 12000};
    for (u64 cmb : chunks_mb)
    {
        const u64 c16 = cmb * 1000000ull / 16;
        if (c16 > n16) continue;
        for (int blocked = 0; blocked < 2; ++blocked)
            for (int what = 0; what < 3; ++what)  // 0: copy + reduce, 1: copy only, 2: reduce of the scratch only
            {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep)
                {
                    hipEventRecord(a);
                    u64 launches = 0;
                    for (u64 off = 0; off + c16 <= n16; off += c16)
                    {
                        uint4 * t = blocked ? tmp : tmp + off;
                        const int grid = (int)((c16 + 1023) / 1024 < (u64)cus * 8 ? (c16 + 1023) / 1024 : (u64)cus * 8);
                        if (what != 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, src + off, t, c16);
                        if (what != 1) hipLaunchKernelGGL(k_reduce, dim3(grid), dim3(256), 0, 0, t, c16, out);
                        ++launches;
                    }
                    hipEventRecord(b);
                    hipEventSynchronize(b);
                    float ms = 0; hipEventElapsedTime(&ms, a, b);
                    if (ms < best) best = ms;
                }
                const u64 moved = (n16 / c16) * c16 * 16;
                const double traffic = what == 0 ? 3.0 * moved : what == 1 ? 2.0 * moved : 1.0 * moved;
                printf("chunk %6llu MB  %-9s %-12s %8.3f ms   L2<->fabric %.2f TB/s\n", (unsigned long long)cmb, blocked ? "blocked" : "streaming",
                       what == 0 ? "copy+reduce" : what == 1 ? "copy" : "reduce", best, traffic / (best * 1e-3) / 1e12);
            }
    }
    return 0;
}
