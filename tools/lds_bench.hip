// LDS random-access throughput on gfx950 (ground truth for the GROUP BY kernels): 1024-thread workgroups, one per CU,
// every lane hits a pseudo-random cell of an 8192-cell LDS table.  Prints lane-operations per clock per CU.
// build: hipcc -O3 --offload-arch=gfx950 tools/lds_bench.hip -o tools/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;
constexpr u32 S = 8192;

template <int OP>
__global__ __launch_bounds__(1024) void k(u64 * out, int iters, u32 mask_bits)
{
    __shared__ u64 t64[S];
    u32 * t32 = (u32 *)t64;
    for (u32 i = threadIdx.x; i < S; i += 1024) t64[i] = 0;
    __syncthreads();
    u32 x = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
    u64 acc = 0;
    const u32 m = (1u << mask_bits) - 1;
    for (int it = 0; it < iters; ++it)
    {
        x = x * 1664525u + 1013904223u;
        const u32 s = (x >> 12) & m;
        if (OP == 0) acc += t32[s];                                   // ds_read_b32
        if (OP == 1) acc += t64[s];                                   // ds_read_b64
        if (OP == 2) atomicAdd(&t32[s], 1u);                          // ds_add_u32 (no return)
        if (OP == 3) atomicAdd(&t64[s], (u64)x);                      // ds_add_u64 (no return)
        if (OP == 4) acc += atomicAdd(&t32[s], 1u);                   // ds_add_rtn_u32
        if (OP == 5) acc += atomicCAS(&t32[s], 0u, x | 1u);           // ds_cmpst_rtn_b32
        if (OP == 6) { acc += t32[s]; atomicAdd(&t64[(s + 7) & m], (u64)x); atomicAdd(&t32[(s + 13) & m], 1u); } // the GROUP BY mix
    }
    if (acc == 0x123456789ull) out[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[1 + blockIdx.x] = t64[5];
}

template <int OP>
void run(const char * name, u64 * out, int cus, u32 bits, int ops_per_iter)
{
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(1024), 0, 0, out, 100, bits);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(1024), 0, 0, out, iters, bits);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double lane_ops = (double)iters * 1024 * ops_per_iter;   // per CU
    const double clk = 2.4e9 * ms * 1e-3;
    printf("%-44s cells=%5u  %.3f ms  %.2f lane-ops/clk/CU (at 2.4 GHz)  %.3e lane-ops/s chip\n", name, 1u << bits, ms, lane_ops / clk, lane_ops * cus / (ms * 1e-3));
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    u64 * out; hipMalloc(&out, (cus + 2) * 8);
    for (u32 bits : {13u, 10u, 4u})
    {
        run<0>("ds_read_b32 random", out, cus, bits, 1);
        run<1>("ds_read_b64 random", out, cus, bits, 1);
        run<2>("ds_add_u32 random (no return)", out, cus, bits, 1);
        run<3>("ds_add_u64 random (no return)", out, cus, bits, 1);
        run<4>("ds_add_rtn_u32 random", out, cus, bits, 1);
        run<5>("ds_cmpst_rtn_b32 random", out, cus, bits, 1);
        run<6>("read b32 + add u64 + add u32 (GROUP BY mix)", out, cus, bits, 3);
    }
    return 0;
}
