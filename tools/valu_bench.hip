// VALU issue rates on gfx950 (what an instruction costs in the issue-bound GROUP BY passes): wave64 instructions per clock per CU
// for 32-bit add / xor-shift, 24-bit and 32-bit multiply, and a 64-bit multiply (the partition hash).
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_bench.hip -o tools/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;

template <int OP>
__global__ __launch_bounds__(1024) void k(u64 * out, int iters)
{
    u32 a0 = threadIdx.x * 3 + 1, a1 = threadIdx.x * 5 + 2, a2 = threadIdx.x * 7 + 3, a3 = blockIdx.x + 4;
    u64 b0 = a0 * 11ull + 5, b1 = a1 * 13ull + 6, b2 = a2 * 17ull + 7, b3 = a3 * 19ull + 8;
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int r = 0; r < 8; ++r)
        {
            if (OP == 0) { a0 += a1; a1 += a2; a2 += a3; a3 += a0; }                                   // v_add_u32
            if (OP == 1) { a0 ^= a1 >> 7; a1 ^= a2 >> 9; a2 ^= a3 >> 11; a3 ^= a0 >> 13; }            // shift+xor (2 ops, maybe fused)
            if (OP == 2) { a0 = __umul24(a0, a1) + 1; a1 = __umul24(a1, a2) + 1; a2 = __umul24(a2, a3) + 1; a3 = __umul24(a3, a0) + 1; } // v_mad_u32_u24
            if (OP == 3) { a0 = a0 * 0x9E3779B1u + a1; a1 = a1 * 0x85EBCA6Bu + a2; a2 = a2 * 0xC2B2AE35u + a3; a3 = a3 * 0x27D4EB2Fu + a0; } // v_mul_lo_u32 (+add)
            if (OP == 4) { b0 = b0 * 0x9E3779B97F4A7C15ull + b1; b1 = b1 * 0xC2B2AE3D27D4EB4Full + b2; b2 = b2 * 0x9E3779B97F4A7C15ull + b3; b3 = b3 * 0xC2B2AE3D27D4EB4Full + b0; } // 64-bit multiply
            if (OP == 5) { b0 = (u64)a0 * 0x9E3779B97F4A7C15ull; a0 = (u32)(b0 >> 40) + a1; b1 = (u64)a1 * 0x9E3779B97F4A7C15ull; a1 = (u32)(b1 >> 40) + a2;
                           b2 = (u64)a2 * 0x9E3779B97F4A7C15ull; a2 = (u32)(b2 >> 40) + a3; b3 = (u64)a3 * 0x9E3779B97F4A7C15ull; a3 = (u32)(b3 >> 40) + a0; } // u32 key x 64-bit const, top bits
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3;
}

template <int OP>
void run(const char * name, u64 * out, int cus)
{
    const int iters = 4000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(1024), 0, 0, out, 10);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(1024), 0, 0, out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    const double wave_ops = (double)iters * 8 * 4 * 16;   // source-level operations x waves per CU
    const double clk = 2.4e9 * ms * 1e-3;
    printf("%-52s %.3f ms  %.3f wave64 source-ops/clk/CU  = %.1f clk per op per CU\n", name, ms, wave_ops / clk, clk / wave_ops);
}

int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    u64 * out; (void)hipMalloc(&out, (size_t)cus * 1024 * 8);
    run<0>("add u32", out, cus);
    run<1>("xor + shift u32", out, cus);
    run<2>("mad u24", out, cus);
    run<3>("mul_lo u32 (+add)", out, cus);
    run<4>("mul u64 x const (+add)", out, cus);
    run<5>("u32 key x 64-bit const, bits 40.. (+add)", out, cus);
    return 0;
}
