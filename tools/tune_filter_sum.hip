// tune_filter_sum.hip — A/B harness for the fused filter+sum kernel's launch geometry and load flavour (one process,
// interleaved rounds, median + min per variant: cdna_hip_programming.md §5.4 rule 24).  Not part of the product library.
// build: hipcc -O3 --offload-arch=gfx950 tools/tune_filter_sum.hip -o tools/tune_filter_sum ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
struct alignas(16) V2 { i64 v[2]; };

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(i64 * a, u64 n)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
    {
        u64 x = i * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        a[i] = (i64)(x & 0x7FFFFFFFull);
    }
}

struct IntRangePred
{
    u64 lo, span, flip; u32 invert;
    __device__ __forceinline__ bool operator()(i64 a) const { u64 key = (u64)a ^ flip; return ((key - lo) <= span) != (invert != 0); }
};
__device__ IntRangePred g_dummy;

template <bool NT> __device__ __forceinline__ V2 ld(const V2 * p)
{
    if constexpr (NT) { u32x4 r = __builtin_nontemporal_load((const u32x4 *)p); return __builtin_bit_cast(V2, r); }
    else return *p;
}

__device__ __forceinline__ u64 shfl_down_u64(u64 v, int d)
{
    u32 lo = __shfl_down((u32)v, d, 64), hi = __shfl_down((u32)(v >> 32), d, 64);
    return ((u64)hi << 32) | lo;
}

template <int THREADS, int UNROLL, bool NT, int MODE>
__global__ __launch_bounds__(THREADS) void k_fs(const i64 * __restrict__ a, u64 n, i64 thr, u64 * __restrict__ ps, u64 * __restrict__ pc)
{
    const IntRangePred pr{1ull << 63, (u64)thr - 1, 1ull << 63, 0};
    const u64 nvec = n / 2;
    const V2 * __restrict__ pv = (const V2 *)a;
    u64 s = 0, c = 0;
    u64 i, end, stride;
    if (MODE == 3)
    {
        const u64 chunk = (u64)UNROLL * THREADS;
        const u64 n_chunks = nvec / chunk;
        V2 xa[UNROLL], xb[UNROLL];
        u64 ch = blockIdx.x;
        if (ch < n_chunks)
        {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) xa[k] = ld<NT>(&pv[ch * chunk + (u64)k * THREADS + threadIdx.x]);
        }
        while (ch < n_chunks)
        {
            const u64 nx = ch + gridDim.x;
            if (nx < n_chunks)
            {
#pragma unroll
                for (int k = 0; k < UNROLL; ++k) xb[k] = ld<NT>(&pv[nx * chunk + (u64)k * THREADS + threadIdx.x]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < UNROLL; ++k)
#pragma unroll
                for (int e = 0; e < 2; ++e) { bool p = pr(xa[k].v[e]); s += p ? (u64)xa[k].v[e] : 0; c += p ? 1 : 0; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) xa[k] = xb[k];
            ch = nx;
        }
        i = n_chunks * chunk + (u64)blockIdx.x * THREADS + threadIdx.x;
        stride = (u64)gridDim.x * THREADS;
        for (; i < nvec; i += stride)
        {
            V2 x = pv[i];
            for (int e = 0; e < 2; ++e) { bool p = pr(x.v[e]); s += p ? (u64)x.v[e] : 0; c += p ? 1 : 0; }
        }
        end = 0; i = 0; stride = 1;
    }
    else if (MODE == 2)
    {
        // chunked: per iteration a workgroup reads UNROLL*THREADS consecutive vectors (32 KiB at T256/U8)
        const u64 chunk = (u64)UNROLL * THREADS;
        const u64 n_chunks = nvec / chunk;
        for (u64 ch = blockIdx.x; ch < n_chunks; ch += gridDim.x)
        {
            V2 x[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) x[k] = ld<NT>(&pv[ch * chunk + (u64)k * THREADS + threadIdx.x]);
#pragma unroll
            for (int k = 0; k < UNROLL; ++k)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                {
                    bool p = pr(x[k].v[e]);
                    s += p ? (u64)x[k].v[e] : 0;
                    c += p ? 1 : 0;
                }
        }
        i = n_chunks * chunk + (u64)blockIdx.x * THREADS + threadIdx.x;
        stride = (u64)gridDim.x * THREADS;
        end = nvec;
        for (; i < end; i += stride)
        {
            V2 x = pv[i];
            for (int e = 0; e < 2; ++e) { bool p = pr(x.v[e]); s += p ? (u64)x.v[e] : 0; c += p ? 1 : 0; }
        }
        end = 0; i = 0; stride = 1;
    }
    else if (MODE == 1)
    {
        // each workgroup owns one contiguous span; lanes interleave inside it
        const u64 per = (nvec + gridDim.x - 1) / gridDim.x;
        const u64 b0 = (u64)blockIdx.x * per;
        end = b0 + per < nvec ? b0 + per : nvec;
        i = b0 + threadIdx.x;
        stride = THREADS;
    }
    else
    {
        i = (u64)blockIdx.x * THREADS + threadIdx.x;
        stride = (u64)gridDim.x * THREADS;
        end = nvec;
    }
    for (; i + (UNROLL - 1) * stride < end; i += UNROLL * stride)
    {
        V2 x[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) x[k] = ld<NT>(&pv[i + k * stride]);
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
#pragma unroll
            for (int e = 0; e < 2; ++e)
            {
                bool p = pr(x[k].v[e]);
                s += p ? (u64)x[k].v[e] : 0;
                c += p ? 1 : 0;
            }
    }
    for (; i < end; i += stride)
    {
        V2 x = pv[i];
        for (int e = 0; e < 2; ++e) { bool p = pr(x.v[e]); s += p ? (u64)x.v[e] : 0; c += p ? 1 : 0; }
    }
    for (int d = 32; d >= 1; d >>= 1) { s += shfl_down_u64(s, d); c += shfl_down_u64(c, d); }
    __shared__ u64 ls[THREADS / 64], lc[THREADS / 64];
    if ((threadIdx.x & 63) == 0) { ls[threadIdx.x >> 6] = s; lc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        u64 ss = 0, cc = 0;
        for (int w = 0; w < THREADS / 64; ++w) { ss += ls[w]; cc += lc[w]; }
        ps[blockIdx.x] = ss; pc[blockIdx.x] = cc;
    }
}

struct Variant { std::string name; void (*launch)(const i64 *, u64, i64, u64 *, u64 *, int grid, hipStream_t); int grid; std::vector<float> ms; };

template <int T, int U, bool NT, int B>
void launch(const i64 * a, u64 n, i64 thr, u64 * ps, u64 * pc, int grid, hipStream_t st)
{
    hipLaunchKernelGGL((k_fs<T, U, NT, B>), dim3(grid), dim3(T), 0, st, a, n, thr, ps, pc);
}

int main(int argc, char ** argv)
{
    u64 n = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000000ull;
    int rounds = argc > 2 ? atoi(argv[2]) : 9;
    i64 * a; u64 *ps, *pc;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&ps, 1 << 20)); CK(hipMalloc(&pc, 1 << 20));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, n);
    CK(hipDeviceSynchronize());
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    std::vector<Variant> vs;
#define ADD(T, U, NT, B) for (int bpc : {1, 2, 3, 4}) { int waves = bpc * (T / 64); if (waves > 32) continue; \
        vs.push_back({std::string("T") + #T + "_U" + #U + (NT ? "_nt" : "_pl") + (B == 3 ? "_dbuf" : B == 2 ? "_chk" : B ? "_blk" : "_gs") + "_b" + std::to_string(bpc), launch<T, U, NT, B>, cus * bpc, {}}); }
    ADD(256, 2, true, 2) ADD(256, 3, true, 2) ADD(256, 4, true, 2) ADD(256, 6, true, 2) ADD(256, 8, true, 2)
    ADD(256, 2, true, 3) ADD(256, 3, true, 3) ADD(256, 4, true, 3) ADD(256, 6, true, 3) ADD(256, 8, true, 3)
    ADD(512, 2, true, 2) ADD(512, 3, true, 2) ADD(512, 2, true, 3) ADD(512, 4, true, 3) ADD(128, 8, true, 3) ADD(128, 16, true, 2)
    ADD(256, 4, true, 0) ADD(256, 8, true, 0)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    i64 thr = 214748365;
    for (int r = 0; r < rounds + 1; ++r)
        for (auto & v : vs)
        {
            CK(hipEventRecord(e0, 0));
            v.launch(a, n, thr, ps, pc, v.grid, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms);
        }
    std::sort(vs.begin(), vs.end(), [](const Variant & x, const Variant & y) {
        auto med = [](std::vector<float> m) { std::sort(m.begin(), m.end()); return m[m.size() / 2]; };
        return med(x.ms) < med(y.ms); });
    printf("%-28s %8s %8s %9s\n", "variant", "med_ms", "min_ms", "GB/s(med)");
    for (auto & v : vs)
    {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2];
        printf("%-28s %8.4f %8.4f %9.1f\n", v.name.c_str(), med, v.ms[0], n * 8.0 / (med * 1e-3) / 1e9);
    }
    return 0;
}
