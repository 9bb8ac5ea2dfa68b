// scan.hip — device-wide prefix sums (u32 counts -> u64 offsets) used by filter compaction, scatter and join output.
// Reduce-then-scan in three launches: tile sums -> scan of tile sums (one workgroup) -> per-tile scan + offset.
// HBM traffic: 4 B/elem (pass 1) + 4 B read + 8 B write (pass 3); the tile-sum array is n/2048 u64.
#include "chgpu_internal.h"

static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_ITEMS = 8;
static constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ u64 block_reduce_u64(u64 v, u64 * lds /* >= 4 */)
{
    v = wave_reduce_add_u64(v);
    const u32 wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        lds[wave] = v;
    __syncthreads();
    u64 r = 0;
    const u32 n_waves = blockDim.x >> 6;
    for (u32 w = 0; w < n_waves; ++w)
        r += lds[w];
    __syncthreads();
    return r; // every thread
}

// inclusive wave scan (Hillis-Steele over 64 lanes)
__device__ __forceinline__ u64 wave_scan_inclusive_u64(u64 v)
{
    const u32 lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        u32 lo = __shfl_up((u32)v, d, WAVE);
        u32 hi = __shfl_up((u32)(v >> 32), d, WAVE);
        u64 o = ((u64)hi << 32) | lo;
        if (lane >= (u32)d)
            v += o;
    }
    return v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_sums(const u32 * __restrict__ in, u64 n, u64 * __restrict__ tile_sums)
{
    __shared__ u64 lds[4];
    const u64 base = (u64)blockIdx.x * SCAN_TILE;
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
    {
        u64 i = base + (u64)k * SCAN_THREADS + threadIdx.x;
        if (i < n)
            s += in[i];
    }
    s = block_reduce_u64(s, lds);
    if (threadIdx.x == 0)
        tile_sums[blockIdx.x] = s;
}

// one workgroup: exclusive scan of tile_sums in place, total -> *total
__global__ __launch_bounds__(1024) void k_scan_tile_offsets(u64 * __restrict__ tile_sums, u64 n_tiles, u64 * __restrict__ total)
{
    __shared__ u64 wave_tot[16];
    __shared__ u64 carry_s;
    if (threadIdx.x == 0)
        carry_s = 0;
    __syncthreads();
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u64 base = 0; base < n_tiles; base += 1024)
    {
        u64 i = base + threadIdx.x;
        u64 v = i < n_tiles ? tile_sums[i] : 0;
        u64 inc = wave_scan_inclusive_u64(v);
        if (lane == 63)
            wave_tot[wave] = inc;
        __syncthreads();
        u64 woff = 0;
        for (u32 w = 0; w < wave; ++w)
            woff += wave_tot[w];
        u64 carry = carry_s;
        if (i < n_tiles)
            tile_sums[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023)
            carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        *total = carry_s;
}

template <bool INCLUSIVE>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const u32 * __restrict__ in, u64 * __restrict__ out, u64 n,
                                                             const u64 * __restrict__ tile_offsets)
{
    __shared__ u64 wave_tot[4];
    // thread t owns items [t*ITEMS, t*ITEMS+ITEMS) of the tile so its serial scan is over consecutive elements
    const u64 tile0 = (u64)blockIdx.x * SCAN_TILE;
    const u64 base = tile0 + (u64)threadIdx.x * SCAN_ITEMS;
    // Full tiles go through LDS both ways: the tile is read and written with 16-byte accesses on consecutive addresses and only the
    // LDS side sees the thread-owns-eight-consecutive-items pattern (which, straight to memory, made every load / store instruction
    // touch 64 different lines with 4 / 8 bytes each: 0.34 ms per 6e7 elements against the ~0.15 of its 12 B/element)
    static_assert(SCAN_ITEMS == 8 && SCAN_THREADS == 256, "the staging below is written for 8 items x 256 threads");
    __shared__ __attribute__((aligned(16))) u32 s_in[SCAN_TILE];
    __shared__ __attribute__((aligned(16))) u64 s_out[SCAN_TILE];
    const bool staged = tile0 + SCAN_TILE <= n && (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    u32 v[SCAN_ITEMS];
    u64 s = 0;
    if (staged)
    {
        const uint4 * src = (const uint4 *)(in + tile0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            ((uint4 *)s_in)[j * SCAN_THREADS + threadIdx.x] = src[j * SCAN_THREADS + threadIdx.x];
        __syncthreads();
        const uint4 a = ((const uint4 *)s_in)[threadIdx.x * 2], b = ((const uint4 *)s_in)[threadIdx.x * 2 + 1];
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k)
            s += v[k];
    }
    else
    {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k)
        {
            v[k] = (base + k < n) ? in[base + k] : 0;
            s += v[k];
        }
    }
    u64 inc = wave_scan_inclusive_u64(s);
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63)
        wave_tot[wave] = inc;
    __syncthreads();
    u64 off = tile_offsets[blockIdx.x] + inc - s;
    for (u32 w = 0; w < wave; ++w)
        off += wave_tot[w];
    if (staged)
    {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k)
        {
            s_out[threadIdx.x * SCAN_ITEMS + k] = INCLUSIVE ? off + v[k] : off;
            off += v[k];
        }
        __syncthreads();
        typedef u64 v2u64 __attribute__((ext_vector_type(2)));
        v2u64 * dst = (v2u64 *)(out + tile0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            dst[j * SCAN_THREADS + threadIdx.x] = ((const v2u64 *)s_out)[j * SCAN_THREADS + threadIdx.x];
        return;
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
    {
        if (base + k < n)
            out[base + k] = INCLUSIVE ? off + v[k] : off;
        off += v[k];
    }
}

size_t chgpu_scan_tmp_bytes(u64 n)
{
    u64 n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    return (size_t)(n_tiles + 2) * sizeof(u64);
}

template <bool INCLUSIVE>
static int scan_impl(chgpu_ctx * ctx, const u32 * in, u64 * out, u64 n, u64 * total_dev, void * tmp, size_t tmp_bytes)
{
    u64 n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    CHGPU_REQUIRE(tmp_bytes >= chgpu_scan_tmp_bytes(n), CHGPU_ERR_LOGICAL, "scan: temporary buffer too small");
    CHGPU_REQUIRE(n_tiles < (1ull << 31), CHGPU_ERR_NOT_IMPLEMENTED, "scan: too many elements");
    u64 * tile_sums = (u64 *)tmp;
    if (n == 0)
    {
        CHGPU_HIP(hipMemsetAsync(total_dev, 0, sizeof(u64), ctx->stream));
        return CHGPU_OK;
    }
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((u32)n_tiles), dim3(SCAN_THREADS), 0, ctx->stream, in, n, tile_sums);
    hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(1024), 0, ctx->stream, tile_sums, n_tiles, total_dev);
    hipLaunchKernelGGL(k_scan_apply<INCLUSIVE>, dim3((u32)n_tiles), dim3(SCAN_THREADS), 0, ctx->stream, in, out, n, tile_sums);
    ctx->counters[6] += 3;
    CHGPU_HIP(hipGetLastError());
    return CHGPU_OK;
}

int chgpu_scan_exclusive_u32_u64(chgpu_ctx * ctx, const u32 * in, u64 * out, u64 n, u64 * total_dev, void * tmp, size_t tmp_bytes)
{
    return scan_impl<false>(ctx, in, out, n, total_dev, tmp, tmp_bytes);
}

int chgpu_scan_inclusive_u32_u64(chgpu_ctx * ctx, const u32 * in, u64 * out, u64 n, u64 * total_dev, void * tmp, size_t tmp_bytes)
{
    return scan_impl<true>(ctx, in, out, n, total_dev, tmp, tmp_bytes);
}
