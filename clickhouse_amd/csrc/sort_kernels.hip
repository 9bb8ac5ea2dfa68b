// sort_kernels.hip — SURVEY §8(f) rank 4: ordered output.  IColumn::getPermutation for ColumnVector<T>
// (src/Columns/ColumnVector.cpp:245-330) — the step under sortBlock / MergeSortingTransform / PartialSortingTransform
// (src/Interpreters/sortBlock.cpp, src/Processors/Transforms/MergeSortingTransform.cpp) — as an LSD radix sort on the device.
//
// The reference already radix-sorts here (RadixSort<RadixSortTraits<T>>::executeLSD, ColumnVector.cpp:294-316) but only for the
// cases its CPU radix sort handles (not stable-descending, not stable floats) and falls back to comparison sorts otherwise;
// this path is one algorithm for every case, with the STABLE semantics of less_stable / greater_stable (:120-166): equal values —
// including -0.0 == 0.0 and NaN with NaN — keep their original order, in both directions.  NaN placement follows
// CompareHelper (nan_direction_hint > 0: NaN greater than every number, < 0: smaller; ColumnVector.h FloatCompareHelper).
//   k_sort_keys     value -> order-preserving unsigned key of the same width (sign flip; IEEE total-order fold with -0.0 and NaN
//                   canonicalised; bitwise complement for descending), optionally gathered through an incoming permutation
//                   (ORDER BY a, b = sort by b, then stably by a), and the initial permutation
//   chgpu_partition_by_key_byte (partition_kernels.hip): LDS-staged stable 256-way split of (key, permutation) by one key byte per
//                   pass; the digit is read straight from the key column, no selector column, no host synchronisation
// Algorithmic bytes: per pass (sizeof(key) + 8) read and written, sizeof(T) passes.
#include "chgpu_internal.h"

template <typename T>
struct SortKey;
template <> struct SortKey<u8> { typedef u8 K; static __device__ K key(u8 v, int) { return v; } };
template <> struct SortKey<u16> { typedef u16 K; static __device__ K key(u16 v, int) { return v; } };
template <> struct SortKey<u32> { typedef u32 K; static __device__ K key(u32 v, int) { return v; } };
template <> struct SortKey<u64> { typedef u64 K; static __device__ K key(u64 v, int) { return v; } };
template <> struct SortKey<i8> { typedef u8 K; static __device__ K key(i8 v, int) { return (u8)v ^ 0x80u; } };
template <> struct SortKey<i16> { typedef u16 K; static __device__ K key(i16 v, int) { return (u16)v ^ 0x8000u; } };
template <> struct SortKey<i32> { typedef u32 K; static __device__ K key(i32 v, int) { return (u32)v ^ 0x80000000u; } };
template <> struct SortKey<i64> { typedef u64 K; static __device__ K key(i64 v, int) { return (u64)v ^ 0x8000000000000000ull; } };
template <> struct SortKey<float>
{
    typedef u32 K;
    static __device__ K key(float v, int nan_hint)
    {
        if (v != v)
            return nan_hint > 0 ? 0xFFFFFFFFu : 0u;
        if (v == 0.0f)
            v = 0.0f; // -0.0 == 0.0 (less_stable compares with ==)
        const u32 b = __float_as_uint(v);
        const u32 k = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        // keep the extremes free for NaN: -inf folds to 0x007FFFFF, +inf to 0xFF800000 -- neither is 0 nor all-ones
        return k;
    }
};
template <> struct SortKey<double>
{
    typedef u64 K;
    static __device__ K key(double v, int nan_hint)
    {
        if (v != v)
            return nan_hint > 0 ? ~0ull : 0ull;
        if (v == 0.0)
            v = 0.0;
        const u64 b = (u64)__double_as_longlong(v);
        return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
    }
};

template <typename T>
__global__ __launch_bounds__(256) void k_sort_keys(const T * __restrict__ data, u64 rows, const u64 * __restrict__ perm_in, u64 n, int descending, int nan_hint,
                                                   typename SortKey<T>::K * __restrict__ keys, u64 * __restrict__ perm)
{
    typedef typename SortKey<T>::K K;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        u64 src = perm_in ? perm_in[i] : i;
        if (src >= rows)
            src = 0; // a caller bug in the reference too (no bounds check in IColumn::permute); no fault here
        K k = SortKey<T>::key(data[src], nan_hint);
        keys[i] = descending ? (K)~k : k;
        perm[i] = src;
    }
}

template <typename T>
static int sort_impl(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * perm_in, int descending, int nan_hint, chgpu_col ** perm_out)
{
    typedef typename SortKey<T>::K K;
    const u64 n = perm_in ? perm_in->rows : col->rows;
    constexpr int key_type = sizeof(K) == 8 ? CHGPU_U64 : sizeof(K) == 4 ? CHGPU_U32 : sizeof(K) == 2 ? CHGPU_U16 : CHGPU_U8;
    chgpu_col * keys = nullptr, * perm = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, key_type, n, &keys));
    int rc = chgpu_col_new(ctx, CHGPU_U64, n, &perm);
    if (rc == CHGPU_OK && n)
    {
        const u32 grid = chgpu_grid_for(ctx, n, 256, 8);
        hipLaunchKernelGGL(k_sort_keys<T>, dim3(grid), dim3(256), 0, ctx->stream, (const T *)col->data, (u64)col->rows,
                           perm_in ? (const u64 *)perm_in->data : nullptr, n, descending, nan_hint, (K *)keys->data, (u64 *)perm->data);
        ctx->counters[6] += 1;
        for (u32 pass = 0; pass < sizeof(K) && rc == CHGPU_OK; ++pass)
        {
            const chgpu_col * in[2] = {keys, perm};
            chgpu_col * out[2] = {nullptr, nullptr};
            rc = chgpu_partition_by_key_byte(ctx, keys, pass * 8, 2, in, out); // no host synchronisation between the passes
            if (rc == CHGPU_OK)
            {
                chgpu_col_free(keys);
                chgpu_col_free(perm);
                keys = out[0], perm = out[1];
            }
        }
    }
    if (keys)
        chgpu_col_free(keys);
    if (rc != CHGPU_OK)
    {
        if (perm)
            chgpu_col_free(perm);
        return rc;
    }
    *perm_out = perm;
    return CHGPU_OK;
}

extern "C" int chgpu_sort_permutation(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * perm_in_u64, int descending, int nan_direction_hint,
                                      chgpu_col ** perm_out_u64)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && col && perm_out_u64, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(!perm_in_u64 || perm_in_u64->type == CHGPU_U64, CHGPU_ERR_BAD_ARGUMENTS, "a permutation is a UInt64 column (IColumn::Permutation)");
    CHGPU_REQUIRE(!perm_in_u64 || perm_in_u64->rows <= col->rows, CHGPU_ERR_SIZES_MISMATCH, "Size of permutation (%llu) is greater than the column (%llu)",
                  (unsigned long long)(perm_in_u64 ? perm_in_u64->rows : 0), (unsigned long long)col->rows);
    switch (col->type)
    {
        case CHGPU_I64: return sort_impl<i64>(ctx, col, perm_in_u64, descending, nan_direction_hint, perm_out_u64);
        case CHGPU_U64: return sort_impl<u64>(ctx, col, perm_in_u64, descending, nan_direction_hint, perm_out_u64);
        case CHGPU_I32: return sort_impl<i32>(ctx, col, perm_in_u64, descending, nan_direction_hint, perm_out_u64);
        case CHGPU_U32: return sort_impl<u32>(ctx, col, perm_in_u64, descending, nan_direction_hint, perm_out_u64);
        case CHGPU_I16: return sort_impl<i16>(ctx, col, perm_in_u64, descending, nan_direction_hint, perm_out_u64);
        case CHGPU_U16: return sort_impl<u16>(ctx, col, perm_in_u64, descending, nan_direction_hint, perm_out_u64);
        case CHGPU_I8: return sort_impl<i8>(ctx, col, perm_in_u64, descending, nan_direction_hint, perm_out_u64);
        case CHGPU_U8: return sort_impl<u8>(ctx, col, perm_in_u64, descending, nan_direction_hint, perm_out_u64);
        case CHGPU_F64: return sort_impl<double>(ctx, col, perm_in_u64, descending, nan_direction_hint, perm_out_u64);
        case CHGPU_F32: return sort_impl<float>(ctx, col, perm_in_u64, descending, nan_direction_hint, perm_out_u64);
        default: return chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "unsupported column type");
    }
}

// ---------------------------------------------------------------------------------------------
// ORDER BY ... LIMIT n (getPermutation with limit, ColumnVector.cpp:254-281: the reference switches to a partial sort).  Here the
// k-th order statistic is bracketed from a sample: a strided sample of the column is sorted, the value at (about four times) the
// limit's quantile becomes a threshold, `col <= t` (>= for descending) names the candidate rows (filterToIndices), and only those are
// radix-sorted.  The result is EXACT: every row not among the candidates is greater than t, hence after at least `limit` candidates in
// the order, and the candidates keep their row order, so ties break as in the full stable sort.  Too few candidates (an unlucky
// sample), too many (a heavily repeated value) or NaNs that sort first fall back to the full sort.
// Traffic: one comparison pass + two mask passes instead of sizeof(T) partition passes -- 1e8 Int64 rows, LIMIT 10: see DESIGN.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_sample_strided(const T * __restrict__ data, u64 n, u64 stride, u64 m, T * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < m; i += (u64)gridDim.x * 256)
    {
        const u64 r = i * stride;
        out[i] = data[r < n ? r : n - 1];
    }
}

extern "C" int chgpu_sort_permutation_limit(chgpu_ctx * ctx, const chgpu_col * col, int descending, int nan_direction_hint, uint64_t limit,
                                            chgpu_col ** perm_out_u64)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && col && perm_out_u64, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    const u64 n = col->rows;
    const bool is_float = chgpu_type_is_float(col->type);
    // NaNs sort last when they compare greater in an ascending order or smaller in a descending one; otherwise they would have to lead
    // the result and no value threshold describes that
    const bool nan_last = !is_float || (descending ? nan_direction_hint < 0 : nan_direction_hint > 0);
    constexpr u64 SAMPLE = 1u << 16;
    auto full = [&]() -> int {
        chgpu_col * p = nullptr;
        CHGPU_TRY(chgpu_sort_permutation(ctx, col, nullptr, descending, nan_direction_hint, &p));
        if (limit && limit < p->rows)
            p->rows = limit; // IColumn::Permutation is simply cut (the buffer keeps its size class)
        *perm_out_u64 = p;
        return CHGPU_OK;
    };
    if (limit == 0 || limit >= n || n < (1u << 20) || limit * 64 > n || !nan_last)
        return full();
    const size_t es = chgpu_type_size(col->type);
    const u64 stride = n / SAMPLE;
    chgpu_col * sample = nullptr, * sperm = nullptr, * mask = nullptr, * ids = nullptr, * cand = nullptr, * cperm = nullptr, * res = nullptr;
    auto cleanup = [&]() {
        for (chgpu_col * c : {sample, sperm, mask, ids, cand, cperm})
            if (c)
                chgpu_col_free(c);
    };
    int rc = chgpu_col_new(ctx, col->type, SAMPLE, &sample);
    if (rc == CHGPU_OK)
    {
        const u32 grid = chgpu_grid_for(ctx, SAMPLE, 256, 8);
        switch (es)
        {
            case 8: hipLaunchKernelGGL(k_sample_strided<u64>, dim3(grid), dim3(256), 0, ctx->stream, (const u64 *)col->data, n, stride, SAMPLE, (u64 *)sample->data); break;
            case 4: hipLaunchKernelGGL(k_sample_strided<u32>, dim3(grid), dim3(256), 0, ctx->stream, (const u32 *)col->data, n, stride, SAMPLE, (u32 *)sample->data); break;
            case 2: hipLaunchKernelGGL(k_sample_strided<u16>, dim3(grid), dim3(256), 0, ctx->stream, (const u16 *)col->data, n, stride, SAMPLE, (u16 *)sample->data); break;
            default: hipLaunchKernelGGL(k_sample_strided<u8>, dim3(grid), dim3(256), 0, ctx->stream, (const u8 *)col->data, n, stride, SAMPLE, (u8 *)sample->data); break;
        }
        ctx->counters[6] += 1;
        rc = chgpu_sort_permutation(ctx, sample, nullptr, descending, nan_direction_hint, &sperm);
    }
    u64 threshold_bits = 0;
    if (rc == CHGPU_OK)
    {
        // rank of the limit inside the sample, with a 4x margin + 32 (the sample's quantiles wobble by ~sqrt(rank))
        u64 rank = (limit * SAMPLE + n - 1) / n;
        rank = rank * 4 + 32;
        if (rank >= SAMPLE)
        {
            cleanup();
            return full();
        }
        u64 srow = 0;
        rc = chgpu_read_back(ctx, (const u64 *)sperm->data + rank, &srow, sizeof(srow));
        if (rc == CHGPU_OK)
            rc = chgpu_read_back(ctx, (const char *)sample->data + srow * es, &threshold_bits, es);
    }
    if (rc == CHGPU_OK && is_float)
    {
        const bool is_nan = col->type == CHGPU_F64 ? ((threshold_bits & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull)
                                                   : ((threshold_bits & 0x7FFFFFFFull) > 0x7F800000ull);
        if (is_nan)
        {
            cleanup();
            return full();
        }
    }
    u64 n_cand = 0;
    if (rc == CHGPU_OK)
    {
        int scalar_type = col->type;
        if (col->type == CHGPU_F32) // Float32 columns compare after their exact widening: hand the threshold over as Float64
        {
            float f;
            memcpy(&f, &threshold_bits, 4);
            const double d = (double)f;
            memcpy(&threshold_bits, &d, 8);
            scalar_type = CHGPU_F64;
        }
        rc = chgpu_cmp_const(ctx, col, descending ? CHGPU_GE : CHGPU_LE, scalar_type, &threshold_bits, &mask);
    }
    if (rc == CHGPU_OK)
        rc = chgpu_filter_to_indices(ctx, mask, &ids, &n_cand);
    if (rc == CHGPU_OK && (n_cand < limit || n_cand > n / 8))
    {
        cleanup();
        return full();
    }
    if (rc == CHGPU_OK)
        rc = chgpu_index(ctx, col, ids, 0, 0, &cand);
    if (rc == CHGPU_OK)
        rc = chgpu_sort_permutation(ctx, cand, nullptr, descending, nan_direction_hint, &cperm);
    if (rc == CHGPU_OK)
    {
        cperm->rows = limit;
        rc = chgpu_index(ctx, ids, cperm, 0, 0, &res); // candidate positions -> row numbers
    }
    cleanup();
    if (rc != CHGPU_OK)
        return rc;
    *perm_out_u64 = res;
    return CHGPU_OK;
}
