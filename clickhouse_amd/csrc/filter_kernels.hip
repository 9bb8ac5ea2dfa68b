// filter_kernels.hip — comparison masks, order-preserving stream compaction, no-key sum/count and the fused
// predicate+sum kernel (configs C1/C2 of BASELINE.json).  gfx950: wave64 ballots, 16-B-per-lane coalesced loads.
//
// Reference loops replaced (file:line in the reference checkout):
//   k_cmp_mask          NumComparisonImpl::vectorConstant        src/Functions/FunctionsComparison.h:204-245
//   k_count_mask        countBytesInFilter                       src/Columns/ColumnsCommon.cpp:31-58
//   k_filter_*          ColumnVector<T>::filter/doFilterAligned   src/Columns/ColumnVector.cpp:559-724
//   k_filter_sum        FilterTransform::doTransform + AggregateFunctionSumData::addMany/Count
//                                                                src/Processors/Transforms/FilterTransform.cpp:136-256,
//                                                                src/AggregateFunctions/AggregateFunctionSum.h:62-103
//   k_index/k_replicate ColumnVector<T>::indexImpl / replicate    src/Columns/ColumnVector.cpp:1121-1143, 879-907
#include "chgpu_internal.h"
#include <vector>

#include <cmath>
#include <cstdlib>
#include <type_traits>

static u32 tune_env(const chgpu_ctx * ctx, const char * name, u32 dflt) { return (u32)chgpu_opt(ctx, name, dflt); } // developer knobs (chgpu_ctx_set_option)

// ---------------------------------------------------------------------------------------------
// predicates
// ---------------------------------------------------------------------------------------------

// Integer comparison against a constant, for any op and any signedness mix, as ONE unsigned range test on an
// order-preserving key: pass = ((key(a) - lo) <= span) != invert.  The host folds accurate::lessOp/equalsOp
// (src/Core/AccurateComparison.h:20-130: mixed-sign operands compare mathematically) into lo/span/invert.
struct IntRangePred
{
    u64 lo, span, flip;
    u32 invert;
    template <typename T>
    __device__ __forceinline__ bool operator()(T a) const
    {
        u64 key;
        if constexpr (std::is_signed<T>::value)
            key = (u64)(i64)a ^ flip; // sign-extend, then flip the sign bit
        else
            key = (u64)a ^ flip;
        return ((key - lo) <= span) != (invert != 0);
    }
};

// Float64 column vs Float64 constant: IEEE semantics are exactly the reference's (any NaN -> false, != -> true)
template <int OP>
struct F64Pred
{
    double s;
    __device__ __forceinline__ bool operator()(double a) const
    {
        if constexpr (OP == CHGPU_EQ) return a == s;
        if constexpr (OP == CHGPU_NE) return a != s;
        if constexpr (OP == CHGPU_LT) return a < s;
        if constexpr (OP == CHGPU_GT) return a > s;
        if constexpr (OP == CHGPU_LE) return a <= s;
        return a >= s;
    }
};

struct TruePred
{
    template <typename T>
    __device__ __forceinline__ bool operator()(T) const { return true; }
};

struct CmpSpec
{
    bool is_f64 = false; // floating-point column (Float64, or Float32 compared after its exact widening to double)
    int op = 0;
    double fs = 0;
    IntRangePred ip{0, 0, 0, 0};
};

// Fold (column type, op, scalar) into a CmpSpec; CHGPU_ERR_NOT_IMPLEMENTED for unsupported type mixes.
static int make_cmp_spec(int col_type, int op, int scalar_type, const void * scalar, CmpSpec * spec)
{
    CHGPU_REQUIRE(op >= CHGPU_EQ && op <= CHGPU_GE, CHGPU_ERR_BAD_ARGUMENTS, "unknown comparison op %d", op);
    CHGPU_REQUIRE(scalar, CHGPU_ERR_BAD_ARGUMENTS, "scalar is NULL");
    CHGPU_REQUIRE(chgpu_type_is_float(col_type) || chgpu_type_is_int(col_type), CHGPU_ERR_BAD_ARGUMENTS, "unknown column type %d", col_type);
    __int128 s = 0;
    switch (scalar_type)
    {
        case CHGPU_I64: s = *(const i64 *)scalar; break;
        case CHGPU_U64: s = *(const u64 *)scalar; break;
        case CHGPU_U32: s = *(const u32 *)scalar; break;
        case CHGPU_I32: s = *(const i32 *)scalar; break;
        case CHGPU_U8: s = *(const u8 *)scalar; break;
        case CHGPU_U16: s = *(const u16 *)scalar; break;
        case CHGPU_I16: s = *(const i16 *)scalar; break;
        case CHGPU_I8: s = *(const i8 *)scalar; break;
        case CHGPU_F64: case CHGPU_F32: break;
        default: return chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "unknown scalar type %d", scalar_type);
    }
    if (chgpu_type_is_float(col_type))
    {
        spec->is_f64 = true;
        spec->op = op;
        if (chgpu_type_is_float(scalar_type))
        {
            spec->fs = scalar_type == CHGPU_F64 ? *(const double *)scalar : (double)*(const float *)scalar;
            return CHGPU_OK;
        }
        // Float64 column vs integer constant s, compared mathematically (accurate::lessOp/equalsOp, AccurateComparison.h:20-130):
        // d_down / d_up = the doubles next to s (equal when s is representable); every double is < s iff it is < d_up, etc.
        double d = (double)s, d_down = d, d_up = d;
        const __int128 back = (__int128)d; // |s| < 2^64: exact
        if (back < s)
            d_up = std::nextafter(d, INFINITY);
        else if (back > s)
            d_down = std::nextafter(d, -INFINITY);
        switch (op)
        {
            case CHGPU_LT: spec->fs = d_up; break;
            case CHGPU_GE: spec->fs = d_up; break;
            case CHGPU_LE: spec->fs = d_down; break;
            case CHGPU_GT: spec->fs = d_down; break;
            case CHGPU_EQ:
                if (back == s)
                    spec->fs = d;
                else
                {
                    spec->op = CHGPU_LT; // no double equals s: `a < -inf` is false for every a, NaN included
                    spec->fs = -INFINITY;
                }
                break;
            case CHGPU_NE:
                spec->fs = back == s ? d : (double)NAN; // `a != NaN` holds for every a
                break;
        }
        return CHGPU_OK;
    }
    if (chgpu_type_is_float(scalar_type))
    {
        // integer column vs Float64 (or Float32) constant c, compared mathematically: fold c into the integer threshold of an equivalent
        // integer comparison (a < c <=> a < ceil(c); a <= c <=> a <= floor(c); a == c needs an integral c); NaN compares
        // false except under != (notEqualsOp = !equalsOp)
        const double c = scalar_type == CHGPU_F64 ? *(const double *)scalar : (double)*(const float *)scalar;
        const __int128 big = (__int128)1 << 65; // beyond both 64-bit domains
        auto to_i128 = [&](double x) -> __int128 { return x >= 0x1p65 ? big : x <= -0x1p65 ? -big : (__int128)x; };
        if (std::isnan(c) || ((op == CHGPU_EQ || op == CHGPU_NE) && (std::isinf(c) || std::floor(c) != c)))
        {
            s = big; // equals nothing
            op = op == CHGPU_NE ? CHGPU_NE : CHGPU_EQ;
        }
        else if (op == CHGPU_LT || op == CHGPU_GE)
            s = to_i128(std::ceil(c));
        else if (op == CHGPU_LE || op == CHGPU_GT)
            s = to_i128(std::floor(c));
        else
            s = to_i128(c);
    }
    const bool sgn = chgpu_type_is_signed(col_type);
    // the 64-bit extension domain of the column's signedness class
    const __int128 dmin = sgn ? -((__int128)1 << 63) : 0;
    const __int128 dmax = sgn ? (((__int128)1 << 63) - 1) : (((__int128)1 << 64) - 1);
    const u64 flip = sgn ? (1ull << 63) : 0;
    auto key = [&](__int128 v) { return (u64)v ^ flip; }; // (u64) of an in-domain value is its two's complement
    __int128 lo = dmin, hi = dmax;
    bool none = false, invert = false;
    switch (op)
    {
        case CHGPU_EQ: if (s < dmin || s > dmax) none = true; else lo = hi = s; break;
        case CHGPU_NE: if (s < dmin || s > dmax) none = true; else lo = hi = s; invert = true; break;
        case CHGPU_LT: if (s <= dmin) none = true; else if (s <= dmax) hi = s - 1; break;
        case CHGPU_LE: if (s < dmin) none = true; else if (s < dmax) hi = s; break;
        case CHGPU_GT: if (s >= dmax) none = true; else if (s >= dmin) lo = s + 1; break;
        case CHGPU_GE: if (s > dmax) none = true; else if (s > dmin) lo = s; break;
    }
    if (none)
    {
        // empty range == full range inverted (for NE the inversion cancels: everything passes)
        lo = dmin;
        hi = dmax;
        invert = !invert;
    }
    spec->is_f64 = false;
    spec->op = op;
    spec->ip.lo = key(lo);
    spec->ip.span = key(hi) - key(lo);
    spec->ip.flip = flip;
    spec->ip.invert = invert ? 1u : 0u;
    return CHGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// vector load helpers: 16 B per lane
// ---------------------------------------------------------------------------------------------
template <typename T, int N>
struct alignas(sizeof(T) * N) Vec
{
    T v[N];
};

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

// streaming (read-once) load: 16-B vectors go out as global_load_dwordx4 ... nt so an 8 GB scan does not churn L2/MALL
template <typename V>
__device__ __forceinline__ V load_stream(const V * p)
{
    if constexpr (sizeof(V) == 16)
    {
        u32x4 r = __builtin_nontemporal_load((const u32x4 *)p);
        return __builtin_bit_cast(V, r);
    }
    else
        return *p;
}

typedef u32 u32x2 __attribute__((ext_vector_type(2)));

// streaming (write-once) store of 8 or 16 bytes
template <typename V>
__device__ __forceinline__ void store_stream(V * p, const V & v)
{
    if constexpr (sizeof(V) == 16)
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), (u32x4 *)p);
    else if constexpr (sizeof(V) == 8)
        __builtin_nontemporal_store(__builtin_bit_cast(u32x2, v), (u32x2 *)p);
    else
        *p = v;
}

template <typename T>
struct AccOf { typedef u64 type; };
template <>
struct AccOf<double> { typedef double type; };
template <>
struct AccOf<float> { typedef double type; }; // sum(Float32) accumulates in Float64

template <typename T>
__device__ __forceinline__ typename AccOf<T>::type widen(T a)
{
    if constexpr (std::is_same<T, double>::value || std::is_same<T, float>::value)
        return (double)a;
    else if constexpr (std::is_signed<T>::value)
        return (u64)(i64)a; // wrap-around two's complement sum (AggregateFunctionSum.h:36-39)
    else
        return (u64)a;
}

__device__ __forceinline__ u64 acc_bits(u64 v) { return v; }
__device__ __forceinline__ u64 acc_bits(double v) { return (u64)__double_as_longlong(v); }
__device__ __forceinline__ u64 wave_reduce_acc(u64 v) { return wave_reduce_add_u64(v); }
__device__ __forceinline__ double wave_reduce_acc(double v) { return wave_reduce_add_f64(v); }

// ---------------------------------------------------------------------------------------------
// fused predicate -> sum(val), count()      (the dominant kernel of config C2: 8 B/row, one pass)
// ---------------------------------------------------------------------------------------------
static constexpr int FS_THREADS = 256;
// Launch geometry, measured on MI355X (tools/tune_filter_sum.hip: 1e9 Int64 rows, variants interleaved in one process):
//   8 workgroups/CU x 4 loads/lane, grid-strided (the textbook shape)            6.1 TB/s
//   FS_WG_PER_CU workgroups/CU, each reading one CONTIGUOUS chunk of FS_UNROLL x 256 16-B vectors per iteration
//   with nontemporal loads (24-32 KiB in flight per CU)                             7.0-7.2 TB/s
// i.e. few waves with deep, contiguous, streaming loads beat many waves.  Plain (non-nt) loads cost ~10 %.
// Two workgroups per CU rather than one: the second hides the first's reduce phase without relying on hipcc to
// software-pipeline the loop (it sinks loads below waits when asked to double-buffer in registers).
static constexpr int FS_WG_PER_CU = 2;
#ifndef FS_UNROLL_SAME
#define FS_UNROLL_SAME 4
#endif

template <typename T, int VEC, bool SAME, bool HAS_COND, typename Pred>
__global__ __launch_bounds__(FS_THREADS) void k_filter_sum(const T * __restrict__ pred_col, const T * __restrict__ val_col,
                                                           const u8 * __restrict__ cond, u64 n, Pred p,
                                                           u64 * __restrict__ part_sum, u64 * __restrict__ part_cnt)
{
    typedef typename AccOf<T>::type Acc;
    typedef Vec<T, VEC> V;
    typedef Vec<u8, VEC> CV;
    constexpr int UNROLL = SAME ? FS_UNROLL_SAME : (FS_UNROLL_SAME + 1) / 2;
    const u64 nvec = n / VEC;
    const V * __restrict__ pv = (const V *)pred_col;
    const V * __restrict__ vv = (const V *)val_col;
    const CV * __restrict__ cv = (const CV *)cond;
    Acc s = 0;
    u64 c = 0;

    // main loop: the workgroup reads UNROLL * 256 consecutive vectors per iteration; every load is issued before the
    // first use (sched_barrier pins that order)
    constexpr u64 CHUNK = (u64)UNROLL * FS_THREADS;
    const u64 n_chunks = nvec / CHUNK;
    for (u64 ch = blockIdx.x; ch < n_chunks; ch += gridDim.x)
    {
        const u64 base = ch * CHUNK + threadIdx.x;
        V a[UNROLL], b[UNROLL];
        CV m[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
        {
            a[k] = load_stream(&pv[base + (u64)k * FS_THREADS]);
            if constexpr (!SAME)
                b[k] = load_stream(&vv[base + (u64)k * FS_THREADS]);
            if constexpr (HAS_COND)
                m[k] = cv[base + (u64)k * FS_THREADS];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e)
            {
                bool pass = p(a[k].v[e]);
                if constexpr (HAS_COND)
                    pass = pass && (m[k].v[e] != 0);
                T x = SAME ? a[k].v[e] : b[k].v[e];
                s += pass ? widen<T>(x) : Acc(0);
                c += pass ? 1 : 0;
            }
    }
    // remainder vectors (< one chunk per workgroup), grid-strided
    const u64 tid = (u64)blockIdx.x * FS_THREADS + threadIdx.x;
    const u64 stride = (u64)gridDim.x * FS_THREADS;
    for (u64 i = n_chunks * CHUNK + tid; i < nvec; i += stride)
    {
        V a = pv[i];
        V b;
        if constexpr (!SAME)
            b = vv[i];
        CV m;
        if constexpr (HAS_COND)
            m = cv[i];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
        {
            bool pass = p(a.v[e]);
            if constexpr (HAS_COND)
                pass = pass && (m.v[e] != 0);
            T x = SAME ? a.v[e] : b.v[e];
            s += pass ? widen<T>(x) : Acc(0);
            c += pass ? 1 : 0;
        }
    }
    // scalar tail (n % VEC rows)
    {
        const u64 r = nvec * VEC + tid;
        if (r < n)
        {
            bool pass = p(pred_col[r]);
            if constexpr (HAS_COND)
                pass = pass && (cond[r] != 0);
            T x = SAME ? pred_col[r] : val_col[r];
            s += pass ? widen<T>(x) : Acc(0);
            c += pass ? 1 : 0;
        }
    }

    // wave -> workgroup reduction; one partial per workgroup, folded in fixed order by k_filter_sum_finish
    __shared__ u64 lds_s[FS_THREADS / WAVE];
    __shared__ u64 lds_c[FS_THREADS / WAVE];
    s = wave_reduce_acc(s);
    c = wave_reduce_add_u64(c);
    if ((threadIdx.x & 63) == 0)
    {
        lds_s[threadIdx.x >> 6] = acc_bits(s);
        lds_c[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        u64 cc = 0;
        if constexpr (std::is_same<Acc, double>::value)
        {
            double ss = 0;
            for (int w = 0; w < FS_THREADS / WAVE; ++w)
            {
                ss += __longlong_as_double((long long)lds_s[w]);
                cc += lds_c[w];
            }
            part_sum[blockIdx.x] = acc_bits(ss);
        }
        else
        {
            u64 ss = 0;
            for (int w = 0; w < FS_THREADS / WAVE; ++w)
            {
                ss += lds_s[w];
                cc += lds_c[w];
            }
            part_sum[blockIdx.x] = ss;
        }
        part_cnt[blockIdx.x] = cc;
    }
}

template <bool IS_F64>
__global__ __launch_bounds__(256) void k_filter_sum_finish(const u64 * __restrict__ part_sum, const u64 * __restrict__ part_cnt,
                                                           u32 n_parts, u64 * __restrict__ result /* {sum bits, count} */)
{
    __shared__ u64 lds_s[4];
    __shared__ u64 lds_c[4];
    u64 c = 0;
    u64 sb;
    if constexpr (IS_F64)
    {
        double s = 0;
        for (u32 i = threadIdx.x; i < n_parts; i += 256)
        {
            s += __longlong_as_double((long long)part_sum[i]);
            c += part_cnt[i];
        }
        s = wave_reduce_add_f64(s);
        sb = acc_bits(s);
    }
    else
    {
        u64 s = 0;
        for (u32 i = threadIdx.x; i < n_parts; i += 256)
        {
            s += part_sum[i];
            c += part_cnt[i];
        }
        sb = wave_reduce_add_u64(s);
    }
    c = wave_reduce_add_u64(c);
    if ((threadIdx.x & 63) == 0)
    {
        lds_s[threadIdx.x >> 6] = sb;
        lds_c[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        u64 cc = lds_c[0] + lds_c[1] + lds_c[2] + lds_c[3];
        if constexpr (IS_F64)
        {
            double ss = 0;
            for (int w = 0; w < 4; ++w)
                ss += __longlong_as_double((long long)lds_s[w]);
            result[0] = acc_bits(ss);
        }
        else
            result[0] = lds_s[0] + lds_s[1] + lds_s[2] + lds_s[3];
        result[1] = cc;
    }
}

template <typename T, typename Pred>
static int launch_filter_sum_t(chgpu_ctx * ctx, const void * pred, const void * val, const u8 * cond, u64 n, Pred p, u64 * result_dev)
{
    constexpr int VECW = 16 / sizeof(T);
    const bool same = (pred == val);
    const bool aligned = (((uintptr_t)pred | (uintptr_t)val) & 15) == 0 && (!cond || ((uintptr_t)cond % VECW) == 0);
    // persistent grid: FS_WG_PER_CU 256-thread workgroups per CU (see the measurement note above k_filter_sum)
    const u32 wg_same = tune_env(ctx, "tune_fs_wg", FS_WG_PER_CU), wg_two = tune_env(ctx, "tune_fs2_wg", FS_WG_PER_CU);
    const u32 grid = chgpu_grid_for(ctx, (n + VECW - 1) / VECW, FS_THREADS, same ? wg_same : wg_two);
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, (size_t)grid * 2 * sizeof(u64), &scratch));
    u64 * part_sum = (u64 *)scratch;
    u64 * part_cnt = part_sum + grid;
    const T * pp = (const T *)pred;
    const T * vp = (const T *)val;
#define FS_LAUNCH(VEC, SAME, HC) \
    hipLaunchKernelGGL((k_filter_sum<T, VEC, SAME, HC, Pred>), dim3(grid), dim3(FS_THREADS), 0, ctx->stream, pp, vp, cond, n, p, part_sum, part_cnt)
    if (aligned)
    {
        if (cond) { if (same) FS_LAUNCH(VECW, true, true); else FS_LAUNCH(VECW, false, true); }
        else      { if (same) FS_LAUNCH(VECW, true, false); else FS_LAUNCH(VECW, false, false); }
    }
    else
    {
        if (cond) { if (same) FS_LAUNCH(1, true, true); else FS_LAUNCH(1, false, true); }
        else      { if (same) FS_LAUNCH(1, true, false); else FS_LAUNCH(1, false, false); }
    }
#undef FS_LAUNCH
    hipLaunchKernelGGL((k_filter_sum_finish<std::is_same<typename AccOf<T>::type, double>::value>), dim3(1), dim3(256), 0, ctx->stream, part_sum, part_cnt, grid, result_dev);
    ctx->counters[6] += 2;
    CHGPU_HIP(hipGetLastError());
    return CHGPU_OK;
}

template <typename Pred>
static int launch_filter_sum_int(chgpu_ctx * ctx, int type, const void * pred, const void * val, const u8 * cond, u64 n, Pred p, u64 * result_dev)
{
    switch (type)
    {
        case CHGPU_I64: return launch_filter_sum_t<i64, Pred>(ctx, pred, val, cond, n, p, result_dev);
        case CHGPU_U64: return launch_filter_sum_t<u64, Pred>(ctx, pred, val, cond, n, p, result_dev);
        case CHGPU_U32: return launch_filter_sum_t<u32, Pred>(ctx, pred, val, cond, n, p, result_dev);
        case CHGPU_I32: return launch_filter_sum_t<i32, Pred>(ctx, pred, val, cond, n, p, result_dev);
        case CHGPU_U8: return launch_filter_sum_t<u8, Pred>(ctx, pred, val, cond, n, p, result_dev);
        case CHGPU_U16: return launch_filter_sum_t<u16, Pred>(ctx, pred, val, cond, n, p, result_dev);
        case CHGPU_I16: return launch_filter_sum_t<i16, Pred>(ctx, pred, val, cond, n, p, result_dev);
        case CHGPU_I8: return launch_filter_sum_t<i8, Pred>(ctx, pred, val, cond, n, p, result_dev);
        default: return chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "unsupported column type %d", type);
    }
}

static int launch_filter_sum(chgpu_ctx * ctx, int type, const void * pred, const void * val, const u8 * cond, u64 n,
                             const CmpSpec * spec /* NULL = no predicate */, u64 * result_dev)
{
    if (!spec)
    {
        if (type == CHGPU_F64)
            return launch_filter_sum_t<double, TruePred>(ctx, pred, val, cond, n, TruePred(), result_dev);
        if (type == CHGPU_F32)
            return launch_filter_sum_t<float, TruePred>(ctx, pred, val, cond, n, TruePred(), result_dev);
        return launch_filter_sum_int<TruePred>(ctx, type, pred, val, cond, n, TruePred(), result_dev);
    }
    if (spec->is_f64)
    {
        switch (spec->op)
        {
#define F64CASE(OP) case OP: return type == CHGPU_F32 ? launch_filter_sum_t<float, F64Pred<OP>>(ctx, pred, val, cond, n, F64Pred<OP>{spec->fs}, result_dev) \
                                                   : launch_filter_sum_t<double, F64Pred<OP>>(ctx, pred, val, cond, n, F64Pred<OP>{spec->fs}, result_dev);
            F64CASE(CHGPU_EQ) F64CASE(CHGPU_NE) F64CASE(CHGPU_LT) F64CASE(CHGPU_GT) F64CASE(CHGPU_LE) F64CASE(CHGPU_GE)
#undef F64CASE
        }
        return chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "bad op");
    }
    return launch_filter_sum_int<IntRangePred>(ctx, type, pred, val, cond, n, spec->ip, result_dev);
}

// Predicate and value columns of different types: the one-pass kernel is instantiated per type, so the predicate is first
// materialised as a UInt8 mask (k_cmp_mask) and the value column summed under it (addManyConditional,
// AggregateFunctionSum.h:138-236) -- what the reference does, minus its filtered copy.  9 + w B/row instead of 8 + w.
extern "C" int chgpu_cmp_const(chgpu_ctx * ctx, const chgpu_col * col, int op, int scalar_type, const void * scalar, chgpu_col ** mask_out);
static int filter_sum_mixed(chgpu_ctx * ctx, const chgpu_col * pred, int op, int scalar_type, const void * scalar, const chgpu_col * val, u64 * result_dev)
{
    chgpu_col * mask = nullptr;
    CHGPU_TRY(chgpu_cmp_const(ctx, pred, op, scalar_type, scalar, &mask));
    const int rc = launch_filter_sum(ctx, val->type, val->data, val->data, (const u8 *)mask->data, val->rows, nullptr, result_dev);
    chgpu_col_free(mask); // back to the pool: reuse is ordered behind the kernel on the context's stream
    return rc;
}

extern "C" int chgpu_filter_sum_async(chgpu_ctx * ctx, const chgpu_col * pred, int op, int scalar_type, const void * scalar,
                                      const chgpu_col * val, chgpu_col * result)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && pred && val && result, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(pred->rows == val->rows, CHGPU_ERR_SIZES_MISMATCH, "Size of predicate column (%llu) doesn't match size of value column (%llu)",
                  (unsigned long long)pred->rows, (unsigned long long)val->rows);
    CHGPU_REQUIRE(result->type == CHGPU_U64 && result->rows >= 2, CHGPU_ERR_BAD_ARGUMENTS, "result must be a UInt64 column of 2 rows");
    ctx->counters[5] += pred->rows;
    if (pred->type != val->type)
        return filter_sum_mixed(ctx, pred, op, scalar_type, scalar, val, (u64 *)result->data);
    CmpSpec spec;
    CHGPU_TRY(make_cmp_spec(pred->type, op, scalar_type, scalar, &spec));
    return launch_filter_sum(ctx, pred->type, pred->data, val->data, nullptr, pred->rows, &spec, (u64 *)result->data);
}

extern "C" int chgpu_filter_sum(chgpu_ctx * ctx, const chgpu_col * pred, int op, int scalar_type, const void * scalar,
                                const chgpu_col * val, void * sum_out, uint64_t * count_out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && pred && val && sum_out && count_out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(pred->rows == val->rows, CHGPU_ERR_SIZES_MISMATCH, "Size of predicate column (%llu) doesn't match size of value column (%llu)",
                  (unsigned long long)pred->rows, (unsigned long long)val->rows);
    void * scratch = nullptr;
    // result slot lives behind the partials: ask for the partials' worth first so the pointer stays valid
    const u32 grid_cap = (u32)ctx->num_cus * 8;
    CHGPU_TRY(chgpu_scratch(ctx, (size_t)grid_cap * 2 * sizeof(u64) + 64, &scratch));
    u64 * result_dev = (u64 *)((char *)scratch + (size_t)grid_cap * 2 * sizeof(u64));
    if (pred->type != val->type)
        CHGPU_TRY(filter_sum_mixed(ctx, pred, op, scalar_type, scalar, val, result_dev));
    else
    {
        CmpSpec spec;
        CHGPU_TRY(make_cmp_spec(pred->type, op, scalar_type, scalar, &spec));
        CHGPU_TRY(launch_filter_sum(ctx, pred->type, pred->data, val->data, nullptr, pred->rows, &spec, result_dev));
    }
    u64 res[2];
    CHGPU_TRY(chgpu_read_back(ctx, result_dev, res, sizeof(res)));
    memcpy(sum_out, &res[0], 8);
    *count_out = res[1];
    ctx->counters[0] += res[1];
    ctx->counters[1] += res[1] * chgpu_type_size(val->type);
    ctx->counters[5] += pred->rows;
    return CHGPU_OK;
}

static int sum_add_many_impl(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * cond, u64 row_begin, u64 row_end, void * state8)
{
    CHGPU_REQUIRE(ctx && col && state8, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(row_begin <= row_end && row_end <= col->rows, CHGPU_ERR_BAD_ARGUMENTS, "row range [%llu,%llu) out of bounds (%llu rows)",
                  (unsigned long long)row_begin, (unsigned long long)row_end, (unsigned long long)col->rows);
    if (cond)
    {
        CHGPU_REQUIRE(cond->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS, "condition column must be UInt8");
        CHGPU_REQUIRE(cond->rows == col->rows, CHGPU_ERR_SIZES_MISMATCH, "Size of condition column (%llu) doesn't match size of column (%llu)",
                      (unsigned long long)cond->rows, (unsigned long long)col->rows);
    }
    const size_t es = chgpu_type_size(col->type);
    const u64 n = row_end - row_begin;
    const void * p = (const char *)col->data + row_begin * es;
    const u8 * c = cond ? (const u8 *)cond->data + row_begin : nullptr;
    void * scratch = nullptr;
    const u32 grid_cap = (u32)ctx->num_cus * 8;
    CHGPU_TRY(chgpu_scratch(ctx, (size_t)grid_cap * 2 * sizeof(u64) + 64, &scratch));
    u64 * result_dev = (u64 *)((char *)scratch + (size_t)grid_cap * 2 * sizeof(u64));
    CHGPU_TRY(launch_filter_sum(ctx, col->type, p, p, c, n, nullptr, result_dev));
    u64 res[2];
    CHGPU_TRY(chgpu_read_back(ctx, result_dev, res, sizeof(res)));
    if (chgpu_type_is_float(col->type))
    {
        double batch, st;
        memcpy(&batch, &res[0], 8);
        memcpy(&st, state8, 8);
        st += batch;
        memcpy(state8, &st, 8);
    }
    else
    {
        u64 st;
        memcpy(&st, state8, 8);
        st += res[0];
        memcpy(state8, &st, 8);
    }
    ctx->counters[5] += n;
    return CHGPU_OK;
}

extern "C" int chgpu_sum_add_many(chgpu_ctx * ctx, const chgpu_col * col, uint64_t row_begin, uint64_t row_end, void * state8)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    return sum_add_many_impl(ctx, col, nullptr, row_begin, row_end, state8);
}

extern "C" int chgpu_sum_add_many_conditional(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * cond_u8,
                                              uint64_t row_begin, uint64_t row_end, void * state8)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(cond_u8, CHGPU_ERR_BAD_ARGUMENTS, "condition column is NULL");
    return sum_add_many_impl(ctx, col, cond_u8, row_begin, row_end, state8);
}

// ---------------------------------------------------------------------------------------------
// comparison -> UInt8 mask
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, typename Pred>
__global__ __launch_bounds__(256) void k_cmp_mask(const T * __restrict__ a, u64 n, Pred p, u8 * __restrict__ c)
{
    typedef Vec<T, VEC> V;
    typedef Vec<u8, VEC> CV;
    constexpr int UNROLL = 4;
    const u64 nvec = n / VEC;
    const V * __restrict__ av = (const V *)a;
    CV * __restrict__ cv = (CV *)c;
    // Same streaming geometry as k_filter_sum: contiguous 16 KiB chunk per workgroup iteration, nontemporal loads; inside the
    // chunk each WAVE owns UNROLL*64 consecutive vectors.  The 2-4 mask bytes a lane produces per vector are transposed
    // through a wave-private LDS strip so every lane stores UNROLL*VEC (8 or 16) CONTIGUOUS mask bytes: one wide store per
    // iteration instead of UNROLL narrow ones.  (Giving each lane UNROLL consecutive vectors to the same end was measured
    // 20 % slower: the 64-B-strided loads cost more than the narrow stores.)
    constexpr u64 CHUNK = (u64)UNROLL * 256;
    const u64 n_chunks = nvec / CHUNK;
    constexpr int OB = VEC * UNROLL; // mask bytes per lane per iteration
    constexpr bool TRANSPOSE = (VEC == 2 || VEC == 4);
    __shared__ __attribute__((aligned(16))) u8 lds_m[4][64 * (TRANSPOSE ? OB : 1)];
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (u64 ch = blockIdx.x; ch < n_chunks; ch += gridDim.x)
    {
        const u64 wbase = ch * CHUNK + (u64)wave * (64 * UNROLL); // first vector of this wave's strip
        V x[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
            x[k] = load_stream(&av[wbase + (u64)k * 64 + lane]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TRANSPOSE)
        {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k)
            {
                CV m;
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    m.v[e] = p(x[k].v[e]) ? 1 : 0;
                *(CV *)&lds_m[wave][(k * 64 + lane) * VEC] = m;
            }
            __builtin_amdgcn_wave_barrier();
            typedef Vec<u8, OB> OV;
            const OV o = *(const OV *)&lds_m[wave][lane * OB]; // LDS ops of one wave complete in order
            store_stream((OV *)(c + wbase * VEC + (u64)lane * OB), o);
            __builtin_amdgcn_wave_barrier();
        }
        else
        {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k)
            {
                CV m;
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    m.v[e] = p(x[k].v[e]) ? 1 : 0;
                cv[wbase + (u64)k * 64 + lane] = m;
            }
        }
    }
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = n_chunks * CHUNK + tid; i < nvec; i += stride)
    {
        V x = av[i];
        CV m;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            m.v[e] = p(x.v[e]) ? 1 : 0;
        cv[i] = m;
    }
    const u64 r = nvec * VEC + tid;
    if (r < n)
        c[r] = p(a[r]) ? 1 : 0;
}

template <typename T, typename Pred>
static int launch_cmp_t(chgpu_ctx * ctx, const void * a, u64 n, Pred p, u8 * c)
{
    constexpr int VECW = 16 / sizeof(T);
    const bool aligned = ((uintptr_t)a & 15) == 0 && ((uintptr_t)c % VECW) == 0;
    const u32 wg_per_cu = tune_env(ctx, "tune_cmp_wg", 2);
    const u32 grid = chgpu_grid_for(ctx, (n + VECW - 1) / VECW, 256, wg_per_cu);
    if (aligned)
        hipLaunchKernelGGL((k_cmp_mask<T, VECW, Pred>), dim3(grid), dim3(256), 0, ctx->stream, (const T *)a, n, p, c);
    else
        hipLaunchKernelGGL((k_cmp_mask<T, 1, Pred>), dim3(grid), dim3(256), 0, ctx->stream, (const T *)a, n, p, c);
    ctx->counters[6] += 1;
    CHGPU_HIP(hipGetLastError());
    return CHGPU_OK;
}

extern "C" int chgpu_cmp_const(chgpu_ctx * ctx, const chgpu_col * col, int op, int scalar_type, const void * scalar, chgpu_col ** mask_out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && col && mask_out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CmpSpec spec;
    CHGPU_TRY(make_cmp_spec(col->type, op, scalar_type, scalar, &spec));
    chgpu_col * m = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U8, col->rows, &m));
    int rc = CHGPU_OK;
    const u64 n = col->rows;
    u8 * c = (u8 *)m->data;
    if (n)
    {
        if (spec.is_f64)
        {
            switch (spec.op)
            {
#define F64CASE(OP) case OP: rc = col->type == CHGPU_F32 ? launch_cmp_t<float, F64Pred<OP>>(ctx, col->data, n, F64Pred<OP>{spec.fs}, c) \
                                                       : launch_cmp_t<double, F64Pred<OP>>(ctx, col->data, n, F64Pred<OP>{spec.fs}, c); break;
                F64CASE(CHGPU_EQ) F64CASE(CHGPU_NE) F64CASE(CHGPU_LT) F64CASE(CHGPU_GT) F64CASE(CHGPU_LE) F64CASE(CHGPU_GE)
#undef F64CASE
            }
        }
        else
        {
            switch (col->type)
            {
                case CHGPU_I64: rc = launch_cmp_t<i64, IntRangePred>(ctx, col->data, n, spec.ip, c); break;
                case CHGPU_U64: rc = launch_cmp_t<u64, IntRangePred>(ctx, col->data, n, spec.ip, c); break;
                case CHGPU_U32: rc = launch_cmp_t<u32, IntRangePred>(ctx, col->data, n, spec.ip, c); break;
                case CHGPU_I32: rc = launch_cmp_t<i32, IntRangePred>(ctx, col->data, n, spec.ip, c); break;
                case CHGPU_U8: rc = launch_cmp_t<u8, IntRangePred>(ctx, col->data, n, spec.ip, c); break;
                case CHGPU_U16: rc = launch_cmp_t<u16, IntRangePred>(ctx, col->data, n, spec.ip, c); break;
                case CHGPU_I16: rc = launch_cmp_t<i16, IntRangePred>(ctx, col->data, n, spec.ip, c); break;
                case CHGPU_I8: rc = launch_cmp_t<i8, IntRangePred>(ctx, col->data, n, spec.ip, c); break;
                default: rc = chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "unsupported column type");
            }
        }
    }
    if (rc != CHGPU_OK)
    {
        chgpu_col_free(m);
        return rc;
    }
    *mask_out = m;
    return CHGPU_OK;
}

__global__ __launch_bounds__(256) void k_and_not(const u8 * __restrict__ d, const u8 * __restrict__ nul, u64 n, u8 * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
        out[i] = (d[i] && !nul[i]) ? 1 : 0;
}

extern "C" int chgpu_filter_description_nullable(chgpu_ctx * ctx, const chgpu_col * data, const chgpu_col * nul, chgpu_col ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && data && nul && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(data->type == CHGPU_U8 && nul->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS, "Nullable(UInt8) filter expected");
    CHGPU_REQUIRE(data->rows == nul->rows, CHGPU_ERR_SIZES_MISMATCH, "null map size mismatch");
    chgpu_col * m = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U8, data->rows, &m));
    if (data->rows)
    {
        hipLaunchKernelGGL(k_and_not, dim3(chgpu_grid_for(ctx, data->rows, 256, 8)), dim3(256), 0, ctx->stream,
                           (const u8 *)data->data, (const u8 *)nul->data, data->rows, (u8 *)m->data);
        ctx->counters[6] += 1;
    }
    *out = m;
    return CHGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// filter: order-preserving compaction.  Rows are cut into chunks of 1024; one wave owns one chunk.
//   pass 1  k_mask_chunk_counts: non-zero mask bytes per chunk                    (1 B/row read)
//   pass 2  exclusive scan of the chunk counts (scan.hip)                          (tiny)
//   pass 3  k_filter_scatter: each lane takes R = 16/sizeof(T) consecutive rows (one 16-B load), wave ballots
//           give every kept row its rank inside the chunk, stores land compacted    (1 + sizeof(T) + sizeof(T)*sel B/row)
// ---------------------------------------------------------------------------------------------
static constexpr u32 CHUNK_ROWS = 1024;

__device__ __forceinline__ u32 count_nonzero_bytes16(const uint4 v)
{
    auto nz = [](u32 x) { return __popc((x | ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u); };
    return nz(v.x) + nz(v.y) + nz(v.z) + nz(v.w);
}

__global__ __launch_bounds__(256) void k_mask_chunk_counts(const u8 * __restrict__ mask, u64 n, u32 * __restrict__ counts, u64 n_chunks)
{
    const u32 lane = threadIdx.x & 63;
    const u64 wave0 = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const u64 n_waves = ((u64)gridDim.x * 256) >> 6;
    const bool aligned = ((uintptr_t)mask & 15) == 0;
    for (u64 chunk = wave0; chunk < n_chunks; chunk += n_waves)
    {
        const u64 base = chunk * CHUNK_ROWS + (u64)lane * 16;
        u32 c = 0;
        if (aligned && base + 16 <= n)
            c = count_nonzero_bytes16(*(const uint4 *)(mask + base));
        else
            for (u32 k = 0; k < 16; ++k)
                if (base + k < n)
                    c += mask[base + k] != 0;
        c = wave_reduce_add_u32(c);
        if (lane == 0)
            counts[chunk] = c;
    }
}

template <typename T, bool STAGED = false>
__global__ __launch_bounds__(256) void k_filter_scatter(const T * __restrict__ data, const u8 * __restrict__ mask, u64 n,
                                                        const u64 * __restrict__ chunk_offsets, u64 n_chunks, T * __restrict__ out)
{
    // STAGED: see k_filter_scatter_multi
    __shared__ T stage[STAGED ? 4 : 1][STAGED ? CHUNK_ROWS : 1];
    T * const st = stage[STAGED ? (threadIdx.x >> 6) : 0];
    constexpr int R = 16 / sizeof(T);        // rows per lane per group
    constexpr u32 GROUP = 64 * R;            // rows per wave per group
    constexpr int G = CHUNK_ROWS / GROUP;    // groups per chunk
    typedef Vec<T, R> V;
    typedef Vec<u8, R> MV;
    const u32 lane = threadIdx.x & 63;
    const u64 wave0 = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const u64 n_waves = ((u64)gridDim.x * 256) >> 6;
    const bool aligned = (((uintptr_t)data) & 15) == 0 && (((uintptr_t)mask) % R) == 0;

    for (u64 chunk = wave0; chunk < n_chunks; chunk += n_waves)
    {
        u64 pos = chunk_offsets[chunk];
        const u64 cbase = chunk * CHUNK_ROWS;
        if (aligned && cbase + CHUNK_ROWS <= n)
        {
            V x[G];
            MV m[G];
#pragma unroll
            for (int g = 0; g < G; ++g)
            {
                const u64 row = cbase + (u64)g * GROUP + (u64)lane * R;
                x[g] = *(const V *)(data + row);
                m[g] = *(const MV *)(mask + row);
            }
            if constexpr (STAGED)
            {
                u32 run = 0;
#pragma unroll
                for (int g = 0; g < G; ++g)
                {
                    u32 before = 0, total = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                    {
                        const u64 b = __ballot(m[g].v[r] != 0);
                        before += mbcnt(b);
                        total += __popcll(b);
                    }
                    u32 o = run + before;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (m[g].v[r] != 0)
                            st[o++] = x[g].v[r];
                    run += total;
                }
                T * const dst = out + pos;
                for (u32 i = lane; i < run; i += 64) // LDS operations of one wave complete in order: no barrier
                    dst[i] = st[i];
            }
            else
            {
#pragma unroll
                for (int g = 0; g < G; ++g)
                {
                    u32 before = 0, total = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                    {
                        const u64 b = __ballot(m[g].v[r] != 0);
                        before += mbcnt(b);
                        total += __popcll(b);
                    }
                    u64 o = pos + before;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (m[g].v[r] != 0)
                            out[o++] = x[g].v[r]; // plain store: L2 merges the partial lines (nontemporal stores here: +25 % time)
                    pos += total;
                }
            }
        }
        else
        {
            // ragged tail chunk / unaligned view: same order, guarded scalar accesses
            for (int g = 0; g < G; ++g)
            {
                const u64 row = cbase + (u64)g * GROUP + (u64)lane * R;
                T xv[R];
                bool keep[R];
                u32 before = 0, total = 0;
#pragma unroll
                for (int r = 0; r < R; ++r)
                {
                    const bool in = row + r < n;
                    keep[r] = in && mask[in ? row + r : 0] != 0;
                    if (keep[r])
                        xv[r] = data[row + r];
                    const u64 b = __ballot(keep[r]);
                    before += mbcnt(b);
                    total += __popcll(b);
                }
                u64 o = pos + before;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (keep[r])
                        out[o++] = xv[r];
                pos += total;
            }
        }
    }
}

// The same compaction for NC columns of one element width at once (IColumn::filter over every column of a Block,
// FilterTransform.cpp:190-206): the mask bytes are loaded and balloted ONCE per group of rows and every column's rows go out with
// the same ranks -- per column the pass then costs sizeof(T) * (1 + selectivity) bytes per row instead of 1 + that.
template <typename T, int NC>
struct FilterCols
{
    const T * data[NC];
    T * out[NC];
};
template <typename T, int NC, bool STAGED = false>
__global__ __launch_bounds__(256) void k_filter_scatter_multi(FilterCols<T, NC> c, const u8 * __restrict__ mask, u64 n, const u64 * __restrict__ chunk_offsets, u64 n_chunks)
{
    // STAGED: the kept rows of a chunk are compacted in a wave-private LDS buffer first and leave as full 64-lane stores of consecutive
    // elements (run / 64 store instructions per column instead of one sparse store per row slot)
    __shared__ T stage[STAGED ? 4 : 1][STAGED ? CHUNK_ROWS : 1];
    T * const st = stage[STAGED ? (threadIdx.x >> 6) : 0];
    constexpr int R = 16 / sizeof(T);
    constexpr u32 GROUP = 64 * R;
    constexpr int G = CHUNK_ROWS / GROUP;
    typedef Vec<T, R> V;
    typedef Vec<u8, R> MV;
    const u32 lane = threadIdx.x & 63;
    const u64 wave0 = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const u64 n_waves = ((u64)gridDim.x * 256) >> 6;
    bool aligned = (((uintptr_t)mask) % R) == 0;
#pragma unroll
    for (int k = 0; k < NC; ++k)
        aligned = aligned && (((uintptr_t)c.data[k]) & 15) == 0;
    for (u64 chunk = wave0; chunk < n_chunks; chunk += n_waves)
    {
        const u64 pos0 = chunk_offsets[chunk];
        const u64 cbase = chunk * CHUNK_ROWS;
        if (aligned && cbase + CHUNK_ROWS <= n)
        {
            MV m[G];
#pragma unroll
            for (int g = 0; g < G; ++g)
                m[g] = *(const MV *)(mask + cbase + (u64)g * GROUP + (u64)lane * R);
            // rank of this lane's first kept row of every group, and what to add per kept row
            u32 first[G];
            u32 run = 0;
#pragma unroll
            for (int g = 0; g < G; ++g)
            {
                u32 before = 0, total = 0;
#pragma unroll
                for (int r = 0; r < R; ++r)
                {
                    const u64 b = __ballot(m[g].v[r] != 0);
                    before += mbcnt(b);
                    total += __popcll(b);
                }
                first[g] = run + before;
                run += total;
            }
#pragma unroll
            for (int k = 0; k < NC; ++k)
            {
                V x[G];
#pragma unroll
                for (int g = 0; g < G; ++g)
                    x[g] = *(const V *)(c.data[k] + cbase + (u64)g * GROUP + (u64)lane * R);
                if constexpr (STAGED)
                {
#pragma unroll
                    for (int g = 0; g < G; ++g)
                    {
                        u32 o = first[g];
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (m[g].v[r] != 0)
                                st[o++] = x[g].v[r];
                    }
                    T * const dst = c.out[k] + pos0;
                    for (u32 i = lane; i < run; i += 64) // LDS operations of one wave complete in order: no barrier
                        dst[i] = st[i];
                }
                else
                {
#pragma unroll
                    for (int g = 0; g < G; ++g)
                    {
                        u64 o = pos0 + first[g];
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (m[g].v[r] != 0)
                                c.out[k][o++] = x[g].v[r];
                    }
                }
            }
        }
        else
        {
            // ragged tail chunk / unaligned view: same order, guarded scalar accesses
            u64 pos = pos0;
            for (int g = 0; g < G; ++g)
            {
                const u64 row = cbase + (u64)g * GROUP + (u64)lane * R;
                bool keep[R];
                u32 before = 0, total = 0;
#pragma unroll
                for (int r = 0; r < R; ++r)
                {
                    const bool in = row + r < n;
                    keep[r] = in && mask[in ? row + r : 0] != 0;
                    const u64 b = __ballot(keep[r]);
                    before += mbcnt(b);
                    total += __popcll(b);
                }
#pragma unroll
                for (int k = 0; k < NC; ++k)
                {
                    u64 o = pos + before;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (keep[r])
                            c.out[k][o++] = c.data[k][row + r];
                }
                pos += total;
            }
        }
    }
}

__device__ __forceinline__ u32 count_nonzero_bytes16(const uint4 v);

// countBytesInFilter (ColumnsCommon.cpp:31-58): non-zero bytes of the mask.  16-byte nontemporal loads, four in flight per
// lane, a 5-op bit trick per 4 bytes (the generic filter+sum kernel instantiated for UInt8 did ~3 VALU ops per BYTE and ran
// at 2.4 TB/s); one atomic per wave at the end.
__global__ __launch_bounds__(256) void k_count_nonzero(const u8 * __restrict__ mask, u64 n, unsigned long long * __restrict__ result)
{
    constexpr int U = 4;
    u64 c = 0;
    const u64 nvec = n / 16;
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const v4u * __restrict__ mv = (const v4u *)mask;
    auto nz16 = [](const v4u v) {
        uint4 q;
        q.x = v.x, q.y = v.y, q.z = v.z, q.w = v.w;
        return count_nonzero_bytes16(q);
    };
    const u64 stride = (u64)gridDim.x * 256;
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    for (; i + (u64)(U - 1) * stride < nvec; i += (u64)U * stride)
    {
        v4u v[U];
#pragma unroll
        for (int k = 0; k < U; ++k)
            v[k] = __builtin_nontemporal_load(&mv[i + (u64)k * stride]);
#pragma unroll
        for (int k = 0; k < U; ++k)
            c += nz16(v[k]);
    }
    for (; i < nvec; i += stride)
        c += nz16(mv[i]);
    for (u64 r = nvec * 16 + (u64)blockIdx.x * 256 + threadIdx.x; r < n; r += stride)
        c += mask[r] != 0;
    c = wave_reduce_add_u64(c);
    if ((threadIdx.x & 63) == 0 && c)
        atomicAdd(result, (unsigned long long)c);
}

extern "C" int chgpu_count_bytes_in_filter(chgpu_ctx * ctx, const chgpu_col * mask, uint64_t * count)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && mask && count, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(mask->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS, "filter must be a UInt8 column");
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, 256, &scratch));
    unsigned long long * result_dev = (unsigned long long *)scratch;
    CHGPU_HIP(hipMemsetAsync(result_dev, 0, sizeof(u64), ctx->stream));
    if (((uintptr_t)mask->data & 15) == 0)
    {
        const u32 grid = chgpu_grid_for(ctx, (mask->rows + 15) / 16, 256, 8);
        hipLaunchKernelGGL(k_count_nonzero, dim3(grid), dim3(256), 0, ctx->stream, (const u8 *)mask->data, (u64)mask->rows, result_dev);
        ctx->counters[6] += 1;
        CHGPU_HIP(hipGetLastError());
    }
    else
    {
        // a view that is not 16-byte aligned: the generic reduction (sum of the "non-zero" indicator)
        CmpSpec spec;
        const u8 zero = 0;
        CHGPU_TRY(make_cmp_spec(CHGPU_U8, CHGPU_NE, CHGPU_U8, &zero, &spec));
        const u32 grid_cap = (u32)ctx->num_cus * 8;
        CHGPU_TRY(chgpu_scratch(ctx, (size_t)grid_cap * 2 * sizeof(u64) + 64, &scratch));
        u64 * res2 = (u64 *)((char *)scratch + (size_t)grid_cap * 2 * sizeof(u64));
        CHGPU_TRY(launch_filter_sum(ctx, CHGPU_U8, mask->data, mask->data, nullptr, mask->rows, &spec, res2));
        u64 res[2];
        CHGPU_TRY(chgpu_read_back(ctx, res2, res, sizeof(res)));
        *count = res[1];
        return CHGPU_OK;
    }
    u64 res = 0;
    CHGPU_TRY(chgpu_read_back(ctx, result_dev, &res, sizeof(res)));
    *count = res;
    return CHGPU_OK;
}

// Shared by chgpu_filter and chgpu_filter_columns: per-chunk popcounts of the mask, exclusive scan, total (one read-back).
struct FilterPlan
{
    u64 n = 0, n_chunks = 0, total = 0;
    u64 * offsets = nullptr;
    u32 grid = 0;
};
static int filter_plan(chgpu_ctx * ctx, const chgpu_col * mask, FilterPlan * fp)
{
    const u64 n = mask->rows;
    const u64 n_chunks = (n + CHUNK_ROWS - 1) / CHUNK_ROWS;
    // scratch layout: counts u32[n_chunks] | offsets u64[n_chunks] | total u64 | scan tmp
    const size_t counts_b = ((n_chunks * sizeof(u32) + 255) / 256) * 256;
    const size_t offs_b = ((n_chunks * sizeof(u64) + 255) / 256) * 256;
    const size_t tmp_b = chgpu_scan_tmp_bytes(n_chunks);
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, counts_b + offs_b + 256 + tmp_b, &scratch));
    u32 * counts = (u32 *)scratch;
    u64 * offsets = (u64 *)((char *)scratch + counts_b);
    u64 * total_dev = (u64 *)((char *)scratch + counts_b + offs_b);
    void * tmp = (char *)scratch + counts_b + offs_b + 256;
    const u32 wg_cnt = tune_env(ctx, "tune_fcount_wg", 8), wg_sc = tune_env(ctx, "tune_fscatter_wg", 8);
    const u32 grid_cnt = chgpu_grid_for(ctx, n_chunks * 64, 256, wg_cnt);
    hipLaunchKernelGGL(k_mask_chunk_counts, dim3(grid_cnt), dim3(256), 0, ctx->stream, (const u8 *)mask->data, n, counts, n_chunks);
    ctx->counters[6] += 1;
    CHGPU_TRY(chgpu_scan_exclusive_u32_u64(ctx, counts, offsets, n_chunks, total_dev, tmp, tmp_b));
    CHGPU_TRY(chgpu_read_back(ctx, total_dev, &fp->total, sizeof(fp->total)));
    fp->n = n;
    fp->n_chunks = n_chunks;
    fp->offsets = offsets;
    fp->grid = chgpu_grid_for(ctx, n_chunks * 64, 256, wg_sc);
    return CHGPU_OK;
}
// 4- and 8-byte columns: kept rows compacted in LDS, full-width stores (A/B: CHGPU_TUNE_FILTER_NO_STAGED)
static bool filter_staged(const chgpu_ctx * ctx)
{
    const bool on = chgpu_opt(ctx, "tune_filter_no_staged", 0) == 0;
    return on;
}
static int filter_apply(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * mask, const FilterPlan & fp, chgpu_col ** out)
{
    chgpu_col * res = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, col->type, fp.total, &res));
    if (fp.total)
    {
        switch (chgpu_type_size(col->type))
        {
            case 8:
                if (filter_staged(ctx))
                    hipLaunchKernelGGL((k_filter_scatter<u64, true>), dim3(fp.grid), dim3(256), 0, ctx->stream, (const u64 *)col->data, (const u8 *)mask->data, fp.n, fp.offsets, fp.n_chunks, (u64 *)res->data);
                else
                    hipLaunchKernelGGL((k_filter_scatter<u64, false>), dim3(fp.grid), dim3(256), 0, ctx->stream, (const u64 *)col->data, (const u8 *)mask->data, fp.n, fp.offsets, fp.n_chunks, (u64 *)res->data);
                break;
            case 4:
                if (filter_staged(ctx))
                    hipLaunchKernelGGL((k_filter_scatter<u32, true>), dim3(fp.grid), dim3(256), 0, ctx->stream, (const u32 *)col->data, (const u8 *)mask->data, fp.n, fp.offsets, fp.n_chunks, (u32 *)res->data);
                else
                    hipLaunchKernelGGL((k_filter_scatter<u32, false>), dim3(fp.grid), dim3(256), 0, ctx->stream, (const u32 *)col->data, (const u8 *)mask->data, fp.n, fp.offsets, fp.n_chunks, (u32 *)res->data);
                break;
            case 2:
                hipLaunchKernelGGL(k_filter_scatter<u16>, dim3(fp.grid), dim3(256), 0, ctx->stream, (const u16 *)col->data, (const u8 *)mask->data, fp.n, fp.offsets, fp.n_chunks, (u16 *)res->data);
                break;
            default:
                hipLaunchKernelGGL(k_filter_scatter<u8>, dim3(fp.grid), dim3(256), 0, ctx->stream, (const u8 *)col->data, (const u8 *)mask->data, fp.n, fp.offsets, fp.n_chunks, (u8 *)res->data);
                break;
        }
        ctx->counters[6] += 1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        chgpu_col_free(res);
        return chgpu_set_error(CHGPU_ERR_DEVICE, "filter launch: %s", hipGetErrorString(e));
    }
    ctx->counters[0] += fp.total;
    ctx->counters[1] += fp.total * chgpu_type_size(col->type);
    *out = res;
    return CHGPU_OK;
}

extern "C" int chgpu_filter(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * mask, int64_t result_size_hint,
                            chgpu_col ** out, uint64_t * out_rows)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    (void)result_size_hint; // the device path sizes the result exactly from the scan; the hint only matters for CPU reserve()
    CHGPU_REQUIRE(ctx && col && mask && out && out_rows, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(mask->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS, "filter must be a UInt8 column");
    CHGPU_REQUIRE(col->rows == mask->rows, CHGPU_ERR_SIZES_MISMATCH, "Size of filter (%llu) doesn't match size of column (%llu)",
                  (unsigned long long)mask->rows, (unsigned long long)col->rows);
    if (col->rows == 0)
    {
        CHGPU_TRY(chgpu_col_new(ctx, col->type, 0, out));
        *out_rows = 0;
        return CHGPU_OK;
    }
    FilterPlan fp;
    CHGPU_TRY(filter_plan(ctx, mask, &fp));
    CHGPU_TRY(filter_apply(ctx, col, mask, fp, out));
    *out_rows = fp.total;
    return CHGPU_OK;
}

// filterToIndices (src/Columns/ColumnsCommon.cpp:384-470): the row numbers whose filter byte is non-zero, ascending -- the form
// ORDER BY ... LIMIT uses to name its candidate rows.  Same chunk counts + scan as chgpu_filter; one wave per 1024-row chunk.
__global__ __launch_bounds__(256) void k_filter_emit_indices(const u8 * __restrict__ mask, u64 n, const u64 * __restrict__ chunk_offsets, u64 n_chunks, u64 * __restrict__ out)
{
    const u32 lane = threadIdx.x & 63;
    const u64 wave0 = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const u64 n_waves = ((u64)gridDim.x * 256) >> 6;
    for (u64 chunk = wave0; chunk < n_chunks; chunk += n_waves)
    {
        u64 pos = chunk_offsets[chunk];
        for (u32 st = 0; st < CHUNK_ROWS / 64; ++st)
        {
            const u64 i = chunk * CHUNK_ROWS + st * 64 + lane;
            const bool keep = i < n && mask[i] != 0;
            const u64 b = __ballot(keep);
            if (keep)
                out[pos + mbcnt(b)] = i;
            pos += (u64)__popcll(b);
        }
    }
}

extern "C" int chgpu_filter_to_indices(chgpu_ctx * ctx, const chgpu_col * mask, chgpu_col ** indexes_u64, uint64_t * rows_out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && mask && indexes_u64 && rows_out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(mask->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS, "filter must be a UInt8 column");
    chgpu_col * res = nullptr;
    if (mask->rows == 0)
    {
        CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, 0, &res));
        *indexes_u64 = res;
        *rows_out = 0;
        return CHGPU_OK;
    }
    FilterPlan fp;
    CHGPU_TRY(filter_plan(ctx, mask, &fp));
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, fp.total, &res));
    if (fp.total)
    {
        hipLaunchKernelGGL(k_filter_emit_indices, dim3(fp.grid), dim3(256), 0, ctx->stream, (const u8 *)mask->data, fp.n, (const u64 *)fp.offsets, fp.n_chunks, (u64 *)res->data);
        ctx->counters[6] += 1;
        if (hipGetLastError() != hipSuccess)
        {
            chgpu_col_free(res);
            return chgpu_set_error(CHGPU_ERR_DEVICE, "filter_to_indices launch failed");
        }
    }
    *indexes_u64 = res;
    *rows_out = fp.total;
    return CHGPU_OK;
}

extern "C" int chgpu_filter_columns(chgpu_ctx * ctx, uint32_t n_cols, const chgpu_col * const * cols, const chgpu_col * mask,
                                    int64_t result_size_hint, chgpu_col ** outs, uint64_t * out_rows)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    (void)result_size_hint;
    CHGPU_REQUIRE(ctx && mask && out_rows && (n_cols == 0 || (cols && outs)), CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(mask->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS, "filter must be a UInt8 column");
    for (u32 k = 0; k < n_cols; ++k)
    {
        CHGPU_REQUIRE(cols[k], CHGPU_ERR_BAD_ARGUMENTS, "column %u is NULL", k);
        CHGPU_REQUIRE(cols[k]->rows == mask->rows, CHGPU_ERR_SIZES_MISMATCH, "Size of filter (%llu) doesn't match size of column (%llu)",
                      (unsigned long long)mask->rows, (unsigned long long)cols[k]->rows);
        outs[k] = nullptr;
    }
    FilterPlan fp;
    if (mask->rows)
        CHGPU_TRY(filter_plan(ctx, mask, &fp)); // the mask is counted and scanned ONCE for the whole Block
    auto fail = [&](int rc) {
        for (u32 q = 0; q < n_cols; ++q)
            if (outs[q])
            {
                chgpu_col_free(outs[q]);
                outs[q] = nullptr;
            }
        return rc;
    };
    // columns of one element width go through k_filter_scatter_multi up to six at a time: one read of the mask for all of them
    const bool no_multi = chgpu_opt(ctx, "tune_filter_no_multi", 0) != 0;
    const bool staged = filter_staged(ctx);
    std::vector<char> done(n_cols, 0);
    if (mask->rows && fp.total && !no_multi)
        for (size_t w : {(size_t)8, (size_t)4})
        {
            std::vector<u32> same;
            for (u32 k = 0; k < n_cols; ++k)
                if (chgpu_type_size(cols[k]->type) == w)
                    same.push_back(k);
            for (size_t b = 0; b + 1 < same.size();)
            {
                const u32 nc = (u32)(same.size() - b >= 6 ? 6 : same.size() - b);
                if (nc < 2)
                    break;
                for (u32 q = 0; q < nc; ++q)
                {
                    const int rc = chgpu_col_new(ctx, cols[same[b + q]]->type, fp.total, &outs[same[b + q]]);
                    if (rc != CHGPU_OK)
                        return fail(rc);
                }
#define FILTER_MULTI(T_, NC_)                                                                                                                  \
    do                                                                                                                                         \
    {                                                                                                                                          \
        FilterCols<T_, NC_> fc;                                                                                                                \
        for (u32 q = 0; q < NC_; ++q)                                                                                                          \
        {                                                                                                                                      \
            fc.data[q] = (const T_ *)cols[same[b + q]]->data;                                                                                  \
            fc.out[q] = (T_ *)outs[same[b + q]]->data;                                                                                         \
        }                                                                                                                                      \
        if (staged)                                                                                                                            \
            hipLaunchKernelGGL((k_filter_scatter_multi<T_, NC_, true>), dim3(fp.grid), dim3(256), 0, ctx->stream, fc, (const u8 *)mask->data, fp.n, (const u64 *)fp.offsets, fp.n_chunks); \
        else                                                                                                                                   \
            hipLaunchKernelGGL((k_filter_scatter_multi<T_, NC_, false>), dim3(fp.grid), dim3(256), 0, ctx->stream, fc, (const u8 *)mask->data, fp.n, (const u64 *)fp.offsets, fp.n_chunks); \
    } while (0)
                if (w == 8) { if (nc == 6) FILTER_MULTI(u64, 6); else if (nc == 5) FILTER_MULTI(u64, 5); else if (nc == 4) FILTER_MULTI(u64, 4); else if (nc == 3) FILTER_MULTI(u64, 3); else FILTER_MULTI(u64, 2); }
                else        { if (nc == 6) FILTER_MULTI(u32, 6); else if (nc == 5) FILTER_MULTI(u32, 5); else if (nc == 4) FILTER_MULTI(u32, 4); else if (nc == 3) FILTER_MULTI(u32, 3); else FILTER_MULTI(u32, 2); }
#undef FILTER_MULTI
                ctx->counters[6] += 1;
                if (hipGetLastError() != hipSuccess)
                    return fail(chgpu_set_error(CHGPU_ERR_DEVICE, "filter launch failed"));
                for (u32 q = 0; q < nc; ++q)
                {
                    done[same[b + q]] = 1;
                    ctx->counters[0] += fp.total;
                    ctx->counters[1] += fp.total * w;
                }
                b += nc;
            }
        }
    for (u32 k = 0; k < n_cols; ++k)
    {
        if (done[k])
            continue;
        const int rc = mask->rows ? filter_apply(ctx, cols[k], mask, fp, &outs[k]) : chgpu_col_new(ctx, cols[k]->type, 0, &outs[k]);
        if (rc != CHGPU_OK)
            return fail(rc);
    }
    *out_rows = fp.total;
    return CHGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// index (gather) and replicate
// ---------------------------------------------------------------------------------------------
template <typename T, typename I>
__global__ __launch_bounds__(256) void k_index(const T * __restrict__ data, const I * __restrict__ idx, u64 limit, u64 rows,
                                               int default_for_missing, T * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < limit; i += (u64)gridDim.x * 256)
    {
        const I j = idx[i];
        const bool missing = default_for_missing && j == (I)~(I)0;
        // out-of-range indexes are a caller bug in the reference too (no bounds check, ColumnVector.cpp:1137-1140);
        // here they read row 0 instead of faulting the GPU
        out[i] = missing ? T(0) : data[(u64)j < rows ? (u64)j : 0];
    }
}

extern "C" int chgpu_index(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * indexes, uint64_t limit,
                           int default_for_missing, chgpu_col ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && col && indexes && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(indexes->type == CHGPU_U64 || indexes->type == CHGPU_U32, CHGPU_ERR_BAD_ARGUMENTS, "indexes must be UInt64 or UInt32");
    if (limit == 0)
        limit = indexes->rows;
    CHGPU_REQUIRE(limit <= indexes->rows, CHGPU_ERR_SIZES_MISMATCH, "Size of indexes (%llu) is less than required (%llu)",
                  (unsigned long long)indexes->rows, (unsigned long long)limit); // ColumnVector.cpp:1126-1127
    CHGPU_REQUIRE(col->rows > 0 || limit == 0 || default_for_missing, CHGPU_ERR_BAD_ARGUMENTS, "index into an empty column");
    chgpu_col * res = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, col->type, limit, &res));
    if (limit)
    {
        const u32 grid = chgpu_grid_for(ctx, limit, 256, 8);
        const size_t es = chgpu_type_size(col->type);
#define IDX_LAUNCH(T, I) hipLaunchKernelGGL((k_index<T, I>), dim3(grid), dim3(256), 0, ctx->stream, (const T *)col->data, (const I *)indexes->data, limit, col->rows, default_for_missing, (T *)res->data)
        if (indexes->type == CHGPU_U64)
        {
            if (es == 8) IDX_LAUNCH(u64, u64); else if (es == 4) IDX_LAUNCH(u32, u64); else if (es == 2) IDX_LAUNCH(u16, u64); else IDX_LAUNCH(u8, u64);
        }
        else
        {
            if (es == 8) IDX_LAUNCH(u64, u32); else if (es == 4) IDX_LAUNCH(u32, u32); else if (es == 2) IDX_LAUNCH(u16, u32); else IDX_LAUNCH(u8, u32);
        }
#undef IDX_LAUNCH
        ctx->counters[6] += 1;
    }
    *out = res;
    return CHGPU_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void k_replicate(const T * __restrict__ data, const u64 * __restrict__ offsets, u64 n, T * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        const u64 end = offsets[i];
        const u64 begin = i ? offsets[i - 1] : 0;
        const T v = data[i];
        for (u64 o = begin; o < end; ++o)
            out[o] = v;
    }
}

extern "C" int chgpu_replicate(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * offsets, chgpu_col ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && col && offsets && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(offsets->type == CHGPU_U64, CHGPU_ERR_BAD_ARGUMENTS, "offsets must be UInt64");
    CHGPU_REQUIRE(offsets->rows == col->rows, CHGPU_ERR_SIZES_MISMATCH, "Size of offsets doesn't match size of column."); // ColumnVector.cpp:881-883
    u64 total = 0;
    if (col->rows)
        CHGPU_TRY(chgpu_read_back(ctx, (const u64 *)offsets->data + (col->rows - 1), &total, sizeof(total)));
    chgpu_col * res = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, col->type, total, &res));
    if (total)
    {
        const u32 grid = chgpu_grid_for(ctx, col->rows, 256, 8);
        switch (chgpu_type_size(col->type))
        {
            case 8: hipLaunchKernelGGL(k_replicate<u64>, dim3(grid), dim3(256), 0, ctx->stream, (const u64 *)col->data, (const u64 *)offsets->data, col->rows, (u64 *)res->data); break;
            case 4: hipLaunchKernelGGL(k_replicate<u32>, dim3(grid), dim3(256), 0, ctx->stream, (const u32 *)col->data, (const u64 *)offsets->data, col->rows, (u32 *)res->data); break;
            case 2: hipLaunchKernelGGL(k_replicate<u16>, dim3(grid), dim3(256), 0, ctx->stream, (const u16 *)col->data, (const u64 *)offsets->data, col->rows, (u16 *)res->data); break;
            default: hipLaunchKernelGGL(k_replicate<u8>, dim3(grid), dim3(256), 0, ctx->stream, (const u8 *)col->data, (const u64 *)offsets->data, col->rows, (u8 *)res->data); break;
        }
        ctx->counters[6] += 1;
    }
    *out = res;
    return CHGPU_OK;
}

// The same for every column of a Block (joinBlock replicates all left columns with one offsets_to_replicate,
// HashJoinMethodsImpl.h:186-194): one read-back of the total, and columns of one element width go through one kernel up to four at a
// time -- the 8-byte offsets are read once for all of them instead of once per column.
template <typename T, int NC>
struct ReplCols
{
    const T * data[NC];
    T * out[NC];
};
template <typename T, int NC>
__global__ __launch_bounds__(256) void k_replicate_multi(ReplCols<T, NC> c, const u64 * __restrict__ offsets, u64 n)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        const u64 end = offsets[i];
        const u64 begin = i ? offsets[i - 1] : 0;
        T v[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k)
            v[k] = c.data[k][i];
        for (u64 o = begin; o < end; ++o)
#pragma unroll
            for (int k = 0; k < NC; ++k)
                c.out[k][o] = v[k];
    }
}

extern "C" int chgpu_replicate_columns(chgpu_ctx * ctx, uint32_t n_cols, const chgpu_col * const * cols, const chgpu_col * offsets, chgpu_col ** outs)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && offsets && (n_cols == 0 || (cols && outs)), CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(offsets->type == CHGPU_U64, CHGPU_ERR_BAD_ARGUMENTS, "offsets must be UInt64");
    for (u32 k = 0; k < n_cols; ++k)
    {
        CHGPU_REQUIRE(cols[k], CHGPU_ERR_BAD_ARGUMENTS, "NULL column");
        CHGPU_REQUIRE(offsets->rows == cols[k]->rows, CHGPU_ERR_SIZES_MISMATCH, "Size of offsets doesn't match size of column."); // ColumnVector.cpp:881-883
        outs[k] = nullptr;
    }
    const u64 n = offsets->rows;
    u64 total = 0;
    if (n)
        CHGPU_TRY(chgpu_read_back(ctx, (const u64 *)offsets->data + (n - 1), &total, sizeof(total)));
    auto fail = [&](int rc) {
        for (u32 q = 0; q < n_cols; ++q)
            if (outs[q])
            {
                chgpu_col_free(outs[q]);
                outs[q] = nullptr;
            }
        return rc;
    };
    for (u32 k = 0; k < n_cols; ++k)
    {
        const int rc = chgpu_col_new(ctx, cols[k]->type, total, &outs[k]);
        if (rc != CHGPU_OK)
            return fail(rc);
    }
    if (!total)
        return CHGPU_OK;
    const u32 grid = chgpu_grid_for(ctx, n, 256, 8);
    std::vector<char> done(n_cols, 0);
    for (size_t w : {(size_t)8, (size_t)4})
    {
        std::vector<u32> same;
        for (u32 k = 0; k < n_cols; ++k)
            if (chgpu_type_size(cols[k]->type) == w)
                same.push_back(k);
        for (size_t b = 0; b + 1 < same.size();)
        {
            const u32 nc = (u32)(same.size() - b >= 4 ? 4 : same.size() - b);
            if (nc < 2)
                break;
#define REPL_MULTI(T_, NC_)                                                                                                        \
    do                                                                                                                             \
    {                                                                                                                              \
        ReplCols<T_, NC_> rc_;                                                                                                     \
        for (u32 q = 0; q < NC_; ++q)                                                                                              \
        {                                                                                                                          \
            rc_.data[q] = (const T_ *)cols[same[b + q]]->data;                                                                     \
            rc_.out[q] = (T_ *)outs[same[b + q]]->data;                                                                            \
        }                                                                                                                          \
        hipLaunchKernelGGL((k_replicate_multi<T_, NC_>), dim3(grid), dim3(256), 0, ctx->stream, rc_, (const u64 *)offsets->data, n); \
    } while (0)
            if (w == 8) { if (nc == 4) REPL_MULTI(u64, 4); else if (nc == 3) REPL_MULTI(u64, 3); else REPL_MULTI(u64, 2); }
            else        { if (nc == 4) REPL_MULTI(u32, 4); else if (nc == 3) REPL_MULTI(u32, 3); else REPL_MULTI(u32, 2); }
#undef REPL_MULTI
            ctx->counters[6] += 1;
            for (u32 q = 0; q < nc; ++q)
                done[same[b + q]] = 1;
            b += nc;
        }
    }
    for (u32 k = 0; k < n_cols; ++k)
    {
        if (done[k])
            continue;
        switch (chgpu_type_size(cols[k]->type))
        {
            case 8: hipLaunchKernelGGL(k_replicate<u64>, dim3(grid), dim3(256), 0, ctx->stream, (const u64 *)cols[k]->data, (const u64 *)offsets->data, n, (u64 *)outs[k]->data); break;
            case 4: hipLaunchKernelGGL(k_replicate<u32>, dim3(grid), dim3(256), 0, ctx->stream, (const u32 *)cols[k]->data, (const u64 *)offsets->data, n, (u32 *)outs[k]->data); break;
            case 2: hipLaunchKernelGGL(k_replicate<u16>, dim3(grid), dim3(256), 0, ctx->stream, (const u16 *)cols[k]->data, (const u64 *)offsets->data, n, (u16 *)outs[k]->data); break;
            default: hipLaunchKernelGGL(k_replicate<u8>, dim3(grid), dim3(256), 0, ctx->stream, (const u8 *)cols[k]->data, (const u64 *)offsets->data, n, (u8 *)outs[k]->data); break;
        }
        ctx->counters[6] += 1;
    }
    if (hipGetLastError() != hipSuccess)
        return fail(chgpu_set_error(CHGPU_ERR_DEVICE, "replicate launch failed"));
    return CHGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// and / arithmetic columns, and the fused multi-predicate filter + value expression + sum
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_and_u8(const u8 * __restrict__ a, const u8 * __restrict__ b, u64 n, u8 * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
        out[i] = a[i] & b[i]; // AndImpl::apply
}

extern "C" int chgpu_and(chgpu_ctx * ctx, const chgpu_col * a, const chgpu_col * b, chgpu_col ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && a && b && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(a->type == CHGPU_U8 && b->type == CHGPU_U8, CHGPU_ERR_NOT_IMPLEMENTED, "and() over non-UInt8 arguments: CPU path");
    CHGPU_REQUIRE(a->rows == b->rows, CHGPU_ERR_SIZES_MISMATCH, "arguments of function and have different sizes");
    chgpu_col * m = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U8, a->rows, &m));
    if (a->rows)
    {
        hipLaunchKernelGGL(k_and_u8, dim3(chgpu_grid_for(ctx, a->rows, 256, 8)), dim3(256), 0, ctx->stream, (const u8 *)a->data, (const u8 *)b->data, a->rows, (u8 *)m->data);
        ctx->counters[6] += 1;
    }
    *out = m;
    return CHGPU_OK;
}

// Type of sum(a OP b): the operation's result type (NumberTraits.h:73-87) summed (Int64 for signed, UInt64 for unsigned).
static int arith_sum_type(int value_op, int a_type, int b_type);
// Result type of a OP b as a column; two 1-byte operands promote to a 2-byte type this library does not carry: -1.
static int arith_result_type(int value_op, int a_type, int b_type)
{
    if (chgpu_type_size(a_type) == 1 && chgpu_type_size(b_type) == 1)
        return -1;
    return arith_sum_type(value_op, a_type, b_type);
}
static int arith_sum_type(int value_op, int a_type, int b_type)
{
    if (!chgpu_type_is_int(a_type) || !chgpu_type_is_int(b_type))
        return -1;
    if (a_type > CHGPU_I32 || b_type > CHGPU_I32)
        return -1; // UInt16 / Int16 / Int8 operands: arithmetic stays on the CPU path
    if (value_op == CHGPU_VAL_MINUS)
        return CHGPU_I64; // ResultOfSubtraction: always signed (NumberTraits.h:81-87)
    if (value_op == CHGPU_VAL_MUL || value_op == CHGPU_VAL_PLUS)
        return (chgpu_type_is_signed(a_type) || chgpu_type_is_signed(b_type)) ? CHGPU_I64 : CHGPU_U64; // :73-79
    return -1;
}

template <typename T>
__device__ __forceinline__ u64 ext64(T v)
{
    if constexpr (std::is_signed<T>::value)
        return (u64)(i64)v;
    else
        return (u64)v;
}

__device__ __forceinline__ u64 apply_val(int op, u64 x, u64 y)
{
    return op == CHGPU_VAL_MUL ? x * y : (op == CHGPU_VAL_PLUS ? x + y : x - y); // static_cast<Result>(a) OP b, wrap-around
}

template <typename TA, typename TB>
__global__ __launch_bounds__(256) void k_arith(const TA * __restrict__ a, const TB * __restrict__ b, u64 n, int op, u64 * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
        out[i] = apply_val(op, ext64(a[i]), ext64(b[i]));
}

extern "C" int chgpu_arith(chgpu_ctx * ctx, int value_op, const chgpu_col * a, const chgpu_col * b, chgpu_col ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && a && b && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    const int rt = arith_result_type(value_op, a->type, b->type);
    CHGPU_REQUIRE(rt >= 0, CHGPU_ERR_NOT_IMPLEMENTED, "arithmetic on these types/operator: CPU path");
    CHGPU_REQUIRE(a->rows == b->rows, CHGPU_ERR_SIZES_MISMATCH, "arguments of an arithmetic function have different sizes");
    CHGPU_REQUIRE(a->type == b->type, CHGPU_ERR_NOT_IMPLEMENTED, "mixed-type arithmetic: CPU path");
    chgpu_col * r = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, rt, a->rows, &r));
    const u64 n = a->rows;
    if (n)
    {
        const u32 grid = chgpu_grid_for(ctx, n, 256, 8);
#define AR(T) hipLaunchKernelGGL((k_arith<T, T>), dim3(grid), dim3(256), 0, ctx->stream, (const T *)a->data, (const T *)b->data, n, value_op, (u64 *)r->data)
        switch (a->type)
        {
            case CHGPU_I64: AR(i64); break;
            case CHGPU_U64: AR(u64); break;
            case CHGPU_U32: AR(u32); break;
            case CHGPU_I32: AR(i32); break;
            default: AR(u8); break;
        }
#undef AR
        ctx->counters[6] += 1;
    }
    *out = r;
    return CHGPU_OK;
}

__device__ __forceinline__ u32 chgpu_type_size_dev(int type) { return type == CHGPU_U8 ? 1u : (type == CHGPU_U32 || type == CHGPU_I32) ? 4u : 8u; }

static constexpr u32 EX_MAX_COLS = 4;
static constexpr u32 EX_MAX_PREDS = 8;

// range predicate in the column's own width: 3 VALU ops per element for 4-byte columns (xor, sub, cmp) instead of the
// ~8 a 64-bit key costs; lo/span/flip are folded on the host exactly like IntRangePred's
struct ExprPred
{
    u64 lo, span, flip;
    u32 invert;
    u32 col;
};

struct ExprSpec
{
    u32 n_cols, n_preds;
    int col_type[EX_MAX_COLS]; // used by the mixed-width kernel
    const void * col[EX_MAX_COLS];
    ExprPred pred[EX_MAX_PREDS];
    int value_op;
    u32 val_a, val_b;
};

template <typename T>
__device__ __forceinline__ bool expr_pass(const ExprPred & p, T x)
{
    if constexpr (sizeof(T) <= 4)
    {
        const u32 key = (u32)x ^ (u32)p.flip; // the sign flip for signed 4-byte types is folded to bit 31
        return ((key - (u32)p.lo) <= (u32)p.span) != (p.invert != 0);
    }
    else
    {
        const u64 key = (u64)x ^ p.flip;
        return ((key - p.lo) <= p.span) != (p.invert != 0);
    }
}

// One pass: up to 4 columns of one element type, <= 8 range predicates and-ed, value = column or a binary op of two columns.
// Work is ordered column-major: for column k (static) loop over the predicates that test it (wave-uniform branch), apply each
// to all E elements held in registers — the predicate constants are read once per (column, predicate), no runtime-indexed
// register arrays (they would go to scratch) and no select chains.
template <typename T>
__global__ __launch_bounds__(FS_THREADS) void k_expr_filter_sum(ExprSpec sp, u64 n, u64 * __restrict__ part_sum, u64 * __restrict__ part_cnt)
{
#ifndef EX_UNROLL
#define EX_UNROLL 4 // measured with 3 workgroups/CU on the 4-column Q1.1 shape: U=1 4.3, U=2 6.1, U=3 6.6, U=4 6.7 TB/s
#endif
    constexpr int VEC = 16 / sizeof(T);
    typedef Vec<T, VEC> V;
    constexpr int UNROLL = EX_UNROLL;
    constexpr int E = VEC * UNROLL; // elements per lane per iteration
    const u64 nvec = n / VEC;
    u64 s = 0, c = 0;

    auto reduce_rows = [&](const T (&x0)[E], const T (&x1)[E], const T (&x2)[E], const T (&x3)[E], int n_elem) {
        bool pass[E];
        u64 va[E], vb[E];
#pragma unroll
        for (int e = 0; e < E; ++e)
        {
            pass[e] = e < n_elem;
            va[e] = 0;
            vb[e] = 0;
        }
        auto column = [&](u32 k, const T (&x)[E]) {
            // (keeping the predicate constants in registers for the whole kernel, as the narrow mixed-width kernel does, was
            //  measured SLOWER here: 1.04 -> 1.24 ms at 4e8 rows; this kernel holds 16 elements per lane and is HBM-bound)
            for (u32 q = 0; q < sp.n_preds; ++q)
                if (sp.pred[q].col == k)
                {
                    const ExprPred pr = sp.pred[q];
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        pass[e] = pass[e] && expr_pass<T>(pr, x[e]);
                }
            if (sp.val_a == k)
            {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    va[e] = ext64(x[e]);
            }
            if (sp.value_op != CHGPU_VAL_COL && sp.val_b == k)
            {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    vb[e] = ext64(x[e]);
            }
        };
        column(0, x0);
        if (sp.n_cols > 1) column(1, x1);
        if (sp.n_cols > 2) column(2, x2);
        if (sp.n_cols > 3) column(3, x3);
#pragma unroll
        for (int e = 0; e < E; ++e)
        {
            const u64 v = sp.value_op == CHGPU_VAL_COL ? va[e] : apply_val(sp.value_op, va[e], vb[e]);
            s += pass[e] ? v : 0;
            c += pass[e] ? 1 : 0;
        }
    };

    constexpr u64 CHUNK = (u64)UNROLL * FS_THREADS;
    const u64 n_chunks = nvec / CHUNK;
    const V * __restrict__ c0 = (const V *)sp.col[0];
    const V * __restrict__ c1 = (const V *)sp.col[1];
    const V * __restrict__ c2 = (const V *)sp.col[2];
    const V * __restrict__ c3 = (const V *)sp.col[3];
    for (u64 ch = blockIdx.x; ch < n_chunks; ch += gridDim.x)
    {
        const u64 base = ch * CHUNK + threadIdx.x;
        V y0[UNROLL], y1[UNROLL], y2[UNROLL], y3[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
        {
            const u64 i = base + (u64)k * FS_THREADS;
            y0[k] = load_stream(&c0[i]);
            if (sp.n_cols > 1) y1[k] = load_stream(&c1[i]);
            if (sp.n_cols > 2) y2[k] = load_stream(&c2[i]);
            if (sp.n_cols > 3) y3[k] = load_stream(&c3[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        T x0[E], x1[E], x2[E], x3[E];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e)
            {
                x0[k * VEC + e] = y0[k].v[e];
                x1[k * VEC + e] = y1[k].v[e];
                x2[k * VEC + e] = y2[k].v[e];
                x3[k * VEC + e] = y3[k].v[e];
            }
        reduce_rows(x0, x1, x2, x3, E);
    }
    const u64 tid = (u64)blockIdx.x * FS_THREADS + threadIdx.x;
    const u64 stride = (u64)gridDim.x * FS_THREADS;
    const T * s0 = (const T *)sp.col[0];
    const T * s1 = (const T *)sp.col[1];
    const T * s2 = (const T *)sp.col[2];
    const T * s3 = (const T *)sp.col[3];
    for (u64 r = n_chunks * CHUNK * VEC + tid; r < n; r += stride)
    {
        T x0[E] = {}, x1[E] = {}, x2[E] = {}, x3[E] = {};
        x0[0] = s0[r];
        x1[0] = s1[r];
        x2[0] = s2[r];
        x3[0] = s3[r];
        reduce_rows(x0, x1, x2, x3, 1);
    }

    __shared__ u64 lds_s[FS_THREADS / WAVE];
    __shared__ u64 lds_c[FS_THREADS / WAVE];
    s = wave_reduce_add_u64(s);
    c = wave_reduce_add_u64(c);
    if ((threadIdx.x & 63) == 0)
    {
        lds_s[threadIdx.x >> 6] = s;
        lds_c[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        u64 ss = 0, cc = 0;
        for (int w = 0; w < FS_THREADS / WAVE; ++w)
        {
            ss += lds_s[w];
            cc += lds_c[w];
        }
        part_sum[blockIdx.x] = ss;
        part_cnt[blockIdx.x] = cc;
    }
}

// Columns of DIFFERENT integer widths, general case (any mix of 1-, 4- and 8-byte columns): one row per lane and step, four
// steps in flight; every column is fetched through a wave-uniform type switch and widened to 64 bits, predicates are the
// 64-bit range tests.  Simple on purpose: the mix with 8-byte columns is rare, the 1/4-byte mix has its own kernel below (a
// unit-vectorised generic version spilled to scratch and ran 3-7x slower than this one).
#ifndef EXM_UNROLL
#define EXM_UNROLL 4
#endif
__device__ __forceinline__ u64 exm_load_ext(const void * col, int type, u64 i)
{
    switch (type) // wave-uniform
    {
        case CHGPU_U8: return ((const u8 *)col)[i];
        case CHGPU_U32: return ((const u32 *)col)[i];
        case CHGPU_I32: return (u64)(i64)((const i32 *)col)[i];
        default: return ((const u64 *)col)[i];
    }
}

__global__ __launch_bounds__(FS_THREADS) void k_expr_filter_sum_mixed(ExprSpec sp, u64 n, u64 * __restrict__ part_sum, u64 * __restrict__ part_cnt)
{
    constexpr int U = EXM_UNROLL;
    u64 s = 0, c = 0;
    const u64 stride = (u64)gridDim.x * FS_THREADS;
    // bit k: column k is 1 or 4 bytes wide (constant indices only: a run-time index into the by-value spec lands it in scratch)
    const u32 narrow_mask = (chgpu_type_size_dev(sp.col_type[0]) <= 4 ? 1u : 0u) | (chgpu_type_size_dev(sp.col_type[1]) <= 4 ? 2u : 0u) |
                            (chgpu_type_size_dev(sp.col_type[2]) <= 4 ? 4u : 0u) | (chgpu_type_size_dev(sp.col_type[3]) <= 4 ? 8u : 0u);
    // 64-bit key domain of each column: signed types are compared after flipping bit 63 of the sign-extended value; the host
    // folded 4- and 1-byte columns' predicates into their own 32-bit key space (ExprPred), so widen those constants back
    for (u64 i0 = (u64)blockIdx.x * FS_THREADS + threadIdx.x; i0 < n; i0 += stride * U)
    {
        u64 x0[U], x1[U], x2[U], x3[U]; // one array per column: constant indices only, so they stay in registers
        bool pass[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const u64 i = i0 + (u64)u * stride;
            pass[u] = i < n;
            const u64 ic = pass[u] ? i : n - 1;
            x0[u] = exm_load_ext(sp.col[0], sp.col_type[0], ic);
            x1[u] = sp.n_cols > 1 ? exm_load_ext(sp.col[1], sp.col_type[1], ic) : 0;
            x2[u] = sp.n_cols > 2 ? exm_load_ext(sp.col[2], sp.col_type[2], ic) : 0;
            x3[u] = sp.n_cols > 3 ? exm_load_ext(sp.col[3], sp.col_type[3], ic) : 0;
        }
        // column-major, with wave-uniform branches on the column index (a select chain over x0..x3 is turned into a scratch
        // lookup table by the compiler: 27 ms instead of 4)
        u64 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            va[u] = 0, vb[u] = 0;
        auto column = [&](u32 k, const u64 (&x)[U]) {
            const bool narrow = (narrow_mask >> k) & 1;
#pragma unroll
            for (u32 q = 0; q < EX_MAX_PREDS; ++q)
            {
                if (q >= sp.n_preds || sp.pred[q].col != k)
                    continue;
                const ExprPred pr = sp.pred[q];
#pragma unroll
                for (int u = 0; u < U; ++u)
                {
                    // narrow columns: the 32-bit key test on the low word (exactly expr_pass<u32/i32/u8>); wide: the 64-bit one
                    const bool ok = narrow ? ((((u32)x[u] ^ (u32)pr.flip) - (u32)pr.lo) <= (u32)pr.span) : (((x[u] ^ pr.flip) - pr.lo) <= pr.span);
                    pass[u] = pass[u] && (ok != (pr.invert != 0));
                }
            }
            if (sp.val_a == k)
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    va[u] = x[u];
            }
            if (sp.value_op != CHGPU_VAL_COL && sp.val_b == k)
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    vb[u] = x[u];
            }
        };
        column(0, x0);
        if (sp.n_cols > 1) column(1, x1);
        if (sp.n_cols > 2) column(2, x2);
        if (sp.n_cols > 3) column(3, x3);
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const u64 v = sp.value_op == CHGPU_VAL_COL ? va[u] : apply_val(sp.value_op, va[u], vb[u]);
            s += pass[u] ? v : 0;
            c += pass[u] ? 1 : 0;
        }
    }

    __shared__ u64 lds_s[FS_THREADS / WAVE];
    __shared__ u64 lds_c[FS_THREADS / WAVE];
    s = wave_reduce_add_u64(s);
    c = wave_reduce_add_u64(c);
    if ((threadIdx.x & 63) == 0)
    {
        lds_s[threadIdx.x >> 6] = s;
        lds_c[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        u64 ss = 0, cc = 0;
        for (int w = 0; w < FS_THREADS / WAVE; ++w)
        {
            ss += lds_s[w];
            cc += lds_c[w];
        }
        part_sum[blockIdx.x] = ss;
        part_cnt[blockIdx.x] = cc;
    }
}

// The common mixed case -- every column 1 or 4 bytes wide (SSB lineorder) -- with ALL loads of an iteration issued before the
// first use, like the same-type kernel: 4 registers per (4-byte column, unit), 1 per (1-byte column, unit).  The generic
// mixed kernel above loads column after column (187 VGPRs, 2 waves/SIMD: 6.7 ms for the 1e9-row Q1.1 shape).
// WMASK: bit k set = column k is 4 bytes wide (else 1 byte).  The widths are compile-time so that the loads of all four
// columns are straight-line code: with a run-time width test around them hipcc drained the load queue (vmcnt(0)) at every
// join and the kernel ran at the speed of four dependent loads per iteration (6.7 ms for 1e9 rows).  Absent columns alias
// column 0 (their loads hit L1) so that no column-count branch surrounds a load either.
template <u32 WMASK>
__global__ __launch_bounds__(FS_THREADS) void k_expr_filter_sum_narrow(ExprSpec sp, u64 n, u64 * __restrict__ part_sum, u64 * __restrict__ part_cnt)
{
#ifndef EXN_UNROLL
#define EXN_UNROLL 2 // 4 units per lane needed 256 VGPRs (1 wave/SIMD)
#endif
    constexpr int U = EXN_UNROLL, E = 4 * U;
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    u64 s = 0, c = 0;
    const u64 n_units = n / 4;
    constexpr u64 CHUNK = (u64)U * FS_THREADS;
    const u64 n_chunks = n_units / CHUNK;

    // predicate constants live in wave-uniform registers for the whole kernel (constant indices after unrolling): fetching
    // sp.pred[q] inside the row loop cost an s_load + wait per (column, predicate) and iteration -- 6.0 ms instead of 2.x
    u32 p_lo[EX_MAX_PREDS], p_span[EX_MAX_PREDS], p_flip[EX_MAX_PREDS], p_inv[EX_MAX_PREDS], p_col[EX_MAX_PREDS];
#pragma unroll
    for (u32 q = 0; q < EX_MAX_PREDS; ++q)
    {
        const bool on = q < sp.n_preds;
        p_lo[q] = on ? (u32)sp.pred[q].lo : 0;
        p_span[q] = on ? (u32)sp.pred[q].span : 0;
        p_flip[q] = on ? (u32)sp.pred[q].flip : 0;
        p_inv[q] = on ? sp.pred[q].invert : 0;
        p_col[q] = on ? sp.pred[q].col : 0xFFu;
    }
    auto load_col = [&](auto kc, u64 unit0, v4u (&raw)[U]) {
        constexpr u32 k = decltype(kc)::value;
        if constexpr (((WMASK >> k) & 1) == 0)
        {
#pragma unroll
            for (int u = 0; u < U; ++u)
                raw[u].x = __builtin_nontemporal_load((const u32 *)sp.col[k] + unit0 + (u64)u * FS_THREADS);
        }
        else
        {
#pragma unroll
            for (int u = 0; u < U; ++u)
                raw[u] = __builtin_nontemporal_load((const v4u *)sp.col[k] + unit0 + (u64)u * FS_THREADS);
        }
    };
    // decode + test + extract for one column; x holds the zero/sign-extended 32-bit pattern of each element
    auto column_t = [&](auto tag, u32 k, const u32 (&x)[E], bool (&pass)[E], u32 (&va)[E], u32 (&vb)[E]) {
        (void)tag; // the element's signedness only matters for the value (finish()); the key test is width-native u32
#pragma unroll
        for (u32 q = 0; q < EX_MAX_PREDS; ++q)
            if (p_col[q] == k)
            {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    pass[e] = pass[e] && ((((x[e] ^ p_flip[q]) - p_lo[q]) <= p_span[q]) != (p_inv[q] != 0));
            }
        if (sp.val_a == k)
        {
#pragma unroll
            for (int e = 0; e < E; ++e)
                va[e] = x[e]; // widened in finish() by the column's signedness
        }
        if (sp.value_op != CHGPU_VAL_COL && sp.val_b == k)
        {
#pragma unroll
            for (int e = 0; e < E; ++e)
                vb[e] = x[e];
        }
    };
    auto column = [&](u32 k, const v4u (&raw)[U], bool (&pass)[E], u32 (&va)[E], u32 (&vb)[E]) {
        u32 x[E];
        if (sp.col_type[k] == CHGPU_U8)
        {
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                const u32 w = raw[u].x;
                x[4 * u] = w & 0xFF, x[4 * u + 1] = (w >> 8) & 0xFF, x[4 * u + 2] = (w >> 16) & 0xFF, x[4 * u + 3] = w >> 24;
            }
            column_t((u8)0, k, x, pass, va, vb);
        }
        else
        {
#pragma unroll
            for (int u = 0; u < U; ++u)
                x[4 * u] = raw[u].x, x[4 * u + 1] = raw[u].y, x[4 * u + 2] = raw[u].z, x[4 * u + 3] = raw[u].w;
            if (sp.col_type[k] == CHGPU_I32)
                column_t((i32)0, k, x, pass, va, vb);
            else
                column_t((u32)0, k, x, pass, va, vb);
        }
    };
    const bool a_signed = sp.col_type[sp.val_a] == CHGPU_I32, b_signed = sp.col_type[sp.val_b < EX_MAX_COLS ? sp.val_b : 0] == CHGPU_I32;
    auto finish = [&](const bool (&pass)[E], const u32 (&va)[E], const u32 (&vb)[E]) {
#pragma unroll
        for (int e = 0; e < E; ++e)
        {
            const u64 xa = a_signed ? (u64)(i64)(i32)va[e] : (u64)va[e], xb = b_signed ? (u64)(i64)(i32)vb[e] : (u64)vb[e];
            const u64 v = sp.value_op == CHGPU_VAL_COL ? xa : apply_val(sp.value_op, xa, xb);
            s += pass[e] ? v : 0;
            c += pass[e] ? 1 : 0;
        }
    };
    for (u64 ch = blockIdx.x; ch < n_chunks; ch += gridDim.x)
    {
        const u64 unit0 = ch * CHUNK + threadIdx.x;
        v4u r0[U] = {}, r1[U] = {}, r2[U] = {}, r3[U] = {};
        load_col(std::integral_constant<u32, 0>{}, unit0, r0);
        load_col(std::integral_constant<u32, 1>{}, unit0, r1);
        load_col(std::integral_constant<u32, 2>{}, unit0, r2);
        load_col(std::integral_constant<u32, 3>{}, unit0, r3);
        __builtin_amdgcn_sched_barrier(0);
        bool pass[E];
        u32 va[E], vb[E];
#pragma unroll
        for (int e = 0; e < E; ++e)
            pass[e] = true, va[e] = 0, vb[e] = 0;
        column(0, r0, pass, va, vb);
        if (sp.n_cols > 1) column(1, r1, pass, va, vb);
        if (sp.n_cols > 2) column(2, r2, pass, va, vb);
        if (sp.n_cols > 3) column(3, r3, pass, va, vb);
        finish(pass, va, vb);
    }
    // tail rows one by one
    const u64 tid = (u64)blockIdx.x * FS_THREADS + threadIdx.x;
    const u64 stride = (u64)gridDim.x * FS_THREADS;
    for (u64 r = n_chunks * CHUNK * 4 + tid; r < n; r += stride)
    {
        bool pass[E];
        u32 va[E], vb[E];
#pragma unroll
        for (int e = 0; e < E; ++e)
            pass[e] = e == 0, va[e] = 0, vb[e] = 0;
        for (u32 k = 0; k < sp.n_cols; ++k)
        {
            v4u raw[U] = {};
            raw[0].x = sp.col_type[k] == CHGPU_U8 ? (u32)((const u8 *)sp.col[k])[r] : ((const u32 *)sp.col[k])[r];
            column(k, raw, pass, va, vb);
        }
        finish(pass, va, vb);
    }

    __shared__ u64 lds_s[FS_THREADS / WAVE];
    __shared__ u64 lds_c[FS_THREADS / WAVE];
    s = wave_reduce_add_u64(s);
    c = wave_reduce_add_u64(c);
    if ((threadIdx.x & 63) == 0)
    {
        lds_s[threadIdx.x >> 6] = s;
        lds_c[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        u64 ss = 0, cc = 0;
        for (int w = 0; w < FS_THREADS / WAVE; ++w)
        {
            ss += lds_s[w];
            cc += lds_c[w];
        }
        part_sum[blockIdx.x] = ss;
        part_cnt[blockIdx.x] = cc;
    }
}

extern "C" int chgpu_expr_filter_sum(chgpu_ctx * ctx, uint32_t n_cols, const chgpu_col * const * cols, uint32_t n_preds,
                                     const uint32_t * pred_col, const int * pred_op, const int * pred_scalar_type,
                                     const uint64_t * pred_scalar_bits, int value_op, uint32_t val_a, uint32_t val_b,
                                     int * result_type_out, void * sum_out, uint64_t * count_out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && cols && sum_out && count_out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(n_cols >= 1 && n_cols <= EX_MAX_COLS, CHGPU_ERR_NOT_IMPLEMENTED, "fused expression over more than %u columns: CPU path", EX_MAX_COLS);
    CHGPU_REQUIRE(n_preds <= EX_MAX_PREDS, CHGPU_ERR_NOT_IMPLEMENTED, "more than %u predicates: CPU path", EX_MAX_PREDS);
    CHGPU_REQUIRE(n_preds == 0 || (pred_col && pred_op && pred_scalar_type && pred_scalar_bits), CHGPU_ERR_BAD_ARGUMENTS, "NULL predicate arrays");
    CHGPU_REQUIRE(value_op >= CHGPU_VAL_COL && value_op <= CHGPU_VAL_MINUS, CHGPU_ERR_BAD_ARGUMENTS, "unknown value operator %d", value_op);
    CHGPU_REQUIRE(val_a < n_cols && (value_op == CHGPU_VAL_COL || val_b < n_cols), CHGPU_ERR_BAD_ARGUMENTS, "value column index out of range");
    ExprSpec sp;
    memset(&sp, 0, sizeof(sp));
    sp.n_cols = n_cols;
    sp.n_preds = n_preds;
    const int type0 = cols[0] ? cols[0]->type : -1;
    u64 n = cols[0] ? cols[0]->rows : 0;
    bool aligned = true, one_type = cols[0] && cols[0]->type != CHGPU_U8; // all-UInt8: the same-type kernel would hold 64 elements per lane
                                                                           // (256 VGPRs + scratch); the narrow kernel takes it
    for (u32 k = 0; k < EX_MAX_COLS; ++k)
    {
        const chgpu_col * cc = cols[k < n_cols ? k : 0];
        CHGPU_REQUIRE(cc, CHGPU_ERR_BAD_ARGUMENTS, "column %u is NULL", k);
        CHGPU_REQUIRE(chgpu_type_is_int(cc->type) && cc->type <= CHGPU_I32, CHGPU_ERR_NOT_IMPLEMENTED, "fused expression over Float64 / 2-byte / Int8 columns: CPU path");
        CHGPU_REQUIRE(cc->rows == n, CHGPU_ERR_SIZES_MISMATCH, "Sizes of columns doesn't match");
        one_type = one_type && cc->type == type0;
        sp.col[k] = cc->data;
        sp.col_type[k] = cc->type;
        aligned = aligned && (((uintptr_t)cc->data) & 15) == 0;
    }
    CHGPU_REQUIRE(aligned, CHGPU_ERR_NOT_IMPLEMENTED, "fused expression needs 16-byte aligned columns");
    for (u32 k = 0; k < n_preds; ++k)
    {
        CHGPU_REQUIRE(pred_col[k] < n_cols, CHGPU_ERR_BAD_ARGUMENTS, "predicate %u refers to column %u of %u", k, pred_col[k], n_cols);
        const int type = cols[pred_col[k]]->type; // predicates are folded in the width of the column they test
        CmpSpec cs;
        CHGPU_TRY(make_cmp_spec(type, pred_op[k], pred_scalar_type[k], &pred_scalar_bits[k], &cs));
        // make_cmp_spec folds the comparison over the 64-bit extension of the column's signedness class; a 4-byte (or
        // 1-byte) column only ever presents values of its own range, so clamping the key range to that sub-range and
        // dropping the upper bits gives the same test in 32-bit arithmetic
        ExprPred ep;
        ep.col = pred_col[k];
        ep.invert = cs.ip.invert;
        if (chgpu_type_size(type) == 8)
        {
            ep.lo = cs.ip.lo;
            ep.span = cs.ip.span;
            ep.flip = cs.ip.flip;
        }
        else
        {
            const bool sgn = chgpu_type_is_signed(type);
            // 64-bit key space: signed -> value + 2^63, unsigned -> value.  The column's values occupy [vmin, vmax] there.
            const u64 vmin = sgn ? (1ull << 63) - (1ull << 31) : 0;
            const u64 vmax = sgn ? (1ull << 63) + ((1ull << 31) - 1) : (chgpu_type_size(type) == 4 ? 0xFFFFFFFFull : 0xFFull);
            u64 lo = cs.ip.lo, hi = cs.ip.lo + cs.ip.span; // inclusive key range (no wrap: span <= max - lo)
            bool empty = hi < vmin || lo > vmax;
            if (lo < vmin) lo = vmin;
            if (hi > vmax) hi = vmax;
            if (empty)
            {
                // nothing in range: full 32-bit range with the inversion flipped
                ep.lo = 0;
                ep.span = 0xFFFFFFFFull;
                ep.invert = cs.ip.invert ? 0u : 1u;
            }
            else
            {
                // 32-bit key = value ^ 0x80000000 for signed (order-preserving), value for unsigned
                ep.lo = sgn ? (lo - vmin) : lo;
                ep.span = hi - lo;
            }
            ep.flip = sgn ? 0x80000000ull : 0;
        }
        sp.pred[k] = ep;
    }
    sp.value_op = value_op;
    sp.val_a = val_a;
    sp.val_b = val_b;
    const int type_a = cols[val_a]->type, type_b = value_op == CHGPU_VAL_COL ? type_a : cols[val_b]->type;
    // the kernels compute in 64 bits (operands sign/zero-extended): exact for every result type up to 8 bytes, and a 2-byte
    // result (UInt8 op UInt8) cannot overflow, so the 64-bit sum is the reference's sum in every supported case
    const int rt = value_op == CHGPU_VAL_COL ? chgpu_sum_result_type(type_a) : arith_sum_type(value_op, type_a, type_b);
    CHGPU_REQUIRE(rt >= 0, CHGPU_ERR_NOT_IMPLEMENTED, "value expression: CPU path");
    if (result_type_out)
        *result_type_out = rt;

    const int type = type0;
    const u32 vecw = one_type ? 16 / (u32)chgpu_type_size(type) : 4;
    bool narrow = true; // every column 1 or 4 bytes wide
    for (u32 k = 0; k < n_cols; ++k)
        narrow = narrow && chgpu_type_size(cols[k]->type) <= 4;
    // workgroups per CU (measured on the 4-column Q1.1 shape): same-type kernel 3; narrow mixed kernel 6 (86 VGPRs, 2 units
    // per lane: 2.36 ms vs 3.02 ms with 3; 3-4 units per lane 2.7 ms)
    const u32 ex_wg = tune_env(ctx, "tune_expr_wg", 3), exn_wg = tune_env(ctx, "tune_exprn_wg", 6);
    const u32 grid = chgpu_grid_for(ctx, (n + vecw - 1) / vecw, FS_THREADS, (!one_type && narrow) ? exn_wg : ex_wg);
    void * scratch = nullptr;
    const u32 grid_cap = (u32)ctx->num_cus * 8;
    CHGPU_TRY(chgpu_scratch(ctx, (size_t)grid_cap * 2 * sizeof(u64) + 64, &scratch));
    u64 * part_sum = (u64 *)scratch;
    u64 * part_cnt = part_sum + grid;
    u64 * result_dev = (u64 *)((char *)scratch + (size_t)grid_cap * 2 * sizeof(u64));
    if (!one_type && narrow)
    {
        u32 wmask = 0;
        for (u32 k = 0; k < EX_MAX_COLS; ++k)
            wmask |= (chgpu_type_size(cols[k < n_cols ? k : 0]->type) == 4 ? 1u : 0u) << k;
#define EXN(M) case M: hipLaunchKernelGGL(k_expr_filter_sum_narrow<M>, dim3(grid), dim3(FS_THREADS), 0, ctx->stream, sp, n, part_sum, part_cnt); break;
        switch (wmask)
        {
            EXN(0) EXN(1) EXN(2) EXN(3) EXN(4) EXN(5) EXN(6) EXN(7) EXN(8) EXN(9) EXN(10) EXN(11) EXN(12) EXN(13) EXN(14) EXN(15)
        }
#undef EXN
    }
    else if (!one_type)
        hipLaunchKernelGGL(k_expr_filter_sum_mixed, dim3(grid), dim3(FS_THREADS), 0, ctx->stream, sp, n, part_sum, part_cnt);
    else
    switch (type)
    {
        case CHGPU_I64: hipLaunchKernelGGL(k_expr_filter_sum<i64>, dim3(grid), dim3(FS_THREADS), 0, ctx->stream, sp, n, part_sum, part_cnt); break;
        case CHGPU_U64: hipLaunchKernelGGL(k_expr_filter_sum<u64>, dim3(grid), dim3(FS_THREADS), 0, ctx->stream, sp, n, part_sum, part_cnt); break;
        case CHGPU_U32: hipLaunchKernelGGL(k_expr_filter_sum<u32>, dim3(grid), dim3(FS_THREADS), 0, ctx->stream, sp, n, part_sum, part_cnt); break;
        case CHGPU_I32: hipLaunchKernelGGL(k_expr_filter_sum<i32>, dim3(grid), dim3(FS_THREADS), 0, ctx->stream, sp, n, part_sum, part_cnt); break;
        default: hipLaunchKernelGGL(k_expr_filter_sum<u8>, dim3(grid), dim3(FS_THREADS), 0, ctx->stream, sp, n, part_sum, part_cnt); break;
    }
    hipLaunchKernelGGL(k_filter_sum_finish<false>, dim3(1), dim3(256), 0, ctx->stream, part_sum, part_cnt, grid, result_dev);
    ctx->counters[6] += 2;
    ctx->counters[5] += n;
    CHGPU_HIP(hipGetLastError());
    u64 res[2];
    CHGPU_TRY(chgpu_read_back(ctx, result_dev, res, sizeof(res)));
    memcpy(sum_out, &res[0], 8);
    *count_out = res[1];
    ctx->counters[0] += res[1];
    return CHGPU_OK;
}
