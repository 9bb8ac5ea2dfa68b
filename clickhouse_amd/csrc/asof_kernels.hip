// asof_kernels.hip — ASOF joins (JoinStrictness::Asof; INNER and LEFT, joinDispatch.h:66-67).
//
// Reference: the right table is a hash map key -> SortedLookupVector of (asof value, row) (src/Interpreters/RowRefs.cpp:40-215); a left row
// finds its key, then the closest right row under the join's inequality by a binary search of that vector (findAsof -> boundSearch,
// :100-166; HashJoinMethodsImpl.h:462-478).  At most one right row per left row; INNER drops the left rows without one (need_filter),
// LEFT keeps them with a default row (JoinFeatures.h:28-34).
//
// Here: ONE sorted array of (key, asof value, row) triples over the whole build side -- sorted by key, then by asof value (two stable LSD
// radix sorts, sort_kernels.hip) -- answers both steps with one binary search on the composite (key, asof): the entry next to the
// bound is the answer iff it carries the left row's key.  No hash table: the search's first ~10 levels stay in L2 for every row, and a
// 1e7-row build side is 24 levels.  Keys compare by their zero-extended bits, asof values through an order key (signed: sign bit
// flipped; floats: the IEEE total-order fold, Float32 after its exact widening), so one unsigned comparison serves every type.  Rows
// whose key is NULL, whose ON mask is 0 or whose asof value is NaN are never inserted (no comparison with NaN holds).
// Among right rows with EQUAL (key, asof) the reference returns whichever its sort left first -- unspecified (std::sort / a radix sort
// in reverse) -- here: the last inserted for >= / >, the first inserted for <= / <.
#include "chgpu_internal.h"

#include <vector>

namespace
{
constexpr u32 AT = 256;
constexpr u64 A_NO_ROW = ~0ull;

__device__ __forceinline__ u64 asof_load_bits(const void * p, int type, u64 i, bool & nan)
{
    nan = false;
    switch (type)
    {
        case CHGPU_I64: return (u64)((const i64 *)p)[i] ^ 0x8000000000000000ull;
        case CHGPU_U64: return ((const u64 *)p)[i];
        case CHGPU_U32: return ((const u32 *)p)[i];
        case CHGPU_I32: return (u64)(i64)((const i32 *)p)[i] ^ 0x8000000000000000ull;
        case CHGPU_U16: return ((const u16 *)p)[i];
        case CHGPU_I16: return (u64)(i64)((const i16 *)p)[i] ^ 0x8000000000000000ull;
        case CHGPU_U8: return ((const u8 *)p)[i];
        case CHGPU_I8: return (u64)(i64)((const i8 *)p)[i] ^ 0x8000000000000000ull;
        default:
        {
            const double d = type == CHGPU_F64 ? ((const double *)p)[i] : (double)((const float *)p)[i];
            nan = d != d;
            u64 b = (u64)__double_as_longlong(d);
            if ((b << 1) == 0)
                b = 0; // -0.0 == +0.0
            return (b >> 63) ? ~b : b ^ 0x8000000000000000ull;
        }
    }
}
__device__ __forceinline__ u64 asof_load_key(const void * p, int type, u64 i)
{
    switch (type)
    {
        case CHGPU_I64: case CHGPU_U64: return ((const u64 *)p)[i];
        case CHGPU_I32: case CHGPU_U32: return ((const u32 *)p)[i];
        case CHGPU_I16: case CHGPU_U16: return ((const u16 *)p)[i];
        default: return ((const u8 *)p)[i];
    }
}

// one build block -> order keys, (block << 32 | row) ids and the rows to keep
__global__ __launch_bounds__(AT) void k_asof_stage(const void * __restrict__ keys, int key_type, const void * __restrict__ asof, int asof_type, const u8 * __restrict__ null_map,
                                                    const u8 * __restrict__ join_mask, u64 n, u64 block_index, u64 * __restrict__ k_out, u64 * __restrict__ a_out,
                                                    u64 * __restrict__ rowid, u8 * __restrict__ keep)
{
    for (u64 i = (u64)blockIdx.x * AT + threadIdx.x; i < n; i += (u64)gridDim.x * AT)
    {
        bool nan;
        k_out[i] = asof_load_key(keys, key_type, i);
        a_out[i] = asof_load_bits(asof, asof_type, i, nan);
        rowid[i] = (block_index << 32) | i;
        keep[i] = (nan || (null_map && null_map[i]) || (join_mask && !join_mask[i])) ? 0 : 1;
    }
}

// inequality: CHGPU_ASOF_* (ASOFJoinInequality, src/Core/Joins.h:78-85): LESS a.t < b.t, GREATER a.t > b.t, LESS_OR_EQUALS, GREATER_OR_EQUALS
__global__ __launch_bounds__(AT) void k_asof_probe(const u64 * __restrict__ sk, const u64 * __restrict__ sa, const u64 * __restrict__ srow, u64 nb, const void * __restrict__ keys,
                                                    int key_type, const void * __restrict__ asof, int asof_type, const u8 * __restrict__ null_map, u64 n, int inequality,
                                                    u8 * __restrict__ match, u64 * __restrict__ rowid)
{
    const bool want_upper = inequality == CHGPU_ASOF_GREATER_OR_EQUALS || inequality == CHGPU_ASOF_LESS; // first entry >  (K, A); else first entry >= (K, A)
    const bool before = inequality == CHGPU_ASOF_GREATER_OR_EQUALS || inequality == CHGPU_ASOF_GREATER;   // the answer sits just in front of the bound
    for (u64 i = (u64)blockIdx.x * AT + threadIdx.x; i < n; i += (u64)gridDim.x * AT)
    {
        bool nan;
        const u64 K = asof_load_key(keys, key_type, i), A = asof_load_bits(asof, asof_type, i, nan);
        u64 found = A_NO_ROW;
        if (!nan && !(null_map && null_map[i]) && nb)
        {
            u64 lo = 0, hi = nb; // first index whose (key, asof) is > (K, A) [want_upper] or >= (K, A)
            while (lo < hi)
            {
                const u64 mid = (lo + hi) >> 1;
                const u64 mk = sk[mid], ma = sa[mid];
                const bool less = mk < K || (mk == K && (want_upper ? ma <= A : ma < A)); // entry sorts in front of the bound
                if (less)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            const u64 pos = before ? lo - 1 : lo; // (lo == 0 and `before`: wraps to ~0, >= nb)
            if (pos < nb && sk[pos] == K)
                found = srow[pos];
        }
        match[i] = found != A_NO_ROW;
        rowid[i] = found;
    }
}
} // namespace

struct chgpu_asof
{
    chgpu_ctx * ctx = nullptr;
    int key_type = 0, asof_type = 0, kind = 0, inequality = 0;
    struct Block
    {
        chgpu_col * k = nullptr;
        chgpu_col * a = nullptr;
        chgpu_col * row = nullptr;
    };
    std::vector<Block> blocks; // the kept rows of every build block
    u64 n_blocks = 0;
    bool built = false;
    chgpu_col * sk = nullptr;
    chgpu_col * sa = nullptr;
    chgpu_col * srow = nullptr;
    u64 rows = 0;
};

static void asof_drop_blocks(chgpu_asof * a)
{
    for (auto & b : a->blocks)
    {
        chgpu_col_free(b.k);
        chgpu_col_free(b.a);
        chgpu_col_free(b.row);
    }
    a->blocks.clear();
}

extern "C" int chgpu_asof_create(chgpu_ctx * ctx, int key_type, int asof_type, int kind, int inequality, chgpu_asof ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(chgpu_type_is_int(key_type), CHGPU_ERR_NOT_IMPLEMENTED, "ASOF join key type %d: CPU path", key_type);
    CHGPU_REQUIRE(chgpu_type_size(asof_type) != 0, CHGPU_ERR_BAD_ARGUMENTS, "bad ASOF column type %d", asof_type);
    CHGPU_REQUIRE(kind == CHGPU_JOIN_INNER || kind == CHGPU_JOIN_LEFT, CHGPU_ERR_NOT_IMPLEMENTED, "ASOF join kind %d: only INNER and LEFT exist (joinDispatch.h:66-67)", kind);
    CHGPU_REQUIRE(inequality >= CHGPU_ASOF_LESS && inequality <= CHGPU_ASOF_GREATER_OR_EQUALS, CHGPU_ERR_BAD_ARGUMENTS, "bad ASOF inequality %d", inequality);
    chgpu_asof * a = new chgpu_asof();
    a->ctx = ctx;
    a->key_type = key_type;
    a->asof_type = asof_type;
    a->kind = kind;
    a->inequality = inequality;
    chgpu_ctx_retain(ctx);
    *out = a;
    return CHGPU_OK;
}

extern "C" int chgpu_asof_free(chgpu_asof * a)
{
    if (!a)
        return CHGPU_OK;
    ChgpuDeviceGuard _dev_guard(a->ctx);
    asof_drop_blocks(a);
    chgpu_col_free(a->sk);
    chgpu_col_free(a->sa);
    chgpu_col_free(a->srow);
    chgpu_ctx * ctx = a->ctx;
    delete a;
    chgpu_ctx_release(ctx);
    return CHGPU_OK;
}

extern "C" int chgpu_asof_add_block(chgpu_asof * a, const chgpu_col * key_col, const chgpu_col * asof_col, const chgpu_col * null_map, const chgpu_col * join_mask,
                                    uint64_t * block_index)
{
    ChgpuDeviceGuard _dev_guard(a ? a->ctx : nullptr);
    CHGPU_REQUIRE(a && key_col && asof_col, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(!a->built, CHGPU_ERR_LOGICAL, "addBlockToJoin after the first joinBlock (the sorted vectors are immutable, RowRefs.cpp:174-178)");
    CHGPU_REQUIRE(key_col->type == a->key_type && asof_col->type == a->asof_type, CHGPU_ERR_BAD_ARGUMENTS, "column types differ from the join's");
    const u64 n = key_col->rows;
    CHGPU_REQUIRE(asof_col->rows == n && (!null_map || (null_map->type == CHGPU_U8 && null_map->rows == n)) && (!join_mask || (join_mask->type == CHGPU_U8 && join_mask->rows == n)),
                  CHGPU_ERR_SIZES_MISMATCH, "Sizes of columns doesn't match");
    CHGPU_REQUIRE(n < (1ull << 32), CHGPU_ERR_BAD_ARGUMENTS, "Too many rows in right table block for HashJoin: %llu (TOO_MANY_ROWS)", (unsigned long long)n);
    chgpu_ctx * ctx = a->ctx;
    if (block_index)
        *block_index = a->n_blocks;
    const u64 bi = a->n_blocks++;
    if (n == 0)
        return CHGPU_OK;
    chgpu_col * cols[3] = {nullptr, nullptr, nullptr};
    chgpu_col * keep = nullptr;
    int rc = CHGPU_OK;
    for (int c = 0; c < 3 && rc == CHGPU_OK; ++c)
        rc = chgpu_col_new(ctx, CHGPU_U64, n, &cols[c]);
    if (rc == CHGPU_OK)
        rc = chgpu_col_new(ctx, CHGPU_U8, n, &keep);
    if (rc == CHGPU_OK)
    {
        hipLaunchKernelGGL(k_asof_stage, dim3(chgpu_grid_for(ctx, n, AT, 8)), dim3(AT), 0, ctx->stream, (const void *)key_col->data, a->key_type, (const void *)asof_col->data,
                           a->asof_type, null_map ? (const u8 *)null_map->data : nullptr, join_mask ? (const u8 *)join_mask->data : nullptr, n, bi, (u64 *)cols[0]->data,
                           (u64 *)cols[1]->data, (u64 *)cols[2]->data, (u8 *)keep->data);
        ctx->counters[6] += 1;
        if (hipGetLastError() != hipSuccess)
            rc = chgpu_set_error(CHGPU_ERR_DEVICE, "asof stage launch failed");
    }
    chgpu_col * kept[3] = {nullptr, nullptr, nullptr};
    u64 rows = 0;
    if (rc == CHGPU_OK)
    {
        const chgpu_col * in[3] = {cols[0], cols[1], cols[2]};
        rc = chgpu_filter_columns(ctx, 3, in, keep, 0, kept, &rows);
    }
    for (auto * c : cols)
        chgpu_col_free(c);
    chgpu_col_free(keep);
    if (rc != CHGPU_OK)
        return rc;
    a->blocks.push_back({kept[0], kept[1], kept[2]});
    a->rows += rows;
    return CHGPU_OK;
}

static int asof_build(chgpu_asof * a)
{
    if (a->built)
        return CHGPU_OK;
    chgpu_ctx * ctx = a->ctx;
    if (a->rows)
    {
        std::vector<const chgpu_col *> ks, as, rs;
        for (auto & b : a->blocks)
        {
            ks.push_back(b.k);
            as.push_back(b.a);
            rs.push_back(b.row);
        }
        chgpu_col * k = nullptr;
        chgpu_col * av = nullptr;
        chgpu_col * r = nullptr;
        chgpu_col * p1 = nullptr;
        chgpu_col * p2 = nullptr;
        int rc = chgpu_col_concat(ctx, (u32)ks.size(), ks.data(), &k);
        if (rc == CHGPU_OK) rc = chgpu_col_concat(ctx, (u32)as.size(), as.data(), &av);
        if (rc == CHGPU_OK) rc = chgpu_col_concat(ctx, (u32)rs.size(), rs.data(), &r);
        // ORDER BY key, asof: sort by asof, then stably by key with that permutation (chgpu_sort_permutation's perm_in)
        if (rc == CHGPU_OK) rc = chgpu_sort_permutation(ctx, av, nullptr, 0, 1, &p1);
        if (rc == CHGPU_OK) rc = chgpu_sort_permutation(ctx, k, p1, 0, 1, &p2);
        if (rc == CHGPU_OK) rc = chgpu_index(ctx, k, p2, 0, 0, &a->sk);
        if (rc == CHGPU_OK) rc = chgpu_index(ctx, av, p2, 0, 0, &a->sa);
        if (rc == CHGPU_OK) rc = chgpu_index(ctx, r, p2, 0, 0, &a->srow);
        chgpu_col_free(k);
        chgpu_col_free(av);
        chgpu_col_free(r);
        chgpu_col_free(p1);
        chgpu_col_free(p2);
        if (rc != CHGPU_OK)
            return rc;
    }
    asof_drop_blocks(a);
    a->built = true;
    return CHGPU_OK;
}

extern "C" int chgpu_asof_total_rows(chgpu_asof * a, uint64_t * rows)
{
    CHGPU_REQUIRE(a && rows, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    *rows = a->rows;
    return CHGPU_OK;
}

extern "C" int chgpu_asof_probe(chgpu_asof * a, const chgpu_col * key_col, const chgpu_col * asof_col, const chgpu_col * null_map, chgpu_col ** filter_u8,
                                chgpu_col ** right_rowid_u64, uint64_t * n_out)
{
    ChgpuDeviceGuard _dev_guard(a ? a->ctx : nullptr);
    CHGPU_REQUIRE(a && key_col && asof_col && right_rowid_u64 && n_out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(a->kind != CHGPU_JOIN_INNER || filter_u8, CHGPU_ERR_BAD_ARGUMENTS, "an INNER ASOF join produces a filter: filter_u8 must not be NULL");
    CHGPU_REQUIRE(key_col->type == a->key_type && asof_col->type == a->asof_type, CHGPU_ERR_BAD_ARGUMENTS, "column types differ from the join's");
    const u64 n = key_col->rows;
    CHGPU_REQUIRE(asof_col->rows == n && (!null_map || (null_map->type == CHGPU_U8 && null_map->rows == n)), CHGPU_ERR_SIZES_MISMATCH, "Sizes of columns doesn't match");
    CHGPU_TRY(asof_build(a));
    chgpu_ctx * ctx = a->ctx;
    if (filter_u8)
        *filter_u8 = nullptr;
    *right_rowid_u64 = nullptr;
    chgpu_col * match = nullptr;
    chgpu_col * rowid = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U8, n, &match));
    int rc = chgpu_col_new(ctx, CHGPU_U64, n, &rowid);
    if (rc == CHGPU_OK && n)
    {
        hipLaunchKernelGGL(k_asof_probe, dim3(chgpu_grid_for(ctx, n, AT, 8)), dim3(AT), 0, ctx->stream, a->sk ? (const u64 *)a->sk->data : nullptr,
                           a->sa ? (const u64 *)a->sa->data : nullptr, a->srow ? (const u64 *)a->srow->data : nullptr, a->rows, (const void *)key_col->data, a->key_type,
                           (const void *)asof_col->data, a->asof_type, null_map ? (const u8 *)null_map->data : nullptr, n, a->inequality, (u8 *)match->data, (u64 *)rowid->data);
        ctx->counters[6] += 1;
        if (hipGetLastError() != hipSuccess)
            rc = chgpu_set_error(CHGPU_ERR_DEVICE, "asof probe launch failed");
    }
    u64 kept = 0;
    if (rc == CHGPU_OK && a->kind == CHGPU_JOIN_INNER)
    {
        // need_filter: the left rows without a partner go (JoinFeatures.h:31); the right row ids of the kept rows, in order
        chgpu_col * compact = nullptr;
        rc = chgpu_filter(ctx, rowid, match, 0, &compact, &kept);
        if (rc == CHGPU_OK)
        {
            chgpu_col_free(rowid);
            rowid = compact;
        }
    }
    else if (rc == CHGPU_OK)
        kept = n; // LEFT: one row per left row, A_NO_ROW = the default row (addNotFoundRow<add_missing>)
    if (rc != CHGPU_OK)
    {
        chgpu_col_free(match);
        chgpu_col_free(rowid);
        return rc;
    }
    if (filter_u8)
        *filter_u8 = match;
    else
        chgpu_col_free(match);
    *right_rowid_u64 = rowid;
    *n_out = kept;
    ctx->counters[3] += n;
    ctx->counters[4] += kept;
    return CHGPU_OK;
}
