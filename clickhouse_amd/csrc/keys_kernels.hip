// keys_kernels.hip — multi-column fixed-width GROUP BY keys packed into one UInt64 (AggregatedDataVariants keys16/32/64).
//
// Reference loops replaced (file:line in the reference checkout):
//   k_pack_fixed     packFixed<UInt64> / HashMethodKeysFixed   src/Interpreters/AggregationCommon.h:91-158,
//                                                              src/Common/ColumnsHashing/HashMethod.h:228-410,
//                    chooseAggregationMethod keys_bytes <= 8    src/Interpreters/Aggregator.cpp:773-778
//   k_unpack_fixed   insertKeyIntoColumns for fixed keys        src/Interpreters/AggregationMethod.h (AggregationMethodKeysFixed)
// The keys are laid out consecutively, little endian, in the order given (the reference may reorder columns by size for
// its batched packing — an internal layout, not observable in results).
#include "chgpu_internal.h"

static constexpr u32 PK_MAX_COLS = 8;

struct PackCols
{
    u32 n;
    const void * src[PK_MAX_COLS];
    u32 size[PK_MAX_COLS];
    u32 offset[PK_MAX_COLS]; // byte offset inside the packed key
};

__global__ __launch_bounds__(256) void k_pack_fixed(PackCols c, u64 rows, u64 * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < rows; i += (u64)gridDim.x * 256)
    {
        u64 key = 0;
        for (u32 j = 0; j < c.n; ++j)
        {
            u64 v;
            switch (c.size[j])
            {
                case 1: v = ((const u8 *)c.src[j])[i]; break;
                case 2: v = ((const u16 *)c.src[j])[i]; break;
                case 4: v = ((const u32 *)c.src[j])[i]; break;
                default: v = ((const u64 *)c.src[j])[i]; break;
            }
            key |= v << (8 * c.offset[j]);
        }
        out[i] = key;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_unpack_fixed(const u64 * __restrict__ packed, u64 rows, u32 byte_offset, T * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < rows; i += (u64)gridDim.x * 256)
        out[i] = (T)(packed[i] >> (8 * byte_offset));
}

extern "C" int chgpu_pack_fixed_keys(chgpu_ctx * ctx, uint32_t n_cols, const chgpu_col * const * cols, chgpu_col ** packed_u64)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && cols && packed_u64, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(n_cols >= 1 && n_cols <= PK_MAX_COLS, CHGPU_ERR_BAD_ARGUMENTS, "1..%u key columns expected", PK_MAX_COLS);
    PackCols pc;
    pc.n = n_cols;
    u32 off = 0;
    const u64 rows = cols[0] ? cols[0]->rows : 0;
    for (u32 j = 0; j < n_cols; ++j)
    {
        CHGPU_REQUIRE(cols[j], CHGPU_ERR_BAD_ARGUMENTS, "key column %u is NULL", j);
        CHGPU_REQUIRE(cols[j]->rows == rows, CHGPU_ERR_SIZES_MISMATCH, "key columns have different sizes");
        CHGPU_REQUIRE(!chgpu_type_is_float(cols[j]->type), CHGPU_ERR_NOT_IMPLEMENTED, "Float64 in a packed key: CPU path");
        pc.src[j] = cols[j]->data;
        pc.size[j] = (u32)chgpu_type_size(cols[j]->type);
        pc.offset[j] = off;
        off += pc.size[j];
    }
    // keys_bytes <= 8 -> keys64 (Aggregator.cpp:777-778); wider tuples are keys128/256 on the CPU path
    CHGPU_REQUIRE(off <= 8, CHGPU_ERR_NOT_IMPLEMENTED, "packed key of %u bytes exceeds keys64: CPU path (keys128/keys256)", off);
    chgpu_col * out = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, rows, &out));
    if (rows)
    {
        hipLaunchKernelGGL(k_pack_fixed, dim3(chgpu_grid_for(ctx, rows, 256, 8)), dim3(256), 0, ctx->stream, pc, rows, (u64 *)out->data);
        ctx->counters[6] += 1;
    }
    *packed_u64 = out;
    return CHGPU_OK;
}

extern "C" int chgpu_unpack_fixed_key(chgpu_ctx * ctx, const chgpu_col * packed_u64, uint32_t byte_offset, int type, chgpu_col ** out_col)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && packed_u64 && out_col, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(chgpu_type_size(packed_u64->type) == 8, CHGPU_ERR_BAD_ARGUMENTS, "packed keys must be a 64-bit column");
    const size_t es = chgpu_type_size(type);
    CHGPU_REQUIRE(es && !chgpu_type_is_float(type), CHGPU_ERR_BAD_ARGUMENTS, "bad key type %d", type);
    CHGPU_REQUIRE(byte_offset + es <= 8, CHGPU_ERR_BAD_ARGUMENTS, "key slice [%u,+%zu) outside the 8-byte packed key", byte_offset, es);
    chgpu_col * out = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, type, packed_u64->rows, &out));
    const u64 rows = packed_u64->rows;
    if (rows)
    {
        const u32 grid = chgpu_grid_for(ctx, rows, 256, 8);
        if (es == 8)
            hipLaunchKernelGGL(k_unpack_fixed<u64>, dim3(grid), dim3(256), 0, ctx->stream, (const u64 *)packed_u64->data, rows, byte_offset, (u64 *)out->data);
        else if (es == 4)
            hipLaunchKernelGGL(k_unpack_fixed<u32>, dim3(grid), dim3(256), 0, ctx->stream, (const u64 *)packed_u64->data, rows, byte_offset, (u32 *)out->data);
        else if (es == 2)
            hipLaunchKernelGGL(k_unpack_fixed<u16>, dim3(grid), dim3(256), 0, ctx->stream, (const u64 *)packed_u64->data, rows, byte_offset, (u16 *)out->data);
        else
            hipLaunchKernelGGL(k_unpack_fixed<u8>, dim3(grid), dim3(256), 0, ctx->stream, (const u64 *)packed_u64->data, rows, byte_offset, (u8 *)out->data);
        ctx->counters[6] += 1;
    }
    *out_col = out;
    return CHGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// SURVEY §8(f) rank 2 — LowCardinality keys.  A ColumnLowCardinality is a dictionary plus an index column
// (src/Columns/ColumnLowCardinality.h:27-69); every Block may bring its own dictionary.  The reference's
// low_cardinality_key* aggregation methods (AggregatedDataVariants.h:119-127) look each DICTIONARY entry up once per block
// and then walk the rows through a per-position cache (HashMethodSingleLowCardinalityColumn, ColumnsHashing.h:82-260:
// mapped_cache[row]).  Here the host resolves the block's dictionary against the query-wide one (a few thousand entries,
// low_cardinality_max_dictionary_size = 8192) and the rows are translated on the device:
//   k_lc_remap<I>   out[i] = remap[indexes[i]]   — the mapped_cache walk; the table sits in LDS when it fits
// Streaming geometry: four rows per lane and step (a 4- / 8- / 16-byte load), table lookups from LDS, 16-byte stores on consecutive addresses.
// Algorithmic bytes: sizeof(I) + 4 per row.  Out-of-range indexes (a caller bug) read entry 0 instead of faulting.
// ---------------------------------------------------------------------------------------------
static constexpr u32 LC_LDS_ENTRIES = 32768; // 128 KiB of UInt32 ids

// VEC: the index column's first row is 16-byte aligned (vector body + scalar tail); otherwise one row per lane throughout
template <typename I, bool IN_LDS, bool VEC>
__global__ __launch_bounds__(256) void k_lc_remap(const I * __restrict__ idx, u64 n, const u32 * __restrict__ remap, u32 dict_size, u32 * __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) u32 lc_tab[];
    if constexpr (IN_LDS)
    {
        for (u32 k = threadIdx.x; k < dict_size; k += 256)
            lc_tab[k] = remap[k];
        __syncthreads();
    }
    auto look = [&](u64 j) -> u32 {
        const u32 k = j < dict_size ? (u32)j : 0u;
        if constexpr (IN_LDS)
            return lc_tab[k];
        else
            return remap[k];
    };
    const u64 stride = (u64)gridDim.x * 256;
    u64 done = 0;
    if constexpr (VEC && sizeof(I) <= 4)
    {
        // FOUR rows per lane and step whatever the index width (a 4- / 8- / 16-byte load), so that every store instruction writes 16 bytes
        // per lane to CONSECUTIVE addresses: 1 KiB per wave and instruction.  (16 input bytes per lane made a UInt8 lane own 16 rows and
        // each of its four 16-byte stores hit a quarter of 64 different lines: 1.86 ms per 1e9 UInt8 rows against ~1.0 here.)  Four steps
        // are in flight per lane.
        typedef I vin __attribute__((ext_vector_type(4)));
        typedef u32 v4u __attribute__((ext_vector_type(4)));
        const u64 nvec = n / 4;
        constexpr int U = 4;
        for (u64 v0 = (u64)blockIdx.x * 256 + threadIdx.x; v0 < nvec; v0 += stride * U)
        {
            vin x[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                const u64 v = v0 + (u64)u * stride;
                x[u] = __builtin_nontemporal_load((const vin *)idx + (v < nvec ? v : v0));
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                const u64 v = v0 + (u64)u * stride;
                v4u o;
                o.x = look((u64)x[u][0]), o.y = look((u64)x[u][1]), o.z = look((u64)x[u][2]), o.w = look((u64)x[u][3]);
                if (v < nvec)
                    *((v4u *)out + v) = o; // result columns are 64-byte aligned (nontemporal stores here: 0.36 -> 1.14 ms per 2e8 rows)
            }
        }
        done = nvec * 4;
    }
    for (u64 i = done + (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        out[i] = look((u64)idx[i]);
}

template <typename I>
static int lc_launch(chgpu_ctx * ctx, const chgpu_col * indexes, const u32 * rm, u32 dict, u32 * o)
{
    const u64 n = indexes->rows;
    const bool in_lds = dict <= LC_LDS_ENTRIES;
    const size_t lds = in_lds ? (size_t)((dict + 3) & ~3u) * 4 : 0;
    const bool vec = ((uintptr_t)indexes->data & 15) == 0 && sizeof(I) <= 4;
    const u32 per_cu = lds > 64 * 1024 ? 1 : lds > 32 * 1024 ? 2 : 4; // LDS bounds the residency
    const u64 items = vec ? (n + 15) / 16 : n; // four steps of four rows per lane
    const u32 grid = chgpu_grid_for(ctx, items, 256, per_cu);
    const I * ip = (const I *)indexes->data;
#define LC_GO(L, V)                                                                                                                   \
    do                                                                                                                                \
    {                                                                                                                                 \
        if (lds > 64 * 1024)                                                                                                          \
            CHGPU_HIP(hipFuncSetAttribute((const void *)k_lc_remap<I, L, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_lc_remap<I, L, V>), dim3(grid), dim3(256), lds, ctx->stream, ip, n, rm, dict, o);                      \
    } while (0)
    if (in_lds && vec) LC_GO(true, true);
    else if (in_lds) LC_GO(true, false);
    else if (vec) LC_GO(false, true);
    else LC_GO(false, false);
#undef LC_GO
    ctx->counters[6] += 1;
    CHGPU_HIP(hipGetLastError());
    return CHGPU_OK;
}

extern "C" int chgpu_lc_remap(chgpu_ctx * ctx, const chgpu_col * indexes, const chgpu_col * remap_u32, chgpu_col ** out_u32)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && indexes && remap_u32 && out_u32, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(remap_u32->type == CHGPU_U32, CHGPU_ERR_BAD_ARGUMENTS, "the remap table must be UInt32");
    CHGPU_REQUIRE(indexes->type == CHGPU_U8 || indexes->type == CHGPU_U16 || indexes->type == CHGPU_U32 || indexes->type == CHGPU_U64,
                  CHGPU_ERR_BAD_ARGUMENTS, "LowCardinality indexes are UInt8 / UInt16 / UInt32 / UInt64 (ColumnLowCardinality.h Index)");
    CHGPU_REQUIRE(remap_u32->rows > 0 || indexes->rows == 0, CHGPU_ERR_BAD_ARGUMENTS, "empty dictionary");
    CHGPU_REQUIRE(remap_u32->rows < (1ull << 32), CHGPU_ERR_BAD_ARGUMENTS, "dictionary of 2^32 entries");
    chgpu_col * res = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U32, indexes->rows, &res));
    int rc = CHGPU_OK;
    if (indexes->rows)
    {
        const u32 * rm = (const u32 *)remap_u32->data;
        const u32 dict = (u32)remap_u32->rows;
        u32 * o = (u32 *)res->data;
        switch (indexes->type)
        {
            case CHGPU_U8: rc = lc_launch<u8>(ctx, indexes, rm, dict, o); break;
            case CHGPU_U16: rc = lc_launch<u16>(ctx, indexes, rm, dict, o); break;
            case CHGPU_U32: rc = lc_launch<u32>(ctx, indexes, rm, dict, o); break;
            default: rc = lc_launch<u64>(ctx, indexes, rm, dict, o); break;
        }
    }
    if (rc != CHGPU_OK)
    {
        chgpu_col_free(res);
        return rc;
    }
    *out_u32 = res;
    return CHGPU_OK;
}


// ---------------------------------------------------------------------------------------------
// FixedString(N) keys (ColumnFixedString: rows x N bytes, src/Columns/ColumnFixedString.h; AggregatedDataVariants::key_fixed_string,
// AggregatedDataVariants.h:65-66,91-92 / HashJoin's key_fixed_string).  A FixedString value is N raw bytes, padding zeros included, so
// it IS a fixed-width key: its 8-byte words (little endian, zero padded past N) go where the other fixed keys go -- one UInt64 key for
// N <= 8 (key64), keys128 / keys256 through the key dictionary for N <= 32.  Two values are equal iff their words are.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fixed_string_word(const u8 * __restrict__ chars, u64 rows, u32 n, u32 word, u64 * __restrict__ out)
{
    const u32 lo = word * 8, cnt = n - lo < 8 ? n - lo : 8;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < rows; i += (u64)gridDim.x * 256)
    {
        const u8 * p = chars + i * n + lo;
        u64 v = 0;
        for (u32 b = 0; b < cnt; ++b)
            v |= (u64)p[b] << (8 * b);
        out[i] = v;
    }
}

struct FsWords
{
    const u64 * w[4];
};
__global__ __launch_bounds__(256) void k_fixed_string_from_words(FsWords words, u64 rows, u32 n, u8 * __restrict__ chars)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < rows; i += (u64)gridDim.x * 256)
        for (u32 b = 0; b < n; ++b)
            chars[i * n + b] = (u8)(words.w[b >> 3][i] >> (8 * (b & 7)));
}

extern "C" int chgpu_fixed_string_word(chgpu_ctx * ctx, const chgpu_col * chars_u8, uint32_t n, uint32_t word_index, chgpu_col ** out_u64)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && chars_u8 && out_u64, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(chars_u8->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS, "FixedString chars are a UInt8 column");
    CHGPU_REQUIRE(n >= 1 && n <= 32, CHGPU_ERR_NOT_IMPLEMENTED, "FixedString(%u) key: beyond keys256, CPU path", n);
    CHGPU_REQUIRE(chars_u8->rows % n == 0, CHGPU_ERR_SIZES_MISMATCH, "FixedString chars (%llu bytes) are not a multiple of N = %u", (unsigned long long)chars_u8->rows, n);
    CHGPU_REQUIRE(word_index * 8 < n, CHGPU_ERR_BAD_ARGUMENTS, "word %u lies beyond FixedString(%u)", word_index, n);
    const u64 rows = chars_u8->rows / n;
    chgpu_col * out = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, rows, &out));
    if (rows)
    {
        hipLaunchKernelGGL(k_fixed_string_word, dim3(chgpu_grid_for(ctx, rows, 256, 8)), dim3(256), 0, ctx->stream, (const u8 *)chars_u8->data, rows, n, word_index, (u64 *)out->data);
        ctx->counters[6] += 1;
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess)
        {
            chgpu_col_free(out);
            return chgpu_set_error(CHGPU_ERR_DEVICE, "fixed_string_word: %s", hipGetErrorString(e));
        }
    }
    *out_u64 = out;
    return CHGPU_OK;
}

extern "C" int chgpu_fixed_string_from_words(chgpu_ctx * ctx, uint32_t n_words, const chgpu_col * const * words_u64, uint32_t n, chgpu_col ** chars_u8)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && words_u64 && chars_u8, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(n >= 1 && n <= 32 && n_words == (n + 7) / 8, CHGPU_ERR_BAD_ARGUMENTS, "FixedString(%u) takes %u words, %u given", n, (n + 7) / 8, n_words);
    FsWords fw{};
    u64 rows = 0;
    for (u32 w = 0; w < n_words; ++w)
    {
        CHGPU_REQUIRE(words_u64[w] && chgpu_type_size(words_u64[w]->type) == 8, CHGPU_ERR_BAD_ARGUMENTS, "word column %u must be 8 bytes wide", w);
        CHGPU_REQUIRE(w == 0 || words_u64[w]->rows == rows, CHGPU_ERR_SIZES_MISMATCH, "word columns differ in length");
        rows = words_u64[w]->rows;
        fw.w[w] = (const u64 *)words_u64[w]->data;
    }
    chgpu_col * out = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U8, rows * n, &out));
    if (rows)
    {
        hipLaunchKernelGGL(k_fixed_string_from_words, dim3(chgpu_grid_for(ctx, rows, 256, 8)), dim3(256), 0, ctx->stream, fw, rows, n, (u8 *)out->data);
        ctx->counters[6] += 1;
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess)
        {
            chgpu_col_free(out);
            return chgpu_set_error(CHGPU_ERR_DEVICE, "fixed_string_from_words: %s", hipGetErrorString(e));
        }
    }
    *chars_u8 = out;
    return CHGPU_OK;
}
