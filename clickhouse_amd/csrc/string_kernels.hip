// string_kernels.hip — SURVEY §8(f) rank 2: String keys.  A ColumnString (src/Columns/ColumnString.h:40-49: `chars` with a
// terminating zero after every value, `offsets[i]` = end of value i including that zero) is dictionary-encoded on the device:
// every row gets the dense id of its value, ids numbered by first appearance — what ColumnUnique::uniqueInsertRangeFrom builds
// when a String column is turned into a LowCardinality one (src/Columns/ColumnUnique.h:520-620), and what the reference's
// key_string / StringHashMap aggregation (AggregatedDataVariants.h:96, Common/HashTable/StringHashTable.h) achieves per row with
// a CPU hash table keyed by the bytes.  The ids then take the ordinary UInt32 GROUP BY / join path (through
// LowCardinalityDictionary when several stripes or Blocks must agree on ids).
//
//   k_str_hash      64-bit hash of every value (8 bytes per step, unaligned loads inside the column's padding)
//   k_str_insert    open-addressing table of hash tags; every row lowers `first_row` of its tag's cell (atomicMin): the
//                   representative of a value is its FIRST row, independent of scheduling
//   k_str_resolve   every row compares its bytes with its representative's (length + content): equal -> it belongs to that value;
//                   different bytes under one 64-bit tag -> the collision flag (the call answers NOT_IMPLEMENTED: exactness is
//                   never traded; the reference's `hashed` method accepts 128-bit collisions, this path accepts none)
//   scan of the "I am a first row" flags -> dense ids in order of first appearance; k_str_ids gathers them per row
// The hash is internal (placement only).  Algorithmic bytes: chars once + 8 B offsets + 4 B id per row; the table traffic is
// random 16-byte cells, two touches per row.
#include "chgpu_internal.h"

__device__ __forceinline__ u64 str_load8(const u8 * p)
{
    u64 v;
    __builtin_memcpy(&v, p, 8); // unaligned: global memory allows it; the column's 64-byte pad covers the over-read
    return v;
}

__device__ __forceinline__ u64 str_hash_bytes(const u8 * p, u64 len)
{
    u64 h = 0x9E3779B97F4A7C15ull ^ (len * 0xff51afd7ed558ccdull);
    u64 i = 0;
    for (; i + 8 <= len; i += 8)
    {
        h = (h ^ str_load8(p + i)) * 0xc4ceb9fe1a85ec53ull;
        h ^= h >> 29;
    }
    if (i < len)
    {
        const u64 tail = str_load8(p + i) & (~0ull >> (8 * (8 - (len - i))));
        h = (h ^ tail) * 0xc4ceb9fe1a85ec53ull;
        h ^= h >> 29;
    }
    h = dev_intHash64(h);
    return h | 1ull; // 0 marks an empty cell
}

__device__ __forceinline__ bool str_equal(const u8 * a, const u8 * b, u64 len)
{
    u64 i = 0;
    for (; i + 8 <= len; i += 8)
        if (str_load8(a + i) != str_load8(b + i))
            return false;
    if (i < len)
    {
        const u64 m = ~0ull >> (8 * (8 - (len - i)));
        return ((str_load8(a + i) ^ str_load8(b + i)) & m) == 0;
    }
    return true;
}

// ColumnString invariant (ColumnString.h:40-52): offsets strictly increase (every value has at least its terminating zero) and end inside
// chars.  The kernels below compute `offsets[i] - begin - 1` bytes per value and read that many: an offset column that breaks the
// invariant (corrupted or hostile input) would turn into a 2^64-byte walk -- a hang or a fault, not an error code.
__global__ __launch_bounds__(256) void k_str_check_offsets(const u64 * __restrict__ offsets, u64 n, u64 chars_size, u32 * __restrict__ bad)
{
    bool b = false;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        const u64 begin = i ? offsets[i - 1] : 0, end = offsets[i];
        b = b || !(begin < end && end <= chars_size);
    }
    if (b)
        *bad = 1;
}

static int str_validate_offsets(chgpu_ctx * ctx, const chgpu_col * offsets_u64, const chgpu_col * chars_u8)
{
    const u64 n = offsets_u64->rows;
    if (!n)
        return CHGPU_OK;
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, 256, &scratch));
    CHGPU_HIP(hipMemsetAsync(scratch, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_str_check_offsets, dim3(chgpu_grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, (const u64 *)offsets_u64->data, n, (u64)chars_u8->rows, (u32 *)scratch);
    ctx->counters[6] += 1;
    u32 bad = 0;
    CHGPU_TRY(chgpu_read_back(ctx, scratch, &bad, 4));
    CHGPU_REQUIRE(!bad, CHGPU_ERR_BAD_ARGUMENTS, "ColumnString offsets must strictly increase and stay inside chars (%llu bytes)", (unsigned long long)chars_u8->rows);
    return CHGPU_OK;
}

__global__ __launch_bounds__(256) void k_str_hash(const u64 * __restrict__ offsets, const u8 * __restrict__ chars, u64 n, u64 * __restrict__ hash)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        const u64 begin = i ? offsets[i - 1] : 0;
        const u64 len = offsets[i] - begin - 1; // without the terminating zero (ColumnString.h:48-52)
        hash[i] = str_hash_bytes(chars + begin, len);
    }
}

__global__ __launch_bounds__(256) void k_str_insert(const u64 * __restrict__ hash, u64 n, u64 * __restrict__ tags, unsigned long long * __restrict__ first_row, u64 mask,
                                                    u32 * __restrict__ fail)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        const u64 h = hash[i];
        u64 s = (h >> 1) & mask;
        bool placed = false;
        for (u64 probe = 0; probe <= mask; ++probe) // bounded: the table has >= 2 cells per row
        {
            u64 t = tags[s];
            if (t == 0)
            {
                t = atomicCAS((unsigned long long *)&tags[s], 0ull, (unsigned long long)h);
                if (t == 0)
                    t = h;
            }
            if (t == h)
            {
                // first_row only ever decreases, so a (possibly stale) value <= i proves the atomic would change nothing; without
                // this test a low-cardinality column sends every row's atomic to a few thousand addresses (same-address atomics
                // serialise: 1e8 rows over 2500 values took 414 ms, all of it here)
                if (__builtin_nontemporal_load(&first_row[s]) > (unsigned long long)i)
                    atomicMin(&first_row[s], (unsigned long long)i);
                placed = true;
                break;
            }
            s = (s + 1) & mask;
        }
        if (!placed)
            *fail = 1;
    }
}

__global__ __launch_bounds__(256) void k_str_resolve(const u64 * __restrict__ offsets, const u8 * __restrict__ chars, const u64 * __restrict__ hash, u64 n,
                                                     const u64 * __restrict__ tags, const unsigned long long * __restrict__ first_row, u64 mask,
                                                     u64 * __restrict__ rep_row, u32 * __restrict__ is_first, u32 * __restrict__ fail)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        const u64 h = hash[i];
        u64 s = (h >> 1) & mask;
        u64 r = ~0ull;
        for (u64 probe = 0; probe <= mask; ++probe)
        {
            const u64 t = tags[s];
            if (t == h)
            {
                r = first_row[s];
                break;
            }
            if (t == 0)
                break;
            s = (s + 1) & mask;
        }
        if (r >= n)
        {
            *fail = 1;
            r = i;
        }
        if (r != i)
        {
            const u64 b0 = i ? offsets[i - 1] : 0, l0 = offsets[i] - b0 - 1;
            const u64 b1 = r ? offsets[r - 1] : 0, l1 = offsets[r] - b1 - 1;
            if (l0 != l1 || !str_equal(chars + b0, chars + b1, l0))
                *fail = 2; // two different values under one 64-bit tag
        }
        rep_row[i] = r;
        is_first[i] = r == i ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void k_str_ids(const u64 * __restrict__ rep_row, const u32 * __restrict__ is_first, const u64 * __restrict__ prefix, u64 n,
                                                 u32 * __restrict__ ids, u64 * __restrict__ dict_rows)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        ids[i] = (u32)prefix[rep_row[i]]; // exclusive prefix of the first-row flags = number of values that appeared earlier
        if (is_first[i])
            dict_rows[prefix[i]] = i;
    }
}

extern "C" int chgpu_string_dictionary_encode(chgpu_ctx * ctx, const chgpu_col * offsets_u64, const chgpu_col * chars_u8, chgpu_col ** ids_u32,
                                              chgpu_col ** first_rows_u64, uint64_t * n_distinct)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && offsets_u64 && chars_u8 && ids_u32 && first_rows_u64 && n_distinct, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(offsets_u64->type == CHGPU_U64 && chars_u8->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS, "ColumnString = UInt64 offsets + UInt8 chars");
    const u64 n = offsets_u64->rows;
    CHGPU_REQUIRE(n < (1ull << 32), CHGPU_ERR_NOT_IMPLEMENTED, "more than 2^32 rows per call");
    chgpu_col * ids = nullptr, * dict = nullptr;
    *n_distinct = 0;
    if (n == 0)
    {
        CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U32, 0, &ids));
        CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, 0, &dict));
        *ids_u32 = ids, *first_rows_u64 = dict;
        return CHGPU_OK;
    }
    // the last offset must equal the size of chars (ColumnString invariant); checked with one read-back
    u64 last = 0;
    CHGPU_TRY(chgpu_read_back(ctx, (const u64 *)offsets_u64->data + (n - 1), &last, sizeof(last)));
    CHGPU_REQUIRE(last == chars_u8->rows, CHGPU_ERR_SIZES_MISMATCH, "offsets.back() (%llu) != chars.size() (%llu)", (unsigned long long)last,
                  (unsigned long long)chars_u8->rows);
    CHGPU_TRY(str_validate_offsets(ctx, offsets_u64, chars_u8));
    u64 cap = 1024;
    while (cap < 2 * n)
        cap <<= 1;
    // temporaries: hash u64[n] | rep_row u64[n] | is_first u32[n] | prefix u64[n] | tags u64[cap] | first_row u64[cap] | fail u32 | scan tmp
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_hash = al(n * 8), b_rep = al(n * 8), b_first = al(n * 4), b_prefix = al(n * 8), b_tags = al(cap * 8), b_rows = al(cap * 8), b_flag = 256;
    const size_t b_tmp = chgpu_scan_tmp_bytes(n);
    void * mem = nullptr;
    size_t mem_class = 0;
    CHGPU_TRY(chgpu_pool_alloc(ctx, b_hash + b_rep + b_first + b_prefix + b_tags + b_rows + b_flag + 256 + b_tmp, &mem, &mem_class));
    char * p = (char *)mem;
    u64 * hash = (u64 *)p; p += b_hash;
    u64 * rep = (u64 *)p; p += b_rep;
    u32 * isf = (u32 *)p; p += b_first;
    u64 * prefix = (u64 *)p; p += b_prefix;
    u64 * tags = (u64 *)p; p += b_tags;
    unsigned long long * rows = (unsigned long long *)p; p += b_rows;
    u32 * fail = (u32 *)p; p += b_flag;
    u64 * total_dev = (u64 *)p; p += 256;
    void * tmp = p;
    int rc = CHGPU_OK;
    auto done = [&](int code) {
        chgpu_pool_free(ctx, mem, mem_class);
        if (code != CHGPU_OK)
        {
            if (ids)
                chgpu_col_free(ids);
            if (dict)
                chgpu_col_free(dict);
        }
        return code;
    };
    if (hipMemsetAsync(tags, 0, b_tags, ctx->stream) != hipSuccess || hipMemsetAsync(rows, 0xFF, b_rows, ctx->stream) != hipSuccess ||
        hipMemsetAsync(fail, 0, b_flag, ctx->stream) != hipSuccess)
        return done(chgpu_set_error(CHGPU_ERR_DEVICE, "memset failed"));
    const u32 grid = chgpu_grid_for(ctx, n, 256, 8);
    const u64 * offs = (const u64 *)offsets_u64->data;
    const u8 * chars = (const u8 *)chars_u8->data;
    hipLaunchKernelGGL(k_str_hash, dim3(grid), dim3(256), 0, ctx->stream, offs, chars, n, hash);
    hipLaunchKernelGGL(k_str_insert, dim3(grid), dim3(256), 0, ctx->stream, (const u64 *)hash, n, tags, rows, cap - 1, fail);
    hipLaunchKernelGGL(k_str_resolve, dim3(grid), dim3(256), 0, ctx->stream, offs, chars, (const u64 *)hash, n, (const u64 *)tags, (const unsigned long long *)rows,
                       cap - 1, rep, isf, fail);
    ctx->counters[6] += 3;
    rc = chgpu_scan_exclusive_u32_u64(ctx, isf, prefix, n, total_dev, tmp, b_tmp);
    if (rc != CHGPU_OK)
        return done(rc);
    struct { u64 total; } hb;
    rc = chgpu_read_back(ctx, total_dev, &hb.total, sizeof(u64));
    if (rc != CHGPU_OK)
        return done(rc);
    u32 failed = 0;
    rc = chgpu_read_back(ctx, fail, &failed, sizeof(failed));
    if (rc != CHGPU_OK)
        return done(rc);
    if (failed == 2)
        return done(chgpu_set_error(CHGPU_ERR_NOT_IMPLEMENTED, "two different strings share a 64-bit hash tag in this block: CPU path"));
    if (failed)
        return done(chgpu_set_error(CHGPU_ERR_LOGICAL, "string table probe did not terminate"));
    rc = chgpu_col_new(ctx, CHGPU_U32, n, &ids);
    if (rc == CHGPU_OK)
        rc = chgpu_col_new(ctx, CHGPU_U64, hb.total, &dict);
    if (rc != CHGPU_OK)
        return done(rc);
    hipLaunchKernelGGL(k_str_ids, dim3(grid), dim3(256), 0, ctx->stream, (const u64 *)rep, (const u32 *)isf, (const u64 *)prefix, n, (u32 *)ids->data, (u64 *)dict->data);
    ctx->counters[6] += 1;
    if (hipGetLastError() != hipSuccess)
        return done(chgpu_set_error(CHGPU_ERR_DEVICE, "string dictionary kernels failed to launch"));
    *ids_u32 = ids;
    *first_rows_u64 = dict;
    *n_distinct = hb.total;
    return done(CHGPU_OK);
}

// ---------------------------------------------------------------------------------------------
// ColumnString::filter (src/Columns/ColumnString.cpp:270-290 -> filterArraysImpl<UInt8>, src/Columns/ColumnsCommon.cpp:191-286):
// the kept values' bytes are moved together and the offsets rebuilt.  The reference walks the mask 64 rows at a time and memcpy's
// runs of kept values; here two scans (kept rows, kept bytes) give every surviving value its new row and its new byte position,
// and one pass copies the values (8 bytes per step per lane, unaligned; the terminating zero travels with the value).
//   k_str_filter_sizes   flag[i] = mask[i] != 0, bytes[i] = flag ? size of value i incl. its zero : 0
//   k_str_filter_move    out_offsets[row'] = pos' + size;  out_chars[pos' ..] = chars[begin ..]
// Algorithmic bytes: 8 (offset) + 1 (mask) per row + kept bytes read and written + 8 per kept row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_str_filter_sizes(const u64 * __restrict__ offsets, const u8 * __restrict__ mask, u64 n, u32 * __restrict__ flag,
                                                          u32 * __restrict__ bytes, u32 * __restrict__ too_long)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        const u32 f = mask[i] != 0;
        const u64 sz = offsets[i] - (i ? offsets[i - 1] : 0);
        if (f && sz >= (1ull << 32))
            *too_long = 1;
        flag[i] = f;
        bytes[i] = f ? (u32)sz : 0u;
    }
}

__global__ __launch_bounds__(256) void k_str_filter_move(const u64 * __restrict__ offsets, const u8 * __restrict__ chars, const u32 * __restrict__ flag,
                                                         const u64 * __restrict__ row_pos, const u64 * __restrict__ byte_pos, u64 n,
                                                         u64 * __restrict__ out_offsets, u8 * __restrict__ out_chars)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        if (!flag[i])
            continue;
        const u64 begin = i ? offsets[i - 1] : 0;
        const u64 sz = offsets[i] - begin;
        const u64 dst = byte_pos[i];
        out_offsets[row_pos[i]] = dst + sz;
        u64 k = 0;
        for (; k + 8 <= sz; k += 8)
        {
            const u64 v = str_load8(chars + begin + k);
            __builtin_memcpy(out_chars + dst + k, &v, 8);
        }
        for (; k < sz; ++k)
            out_chars[dst + k] = chars[begin + k];
    }
}

extern "C" int chgpu_string_filter(chgpu_ctx * ctx, const chgpu_col * offsets_u64, const chgpu_col * chars_u8, const chgpu_col * filter_u8,
                                   chgpu_col ** out_offsets_u64, chgpu_col ** out_chars_u8, uint64_t * rows_out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && offsets_u64 && chars_u8 && filter_u8 && out_offsets_u64 && out_chars_u8 && rows_out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(offsets_u64->type == CHGPU_U64 && chars_u8->type == CHGPU_U8 && filter_u8->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS,
                  "ColumnString = UInt64 offsets + UInt8 chars; the filter is a UInt8 column");
    const u64 n = offsets_u64->rows;
    CHGPU_REQUIRE(filter_u8->rows == n, CHGPU_ERR_SIZES_MISMATCH, "Size of filter (%llu) doesn't match size of column (%llu)",
                  (unsigned long long)filter_u8->rows, (unsigned long long)n); // ColumnsCommon.cpp:199-200
    chgpu_col * oo = nullptr, * oc = nullptr;
    u64 kept_rows = 0, kept_bytes = 0;
    CHGPU_TRY(str_validate_offsets(ctx, offsets_u64, chars_u8));
    if (n)
    {
        auto al = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t b_flag = al(n * 4), b_bytes = al(n * 4), b_rpos = al(n * 8), b_bpos = al(n * 8), b_tmp = chgpu_scan_tmp_bytes(n);
        void * mem = nullptr;
        size_t mem_class = 0;
        CHGPU_TRY(chgpu_pool_alloc(ctx, b_flag + b_bytes + b_rpos + b_bpos + 512 + b_tmp, &mem, &mem_class));
        char * p = (char *)mem;
        u32 * flag = (u32 *)p; p += b_flag;
        u32 * bytes = (u32 *)p; p += b_bytes;
        u64 * rpos = (u64 *)p; p += b_rpos;
        u64 * bpos = (u64 *)p; p += b_bpos;
        u64 * totals = (u64 *)p; p += 256; // [0] rows, [1] bytes
        u32 * too_long = (u32 *)p; p += 256;
        void * tmp = p;
        auto done = [&](int code) {
            chgpu_pool_free(ctx, mem, mem_class);
            if (code != CHGPU_OK)
            {
                if (oo)
                    chgpu_col_free(oo);
                if (oc)
                    chgpu_col_free(oc);
            }
            return code;
        };
        if (hipMemsetAsync(too_long, 0, 256, ctx->stream) != hipSuccess)
            return done(chgpu_set_error(CHGPU_ERR_DEVICE, "memset failed"));
        const u32 grid = chgpu_grid_for(ctx, n, 256, 8);
        hipLaunchKernelGGL(k_str_filter_sizes, dim3(grid), dim3(256), 0, ctx->stream, (const u64 *)offsets_u64->data, (const u8 *)filter_u8->data, n, flag, bytes, too_long);
        int rc = chgpu_scan_exclusive_u32_u64(ctx, flag, rpos, n, totals, tmp, b_tmp);
        if (rc == CHGPU_OK)
            rc = chgpu_scan_exclusive_u32_u64(ctx, bytes, bpos, n, totals + 1, tmp, b_tmp);
        u64 host_totals[2] = {0, 0};
        u32 host_long = 0;
        if (rc == CHGPU_OK)
            rc = chgpu_read_back(ctx, totals, host_totals, sizeof(host_totals));
        if (rc == CHGPU_OK)
            rc = chgpu_read_back(ctx, too_long, &host_long, sizeof(host_long));
        if (rc == CHGPU_OK && host_long)
            rc = chgpu_set_error(CHGPU_ERR_NOT_IMPLEMENTED, "a value of 4 GiB or more");
        if (rc != CHGPU_OK)
            return done(rc);
        kept_rows = host_totals[0], kept_bytes = host_totals[1];
        rc = chgpu_col_new(ctx, CHGPU_U64, kept_rows, &oo);
        if (rc == CHGPU_OK)
            rc = chgpu_col_new(ctx, CHGPU_U8, kept_bytes, &oc);
        if (rc != CHGPU_OK)
            return done(rc);
        if (kept_rows)
            hipLaunchKernelGGL(k_str_filter_move, dim3(grid), dim3(256), 0, ctx->stream, (const u64 *)offsets_u64->data, (const u8 *)chars_u8->data, (const u32 *)flag,
                               (const u64 *)rpos, (const u64 *)bpos, n, (u64 *)oo->data, (u8 *)oc->data);
        ctx->counters[6] += 2;
        ctx->counters[0] += kept_rows;
        if (hipGetLastError() != hipSuccess)
            return done(chgpu_set_error(CHGPU_ERR_DEVICE, "string filter kernels failed to launch"));
        done(CHGPU_OK);
    }
    else
    {
        CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, 0, &oo));
        int rc = chgpu_col_new(ctx, CHGPU_U8, 0, &oc);
        if (rc != CHGPU_OK)
        {
            chgpu_col_free(oo);
            return rc;
        }
    }
    *out_offsets_u64 = oo;
    *out_chars_u8 = oc;
    *rows_out = kept_rows;
    return CHGPU_OK;
}
