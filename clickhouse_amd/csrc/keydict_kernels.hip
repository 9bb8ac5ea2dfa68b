// keydict_kernels.hip — multi-column fixed-width keys wider than 8 bytes: the keys128 / keys256 variants of the reference.
//
// Reference (file:line in the reference checkout):
//   AggregatedDataVariants keys128 / keys256 = HashMap<UInt128 / UInt256, AggregateDataPtr, UInt128HashCRC32 / UInt256HashCRC32>
//                                      src/Interpreters/AggregatedDataVariants.h:70-71,83-84; AggregatedData.h:57-60
//   HashMethodKeysFixed::getKeyHolder  src/Common/ColumnsHashing/HashMethod.h:236-410  -> packFixed<Key>(row, ...)
//   packFixed                          src/Interpreters/AggregationCommon.h:91-158: the key columns' raw bytes laid consecutively
//   HashJoin keys128 / keys256         src/Interpreters/HashJoin/HashJoin.h:267-358 (same packing, same maps)
//
// Design (not a translation).  The 8-byte-key operators of this library (GROUP BY strategies, join build / probe, sharding) stay as
// they are; wide keys are first turned into dense 32-bit ids by a device-resident, exact dictionary:
//   table  tags u64[cap] (a 64-bit hash of the packed key, low bit forced to 1; 0 = empty) + ids u32[cap]
//   store  W x u64[store_cap]: the packed key of id k, written once by the row that claimed its cell
// One row = one probe walk over `tags` (8 bytes per cell: a cache line holds 8 cells, not 2-4 wide keys).  A row that finds an empty
// cell claims it with ONE compare-and-swap on the tag, takes the next id (one atomic per wave) and writes its key to the store; a
// row that finds its own tag is a candidate and is verified against the stored key by the NEXT kernel -- nobody ever waits for another
// lane inside a kernel, so the protocol cannot deadlock a lock-step wave.  Two different keys with one tag (2^-64 per pair) fail
// the verification and walk on in a further round; the result is exact.  Ids are stable for the dictionary's lifetime (a grown
// table re-inserts (tag, id) pairs, the store is append-only), so partial aggregation states keyed by id survive growth and blocks.
#include "chgpu_internal.h"

#include <cstdlib>

#include <vector>

static constexpr u32 KD_T = 256;
static constexpr u32 KD_MAX_COLS = 16;
static constexpr u32 KD_NO_ID = 0xFFFFFFFFu;
static constexpr u64 KD_CHUNK_ROWS = 64ull << 20; // rows encoded per pass: bounds the capacity the table must guarantee up front

struct KdCols
{
    u32 n;
    const void * ptr[KD_MAX_COLS];
    u32 size[KD_MAX_COLS];   // bytes per element: 1, 2, 4, 8
    u32 offset[KD_MAX_COLS]; // byte offset inside the packed key
};

struct KdCtrl
{
    u32 n_ids;
    u32 retry;
    u32 pad[2];
};

struct KdTable
{
    u64 * tags;
    u32 * ids;
    u64 capacity; // power of two
    u64 * store;  // [W][store_cap]
    u64 store_cap;
    u32 W;
    KdCtrl * ctrl;
};

struct chgpu_keydict
{
    chgpu_ctx * ctx = nullptr;
    u32 key_bytes = 16;
    u32 W = 2;
    KdTable t{};
    void * table_mem = nullptr;
    size_t table_class = 0;
    void * store_mem = nullptr;
    size_t store_class = 0;
    void * ctrl_mem = nullptr;
    size_t ctrl_class = 0;
    u64 n_ids = 0;
    int weak_tags = 0; // test hook: 20-bit tags, so that the collision rounds run
};

// packFixed (AggregationCommon.h:91-158): column j's element of row i copied to bytes [offset_j, offset_j + size_j) of the key
__global__ __launch_bounds__(KD_T) void k_kd_pack(KdCols c, u64 row_begin, u64 n, u32 W, u64 * __restrict__ out /* [W][n] */)
{
    for (u64 i = (u64)blockIdx.x * KD_T + threadIdx.x; i < n; i += (u64)gridDim.x * KD_T)
    {
        u64 w[4] = {0, 0, 0, 0};
        for (u32 j = 0; j < c.n; ++j)
        {
            u64 v;
            const u64 r = row_begin + i;
            switch (c.size[j])
            {
                case 1: v = ((const u8 *)c.ptr[j])[r]; break;
                case 2: v = ((const u16 *)c.ptr[j])[r]; break;
                case 4: v = ((const u32 *)c.ptr[j])[r]; break;
                default: v = ((const u64 *)c.ptr[j])[r]; break;
            }
            const u32 word = c.offset[j] >> 3, shift = (c.offset[j] & 7) * 8;
            w[word] |= v << shift;
            if (shift && shift + c.size[j] * 8 > 64)
                w[word + 1] |= v >> (64 - shift);
        }
        for (u32 q = 0; q < W; ++q)
            out[(u64)q * n + i] = w[q];
    }
}

__device__ __forceinline__ u64 kd_tag(const u64 * w, u32 W, int weak)
{
    u64 h = dev_intHash64(w[0] ^ 0x9E3779B97F4A7C15ull);
    for (u32 q = 1; q < W; ++q)
        h = dev_intHash64(h ^ w[q]);
    if (weak)
        h &= 0xFFFFF; // test hook: 20-bit tags, so that different keys do share tags and the verification rounds run
    return h | 1ull;
}

// mode: 1 = emplace (GROUP BY, join build), 0 = find (join probe: an absent key gets KD_NO_ID).
// round 0: every row; later rounds: the rows whose verification failed, continuing their walk at resume[i].
// cand[i] = the cell whose tag equals the row's (verified by k_kd_verify), or ~0 when the row is settled.
__global__ __launch_bounds__(KD_T) void k_kd_claim(KdTable t, const u64 * __restrict__ pk, u64 n, int mode, int round, int weak, u32 * __restrict__ rid, u64 * __restrict__ cand,
                                                   u64 * __restrict__ resume)
{
    const u64 mask = t.capacity - 1;
    const u64 stride = (u64)gridDim.x * KD_T;
    for (u64 i0 = (u64)blockIdx.x * KD_T; i0 < n; i0 += stride)
    {
        const u64 i = i0 + threadIdx.x;
        bool active = i < n && (round == 0 || cand[i] != ~0ull);
        u64 w[4] = {0, 0, 0, 0};
        u64 tag = 1, slot = 0;
        if (active)
        {
            for (u32 q = 0; q < t.W; ++q)
                w[q] = pk[(u64)q * n + i];
            tag = kd_tag(w, t.W, weak);
            slot = round == 0 ? ((tag >> 1) * 0x9E3779B97F4A7C15ull >> 20) & mask : resume[i];
            cand[i] = ~0ull;
        }
        // wave-synchronous walk: every iteration each unsettled lane looks at one cell; the lanes that claimed a cell in this iteration
        // take their ids with one atomic for the whole wave
        for (u64 step = 0; step <= t.capacity && __any(active); ++step)
        {
            bool claimed = false;
            if (active)
            {
                u64 cur = t.tags[slot];
                if (cur == 0)
                {
                    if (!mode)
                    {
                        rid[i] = KD_NO_ID; // findKey: not there
                        active = false;
                    }
                    else
                    {
                        cur = atomicCAS((unsigned long long *)&t.tags[slot], 0ull, (unsigned long long)tag);
                        claimed = cur == 0;
                    }
                }
                if (active && !claimed)
                {
                    if (cur == tag)
                    {
                        cand[i] = slot; // same tag: my key, or (2^-64) another one -- the next kernel compares the bytes
                        active = false;
                    }
                    else
                        slot = (slot + 1) & mask;
                }
            }
            const u64 claimers = __ballot(claimed);
            if (claimers)
            {
                u32 base = 0;
                const u32 lane = lane_id();
                const u32 leader = (u32)__ffsll((long long)claimers) - 1;
                if (lane == leader)
                    base = atomicAdd(&t.ctrl->n_ids, (u32)__popcll(claimers));
                base = __shfl(base, (int)leader, 64);
                if (claimed)
                {
                    const u32 id = base + mbcnt(claimers);
                    t.ids[slot] = id;
                    for (u32 q = 0; q < t.W; ++q)
                        t.store[(u64)q * t.store_cap + id] = w[q];
                    rid[i] = id;
                    active = false;
                }
            }
        }
    }
}

__global__ __launch_bounds__(KD_T) void k_kd_verify(KdTable t, const u64 * __restrict__ pk, u64 n, u32 * __restrict__ rid, u64 * __restrict__ cand, u64 * __restrict__ resume)
{
    const u64 mask = t.capacity - 1;
    bool any_retry = false;
    for (u64 i = (u64)blockIdx.x * KD_T + threadIdx.x; i < n; i += (u64)gridDim.x * KD_T)
    {
        const u64 slot = cand[i];
        if (slot == ~0ull)
            continue;
        const u32 id = t.ids[slot];
        bool same = true;
        for (u32 q = 0; q < t.W; ++q)
            same = same && t.store[(u64)q * t.store_cap + id] == pk[(u64)q * n + i];
        if (same)
        {
            rid[i] = id;
            cand[i] = ~0ull;
        }
        else
        {
            resume[i] = (slot + 1) & mask; // another key owns this tag here: walk on in the next round (cand[i] stays set)
            any_retry = true;
        }
    }
    if (__any(any_retry) && lane_id() == 0)
        atomicOr(&t.ctrl->retry, 1u);
}

// growth: every (tag, id) pair of the old table into the new one (all keys are distinct: first empty cell, no comparison)
__global__ __launch_bounds__(KD_T) void k_kd_rehash(const u64 * __restrict__ old_tags, const u32 * __restrict__ old_ids, u64 old_cap, KdTable t)
{
    const u64 mask = t.capacity - 1;
    for (u64 s = (u64)blockIdx.x * KD_T + threadIdx.x; s < old_cap; s += (u64)gridDim.x * KD_T)
    {
        const u64 tag = old_tags[s];
        if (tag == 0)
            continue;
        u64 slot = ((tag >> 1) * 0x9E3779B97F4A7C15ull >> 20) & mask;
        for (u64 step = 0; step <= t.capacity; ++step)
        {
            if (t.tags[slot] == 0 && atomicCAS((unsigned long long *)&t.tags[slot], 0ull, (unsigned long long)tag) == 0)
            {
                t.ids[slot] = old_ids[s];
                break;
            }
            slot = (slot + 1) & mask;
        }
    }
}

// insertKeyIntoColumns for one original key column: bytes [offset, offset + size) of the key of ids[i]; KD_NO_ID -> the type default 0
__global__ __launch_bounds__(KD_T) void k_kd_key_column(KdTable t, const u32 * __restrict__ ids, u64 n, u32 offset, u32 size, void * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * KD_T + threadIdx.x; i < n; i += (u64)gridDim.x * KD_T)
    {
        const u32 id = ids[i];
        u64 v = 0;
        if (id != KD_NO_ID)
        {
            const u32 word = offset >> 3, shift = (offset & 7) * 8;
            v = t.store[(u64)word * t.store_cap + id] >> shift;
            if (shift && shift + size * 8 > 64)
                v |= t.store[(u64)(word + 1) * t.store_cap + id] << (64 - shift);
        }
        switch (size)
        {
            case 1: ((u8 *)out)[i] = (u8)v; break;
            case 2: ((u16 *)out)[i] = (u16)v; break;
            case 4: ((u32 *)out)[i] = (u32)v; break;
            default: ((u64 *)out)[i] = v; break;
        }
    }
}

// UInt128HashCRC32 / UInt256HashCRC32 (src/Common/HashTable/Hash.h:346-355, 412-423): crc32c chained over the key's 64-bit words from
// seed -1 -> two-level bucket (TwoLevelHashTable.h:53) & (shards - 1): the shard of a wide key (ConcurrentHashJoin.cpp:426-440)
__global__ __launch_bounds__(KD_T) void k_kd_selector(KdTable t, const u32 * __restrict__ ids, u64 n, const u32 * __restrict__ lut, u32 shards_mask, u32 * __restrict__ sel)
{
    __shared__ u32 slut[8 * 256];
    for (u32 k = threadIdx.x; k < 8 * 256; k += KD_T)
        slut[k] = lut[k];
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * KD_T + threadIdx.x; i < n; i += (u64)gridDim.x * KD_T)
    {
        const u32 id = ids[i];
        u32 crc = 0xFFFFFFFFu;
        for (u32 q = 0; q < t.W; ++q)
        {
            const u64 x = id != KD_NO_ID ? t.store[(u64)q * t.store_cap + id] : 0;
            // crc(seed, x) = crc(seed, 0) ^ crc(0, x): the table part is linear in x, the seed part is eight zero bytes pushed through
            crc = dev_crc32c_zero8(crc) ^ dev_crc32c_tab(slut, x);
        }
        sel[i] = ((crc >> 24) & 0xFF) & shards_mask;
    }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
static int kd_alloc_table(chgpu_keydict * d, u64 cap, KdTable * t, void ** mem, size_t * cls)
{
    const size_t tags_b = (cap * 8 + 255) / 256 * 256;
    CHGPU_TRY(chgpu_pool_alloc(d->ctx, tags_b + cap * 4 + 256, mem, cls));
    t->tags = (u64 *)*mem;
    t->ids = (u32 *)((char *)*mem + tags_b);
    t->capacity = cap;
    CHGPU_HIP(hipMemsetAsync(t->tags, 0, cap * 8, d->ctx->stream));
    return CHGPU_OK;
}

// room for `extra` more keys: table at most half full, store large enough
static int kd_ensure(chgpu_keydict * d, u64 extra)
{
    chgpu_ctx * ctx = d->ctx;
    const u64 need = d->n_ids + extra;
    CHGPU_REQUIRE(need < KD_NO_ID, CHGPU_ERR_NOT_IMPLEMENTED, "more than 2^32 - 1 distinct wide keys: CPU path");
    if (!d->ctrl_mem)
    {
        CHGPU_TRY(chgpu_pool_alloc(ctx, 256, &d->ctrl_mem, &d->ctrl_class));
        CHGPU_HIP(hipMemsetAsync(d->ctrl_mem, 0, 256, ctx->stream));
        d->t.ctrl = (KdCtrl *)d->ctrl_mem;
    }
    if (need > d->t.store_cap)
    {
        u64 cap = d->t.store_cap ? d->t.store_cap : 1024;
        while (cap < need)
            cap *= 2;
        void * m = nullptr;
        size_t cls = 0;
        CHGPU_TRY(chgpu_pool_alloc(ctx, (size_t)cap * 8 * d->W, &m, &cls));
        for (u32 q = 0; q < d->W && d->n_ids; ++q)
            CHGPU_HIP(hipMemcpyAsync((u64 *)m + (u64)q * cap, d->t.store + (u64)q * d->t.store_cap, d->n_ids * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (d->store_mem)
            chgpu_pool_free(ctx, d->store_mem, d->store_class); // reuse is ordered behind the copies above (same stream)
        d->store_mem = m;
        d->store_class = cls;
        d->t.store = (u64 *)m;
        d->t.store_cap = cap;
    }
    if (2 * need > d->t.capacity)
    {
        u64 cap = d->t.capacity ? d->t.capacity : 2048;
        while (cap < 2 * need)
            cap *= 2;
        KdTable nt = d->t;
        void * m = nullptr;
        size_t cls = 0;
        CHGPU_TRY(kd_alloc_table(d, cap, &nt, &m, &cls));
        if (d->table_mem)
        {
            if (d->n_ids)
            {
                hipLaunchKernelGGL(k_kd_rehash, dim3(chgpu_grid_for(ctx, d->t.capacity, KD_T, 8)), dim3(KD_T), 0, ctx->stream, (const u64 *)d->t.tags, (const u32 *)d->t.ids,
                                   d->t.capacity, nt);
                ctx->counters[6] += 1;
                ctx->counters[7] += 1;
            }
            chgpu_pool_free(ctx, d->table_mem, d->table_class);
        }
        d->table_mem = m;
        d->table_class = cls;
        d->t = nt;
    }
    return CHGPU_OK;
}

extern "C" int chgpu_keydict_create(chgpu_ctx * ctx, uint32_t key_bytes, uint64_t size_hint, chgpu_keydict ** out)
{
    CHGPU_REQUIRE(ctx && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(key_bytes == 16 || key_bytes == 32, CHGPU_ERR_BAD_ARGUMENTS, "packed key of %u bytes: keys128 holds 16, keys256 holds 32", key_bytes);
    ChgpuDeviceGuard guard(ctx);
    chgpu_keydict * d = new chgpu_keydict();
    d->ctx = ctx;
    d->key_bytes = key_bytes;
    d->W = key_bytes / 8;
    d->t.W = d->W;
    d->weak_tags = chgpu_opt(ctx, "test_keydict_weak_tags", 0) ? 1 : 0; // test hook: 20-bit tags, so that tag collisions happen
    chgpu_ctx_retain(ctx);
    const int rc = kd_ensure(d, size_hint ? size_hint : 1024);
    if (rc != CHGPU_OK)
    {
        chgpu_keydict_free(d);
        return rc;
    }
    *out = d;
    return CHGPU_OK;
}

extern "C" int chgpu_keydict_free(chgpu_keydict * d)
{
    if (!d)
        return CHGPU_OK;
    ChgpuDeviceGuard guard(d->ctx);
    if (d->table_mem) chgpu_pool_free(d->ctx, d->table_mem, d->table_class);
    if (d->store_mem) chgpu_pool_free(d->ctx, d->store_mem, d->store_class);
    if (d->ctrl_mem) chgpu_pool_free(d->ctx, d->ctrl_mem, d->ctrl_class);
    chgpu_ctx * ctx = d->ctx;
    delete d;
    chgpu_ctx_release(ctx);
    return CHGPU_OK;
}

extern "C" int chgpu_keydict_size(chgpu_keydict * d, uint64_t * n_keys)
{
    CHGPU_REQUIRE(d && n_keys, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    *n_keys = d->n_ids;
    return CHGPU_OK;
}

extern "C" int chgpu_keydict_encode(chgpu_keydict * d, uint32_t n_cols, const chgpu_col * const * cols, uint64_t row_begin, uint64_t row_end, int insert,
                                    chgpu_col ** ids_u32)
{
    CHGPU_REQUIRE(d && cols && ids_u32 && n_cols >= 1, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(n_cols <= KD_MAX_COLS, CHGPU_ERR_NOT_IMPLEMENTED, "more than %u key columns: CPU path", KD_MAX_COLS);
    chgpu_ctx * ctx = d->ctx;
    ChgpuDeviceGuard guard(ctx);
    KdCols kc{};
    kc.n = n_cols;
    u32 off = 0;
    for (u32 j = 0; j < n_cols; ++j)
    {
        CHGPU_REQUIRE(cols[j], CHGPU_ERR_BAD_ARGUMENTS, "key column %u is NULL", j);
        CHGPU_REQUIRE(chgpu_type_is_int(cols[j]->type), CHGPU_ERR_NOT_IMPLEMENTED, "key column %u has type %d: fixed-width integer keys only", j, cols[j]->type);
        CHGPU_REQUIRE(cols[j]->rows == cols[0]->rows, CHGPU_ERR_SIZES_MISMATCH, "key columns of different lengths");
        kc.ptr[j] = cols[j]->data;
        kc.size[j] = (u32)chgpu_type_size(cols[j]->type);
        kc.offset[j] = off; // packFixed: consecutively, no alignment padding
        off += kc.size[j];
    }
    CHGPU_REQUIRE(off <= d->key_bytes, CHGPU_ERR_BAD_ARGUMENTS, "the key columns take %u bytes, the dictionary packs %u", off, d->key_bytes);
    CHGPU_REQUIRE(row_begin <= row_end && row_end <= cols[0]->rows, CHGPU_ERR_BAD_ARGUMENTS, "row range out of bounds");
    const u64 n = row_end - row_begin;
    chgpu_col * out = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U32, n, &out));
    auto fail = [&](int code) {
        chgpu_col_free(out);
        return code;
    };
    int rc = CHGPU_OK;
    for (u64 c0 = 0; c0 < n && rc == CHGPU_OK; c0 += KD_CHUNK_ROWS)
    {
        const u64 m = n - c0 < KD_CHUNK_ROWS ? n - c0 : KD_CHUNK_ROWS;
        if (insert && (rc = kd_ensure(d, m)) != CHGPU_OK)
            break;
        auto al = [](size_t b) { return (b + 255) / 256 * 256; };
        void * scratch = nullptr;
        if ((rc = chgpu_scratch(ctx, al(m * 8 * d->W) + 2 * al(m * 8), &scratch)) != CHGPU_OK)
            break;
        u64 * pk = (u64 *)scratch;
        u64 * cand = (u64 *)((char *)scratch + al(m * 8 * d->W));
        u64 * resume = (u64 *)((char *)cand + al(m * 8));
        u32 * rid = (u32 *)out->data + c0;
        const u32 grid = chgpu_grid_for(ctx, m, KD_T, 8);
        hipLaunchKernelGGL(k_kd_pack, dim3(grid), dim3(KD_T), 0, ctx->stream, kc, row_begin + c0, m, d->W, pk);
        ctx->counters[6] += 1;
        for (int round = 0; round < 64; ++round)
        {
            hipLaunchKernelGGL(k_kd_claim, dim3(grid), dim3(KD_T), 0, ctx->stream, d->t, (const u64 *)pk, m, insert ? 1 : 0, round, d->weak_tags, rid, cand, resume);
            hipLaunchKernelGGL(k_kd_verify, dim3(grid), dim3(KD_T), 0, ctx->stream, d->t, (const u64 *)pk, m, rid, cand, resume);
            ctx->counters[6] += 2;
            KdCtrl c;
            if ((rc = chgpu_read_back(ctx, d->t.ctrl, &c, sizeof(c))) != CHGPU_OK)
                break;
            d->n_ids = c.n_ids;
            if (!c.retry)
                break;
            // some rows met another key under their tag: they walk on from the next cell
            if (hipMemsetAsync(&d->t.ctrl->retry, 0, 4, ctx->stream) != hipSuccess)
                rc = CHGPU_ERR_DEVICE;
            if (round == 63)
                rc = chgpu_set_error(CHGPU_ERR_LOGICAL, "wide-key dictionary did not settle in 64 rounds");
        }
    }
    if (rc == CHGPU_OK && hipGetLastError() != hipSuccess)
        rc = chgpu_set_error(CHGPU_ERR_DEVICE, "keydict launch failed");
    if (rc != CHGPU_OK)
        return fail(rc);
    *ids_u32 = out;
    return CHGPU_OK;
}

extern "C" int chgpu_keydict_key_column(chgpu_keydict * d, const chgpu_col * ids_u32, uint32_t byte_offset, int type, chgpu_col ** out)
{
    CHGPU_REQUIRE(d && ids_u32 && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(ids_u32->type == CHGPU_U32, CHGPU_ERR_BAD_ARGUMENTS, "ids must be UInt32");
    const size_t es = chgpu_type_size(type);
    CHGPU_REQUIRE(es && chgpu_type_is_int(type) && byte_offset + es <= d->key_bytes, CHGPU_ERR_BAD_ARGUMENTS, "key part [%u, +%zu) of type %d", byte_offset, es, type);
    chgpu_ctx * ctx = d->ctx;
    ChgpuDeviceGuard guard(ctx);
    chgpu_col * c = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, type, ids_u32->rows, &c));
    if (ids_u32->rows)
    {
        hipLaunchKernelGGL(k_kd_key_column, dim3(chgpu_grid_for(ctx, ids_u32->rows, KD_T, 8)), dim3(KD_T), 0, ctx->stream, d->t, (const u32 *)ids_u32->data, ids_u32->rows, byte_offset,
                           (u32)es, c->data);
        ctx->counters[6] += 1;
        if (hipGetLastError() != hipSuccess)
        {
            chgpu_col_free(c);
            return chgpu_set_error(CHGPU_ERR_DEVICE, "keydict launch failed");
        }
    }
    *out = c;
    return CHGPU_OK;
}

extern "C" int chgpu_keydict_selector(chgpu_keydict * d, const chgpu_col * ids_u32, uint32_t num_shards, chgpu_col ** selector_u32)
{
    CHGPU_REQUIRE(d && ids_u32 && selector_u32, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(ids_u32->type == CHGPU_U32, CHGPU_ERR_BAD_ARGUMENTS, "ids must be UInt32");
    CHGPU_REQUIRE(num_shards >= 1 && num_shards <= 256 && (num_shards & (num_shards - 1)) == 0, CHGPU_ERR_BAD_ARGUMENTS, "num_shards must be a power of two <= 256");
    chgpu_ctx * ctx = d->ctx;
    ChgpuDeviceGuard guard(ctx);
    const u32 * lut = nullptr;
    CHGPU_TRY(chgpu_crc_lut(ctx, &lut));
    chgpu_col * c = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U32, ids_u32->rows, &c));
    if (ids_u32->rows)
    {
        hipLaunchKernelGGL(k_kd_selector, dim3(chgpu_grid_for(ctx, ids_u32->rows, KD_T, 8)), dim3(KD_T), 0, ctx->stream, d->t, (const u32 *)ids_u32->data, ids_u32->rows, lut,
                           num_shards - 1, (u32 *)c->data);
        ctx->counters[6] += 1;
    }
    *selector_u32 = c;
    return CHGPU_OK;
}
