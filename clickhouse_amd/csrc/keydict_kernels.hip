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
//   table  cells {u64 tag, u64 id + 1}[cap]: tag = a 64-bit hash of the packed key, low bit forced to 1 (0 = empty); a 64-byte line holds 4 cells
//   store  u64[store_cap][W]: the packed key of id k, written once by the row that claimed its cell
// One row = one probe walk over the cells, a line (4 cells) per look.  A row that finds an empty
// cell claims it with ONE compare-and-swap on the tag, takes the next id (one atomic per wave) and writes its key to the store; a
// row that finds its own tag is a candidate and is verified against the stored key by the NEXT kernel -- nobody ever waits for another
// lane inside a kernel, so the protocol cannot deadlock a lock-step wave.  Two different keys with one tag (2^-64 per pair) fail
// the verification and walk on in a further round; the result is exact.  Ids are stable for the dictionary's lifetime (a grown
// table re-inserts (tag, id) pairs, the store is append-only), so partial aggregation states keyed by id survive growth and blocks.
//
// Two things keep the table small enough to stay in the L2 / Infinity Cache instead of being sized for "every row is a new key":
//   * a key inserted by an EARLIER kernel (its id is below the count the host read back before this launch) is compared in place --
//     its tag, id and stored bytes are all visible -- so in the steady state (and always on the find side) no row needs the second kernel;
//   * the table is sized for the keys it holds, not for the rows of the chunk: a row that would insert beyond `limit` (half the cells)
//     defers instead; the host grows the table fourfold and runs the deferred rows again.  Between a lane's look at the counter and its
//     claim at most one grid of lanes can slip through, so cells and store hold `limit` + one grid's worth of lanes (kd_reserve).
#include "chgpu_internal.h"

#include <cstdlib>

#include <vector>

static constexpr u32 KD_T = 256;
static constexpr u32 KD_MAX_COLS = 16;
static constexpr u32 KD_NO_ID = 0xFFFFFFFFu;
static constexpr u64 KD_CHUNK_ROWS = 64ull << 20; // rows encoded per pass (bounds the scratch: 16 bytes per row)
static constexpr u64 KD_FIRST_CHUNK_ROWS = 4ull << 20;

struct KdCols
{
    u32 n;
    const void * ptr[KD_MAX_COLS];
    u32 size[KD_MAX_COLS];   // bytes per element: 1, 2, 4, 8
    u32 offset[KD_MAX_COLS]; // byte offset inside the packed key
    u32 words8;              // every key column is 8 bytes wide: column j IS word j of the key
};

struct KdCtrl
{
    u32 n_ids;
    u32 retry;       // bit 0: a row walks on in a further round (another key under its tag); bit 1: rows deferred, the table must grow
    u32 need_verify; // some row met a tag claimed in this very kernel: k_kd_verify has work
    u32 need_walk;   // k_kd_lookup left rows for k_kd_claim
};

static constexpr u64 KD_SETTLED = ~0ull;
static constexpr u64 KD_DEFERRED = ~0ull - 1; // cand[i]: the row found an empty cell while the dictionary was at its limit

struct KdTable
{
    ulonglong2 * cells; // cell = cs x 16 bytes: {tag (0: empty), id + 1 (0: the claimer has not written it yet)}, then the packed key itself
    u32 cs;             // 16-byte units per cell: 1 + W / 2 (keys128: 32-byte cells, two per 64-byte line; keys256: 48 bytes)
    u64 capacity; // power of two
    u64 * store;  // [store_cap][W]: the packed key of id k, one 16- or 32-byte line
    u64 store_cap;
    u32 W;
    u32 ids_before; // keys inserted by earlier kernels: ids below this are compared in place
    u64 limit;      // inserts stop (rows defer) once n_ids reaches this
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

// packFixed (AggregationCommon.h:91-158): column j's element of row r copied to bytes [offset_j, offset_j + size_j) of the key.  The
// packed key is never written out: k_kd_claim and k_kd_verify assemble it from the key columns where they need it.
__device__ __forceinline__ void kd_pack_row(const KdCols & c, u64 r, u64 (&w)[4])
{
    if (c.words8)
    {
        w[0] = ((const u64 *)c.ptr[0])[r];
        w[1] = c.n > 1 ? ((const u64 *)c.ptr[1])[r] : 0;
        w[2] = c.n > 2 ? ((const u64 *)c.ptr[2])[r] : 0;
        w[3] = c.n > 3 ? ((const u64 *)c.ptr[3])[r] : 0;
        return;
    }
    w[0] = w[1] = w[2] = w[3] = 0;
    for (u32 j = 0; j < c.n; ++j)
    {
        u64 v;
        switch (c.size[j])
        {
            case 1: v = ((const u8 *)c.ptr[j])[r]; break;
            case 2: v = ((const u16 *)c.ptr[j])[r]; break;
            case 4: v = ((const u32 *)c.ptr[j])[r]; break;
            default: v = ((const u64 *)c.ptr[j])[r]; break;
        }
        const u32 word = c.offset[j] >> 3, shift = (c.offset[j] & 7) * 8;
        // static indexes only: a run-time index into w[] would put it in scratch memory
        const u64 lo = v << shift, hi = shift && shift + c.size[j] * 8 > 64 ? v >> (64 - shift) : 0;
        w[0] |= word == 0 ? lo : 0;
        w[1] |= word == 1 ? lo : word == 0 ? hi : 0;
        w[2] |= word == 2 ? lo : word == 1 ? hi : 0;
        w[3] |= word == 3 ? lo : word == 2 ? hi : 0;
    }
}

__device__ __forceinline__ u64 kd_tag(const u64 * w, u32 W, int weak)
{
    // the dictionary's own tag, not one of the reference's hashes: one multiply + xor-shift per word (every step a bijection of the
    // running value, so two keys share a tag only by a 2^-64 accident of the mixing, and then k_kd_verify tells them apart)
    u64 h = (w[0] ^ 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    h ^= h >> 31;
    for (u32 q = 1; q < W; ++q)
    {
        h = (h ^ w[q]) * 0x94D049BB133111EBull;
        h ^= h >> 29;
    }
    if (weak)
        h &= 0xFFFFF; // test hook: 20-bit tags, so that different keys do share tags and the verification rounds run
    return h | 1ull;
}

// The dictionary as it stood before this chunk, looked up with no loop and no atomics: U rows per lane, every load unconditional, so a
// lane has U key loads, then U cell loads, then U stored keys in flight (k_kd_claim's walk has one -- it spends 72 % of its wave cycles
// parked on s_waitcnt, profiles/r03_keys128.json).  A row is settled when its HOME cell holds its key (or, on the find side, is empty);
// every other row -- a new key, a displaced one -- is left to k_kd_claim as a deferred row.  Used while the cells are sparse (kd_encode).
template <u32 WW, u32 U>
__global__ __launch_bounds__(KD_T) void k_kd_lookup(KdTable t, KdCols kc, u64 row_begin, u64 n, int mode, int weak, u32 * __restrict__ rid, u64 * __restrict__ cand)
{
    const u64 mask = t.capacity - 1;
    const u64 stride = (u64)gridDim.x * KD_T;
    u32 walk = 0;
    for (u64 i0 = (u64)blockIdx.x * KD_T + threadIdx.x; i0 < n; i0 += stride * U)
    {
        constexpr u32 CS = 1 + WW / 2;
        u64 w[U][4];
        u64 tag[U];
        ulonglong2 c[U][CS];
#pragma unroll
        for (u32 u = 0; u < U; ++u)
        {
            const u64 i = i0 + u * stride;
            kd_pack_row(kc, row_begin + (i < n ? i : n - 1), w[u]);
        }
#pragma unroll
        for (u32 u = 0; u < U; ++u)
        {
            tag[u] = kd_tag(w[u], WW, weak);
            const ulonglong2 * cp = t.cells + (((tag[u] >> 1) * 0x9E3779B97F4A7C15ull >> 20) & mask) * CS;
#pragma unroll
            for (u32 q = 0; q < CS; ++q)
                c[u][q] = cp[q];
        }
#pragma unroll
        for (u32 u = 0; u < U; ++u)
        {
            const u64 i = i0 + u * stride;
            if (i >= n)
                continue;
            bool same = c[u][0].x == tag[u] && c[u][0].y - 1 < t.ids_before;
#pragma unroll
            for (u32 q = 1; q < CS; ++q)
                same = same && c[u][q].x == w[u][2 * q - 2] && c[u][q].y == w[u][2 * q - 1];
            if (same)
            {
                rid[i] = (u32)(c[u][0].y - 1);
                cand[i] = KD_SETTLED;
            }
            else if (!mode && c[u][0].x == 0)
            {
                rid[i] = KD_NO_ID; // findKey: the home cell is empty
                cand[i] = KD_SETTLED;
            }
            else
            {
                cand[i] = KD_DEFERRED;
                walk = 1;
            }
        }
    }
    if (__any(walk != 0) && lane_id() == 0)
        atomicOr(&t.ctrl->need_walk, 1u);
}

// mode: 1 = emplace (GROUP BY, join build), 0 = find (join probe: an absent key gets KD_NO_ID).
// round 0: every row; later rounds: the rows that are not settled -- a row whose verification failed continues its walk at resume[i], a
// deferred row (and every unsettled row after the table has grown: `restart`) starts again at its home cell.
// cand[i] = the cell whose tag equals the row's and whose key this kernel cannot see yet (k_kd_verify compares), KD_DEFERRED, or KD_SETTLED.
__global__ __launch_bounds__(KD_T) void k_kd_claim(KdTable t, KdCols kc, u64 row_begin, u64 n, int mode, int round, int restart, int weak, int gated, u32 * __restrict__ rid,
                                                   u64 * __restrict__ cand, u64 * __restrict__ resume)
{
    if (gated && !t.ctrl->need_walk) // right behind k_kd_lookup, which settled every row
        return;
    const u64 mask = t.capacity - 1;
    const u64 stride = (u64)gridDim.x * KD_T;
    u32 flags = 0, verify = 0;
    // what this wave knows of the id counter: the host's count at launch, then whatever its own claims returned.  A wave claims with a
    // stale count at most once (the claim tells it the true one), so all waves together pass the limit by at most one grid of lanes
    u32 known_ids = t.ids_before;
    for (u64 i0 = (u64)blockIdx.x * KD_T; i0 < n; i0 += stride)
    {
        const u64 i = i0 + threadIdx.x;
        const u64 was = round != 0 && i < n ? cand[i] : 0;
        bool active = i < n && (round == 0 || was != KD_SETTLED);
        u64 w[4] = {0, 0, 0, 0};
        u64 tag = 1, slot = 0;
        if (active)
        {
            kd_pack_row(kc, row_begin + i, w);
            tag = kd_tag(w, t.W, weak);
            slot = (round == 0 || restart || was == KD_DEFERRED) ? ((tag >> 1) * 0x9E3779B97F4A7C15ull >> 20) & mask : resume[i];
        }
        u64 state = KD_SETTLED;
        // wave-synchronous walk: every iteration each unsettled lane looks at one cell; the lanes that claimed a cell in this iteration
        // take their ids with one atomic for the whole wave
        for (u64 step = 0; step <= t.capacity && __any(active); ++step)
        {
            bool claimed = false;
            // linear probing; keys128 looks at a 64-byte line (its two cells) at a time: the first cell at or after `slot` that is empty or
            // holds the row's tag is where the cell-by-cell walk would stop.  (A stale line can only show an empty cell where a tag has
            // landed since -- the compare-and-swap below then returns the tag.)  The cell carries the key: a hit is ONE random read.
            bool at_cell = false;
            u64 cur = 0, idp1 = 0;
            u64 kw[4] = {0, 0, 0, 0}; // the key stored in the cell looked at
            if (active && t.W == 2)
            {
                const u64 line = slot & ~1ull;
                const ulonglong2 * lp = t.cells + line * 2;
                const ulonglong2 a0 = lp[0], a1 = lp[1], b0 = lp[2], b1 = lp[3];
                const u32 same = (u32)(a0.x == tag) | (u32)(b0.x == tag) << 1;
                const u32 empty = (u32)(a0.x == 0) | (u32)(b0.x == 0) << 1;
                const u32 ev = (same | empty) & (3u << (slot & 1));
                if (ev)
                {
                    const u32 k = (u32)__ffs((int)ev) - 1;
                    slot = line + k;
                    cur = (same >> k) & 1 ? tag : 0;
                    idp1 = k ? b0.y : a0.y;
                    kw[0] = k ? b1.x : a1.x;
                    kw[1] = k ? b1.y : a1.y;
                    at_cell = true;
                }
                else
                    slot = (line + 2) & mask;
            }
            else if (active)
            {
                const ulonglong2 * lp = t.cells + slot * 3;
                const ulonglong2 a0 = lp[0], a1 = lp[1], a2 = lp[2];
                if (a0.x == tag || a0.x == 0)
                {
                    cur = a0.x;
                    idp1 = a0.y;
                    kw[0] = a1.x;
                    kw[1] = a1.y;
                    kw[2] = a2.x;
                    kw[3] = a2.y;
                    at_cell = true;
                }
                else
                    slot = (slot + 1) & mask;
            }
            if (at_cell)
            {
                if (cur == 0)
                {
                    if (!mode)
                    {
                        rid[i] = KD_NO_ID; // findKey: not there
                        active = false;
                    }
                    else if (known_ids >= t.limit)
                    {
                        state = KD_DEFERRED; // the table grows first
                        flags |= 2;
                        active = false;
                    }
                    else
                    {
                        cur = atomicCAS((unsigned long long *)&t.cells[slot * t.cs].x, 0ull, (unsigned long long)tag);
                        claimed = cur == 0;
                        idp1 = 0; // if the cell was taken meanwhile its id is not known here
                    }
                }
                if (active && !claimed)
                {
                    if (cur == tag)
                    {
                        // same tag: my key, or (2^-64) another one.  A key of an earlier kernel is compared here; one claimed in this
                        // kernel (its id or bytes may not be visible yet) by k_kd_verify
                        const u64 id = idp1 - 1; // 2^64 - 1 while the id is not visible
                        if (id < t.ids_before)
                        {
                            const bool same = kw[0] == w[0] && kw[1] == w[1] && kw[2] == w[2] && kw[3] == w[3]; // (words past W are 0 on both sides)
                            if (same)
                            {
                                rid[i] = (u32)id;
                                active = false;
                            }
                            else
                                slot = (slot + 1) & mask;
                        }
                        else
                        {
                            state = slot;
                            verify = 1;
                            active = false;
                        }
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
                known_ids = base + (u32)__popcll(claimers);
                if (claimed)
                {
                    const u32 id = base + mbcnt(claimers);
                    ulonglong2 * cp = t.cells + slot * t.cs;
                    cp[0].y = (u64)id + 1;
                    cp[1] = make_ulonglong2(w[0], w[1]);
                    if (t.W == 4)
                        cp[2] = make_ulonglong2(w[2], w[3]);
                    if (id < t.store_cap)
                        for (u32 q = 0; q < t.W; ++q)
                            t.store[(u64)id * t.W + q] = w[q];
                    else
                        flags |= 4; // cannot happen while kd_reserve's bound holds; never write past the store
                    rid[i] = id;
                    active = false;
                }
            }
        }
        if (i < n && (round == 0 || was != KD_SETTLED))
            cand[i] = state;
    }
    if (__any(flags != 0))
    {
        for (int o = 32; o > 0; o >>= 1)
            flags |= __shfl_xor(flags, o, 64);
        if (lane_id() == 0)
            atomicOr(&t.ctrl->retry, flags);
    }
    if (__any(verify != 0) && lane_id() == 0)
        atomicOr(&t.ctrl->need_verify, 1u);
}

__global__ __launch_bounds__(KD_T) void k_kd_verify(KdTable t, KdCols kc, u64 row_begin, u64 n, u32 * __restrict__ rid, u64 * __restrict__ cand, u64 * __restrict__ resume)
{
    if (!t.ctrl->need_verify) // written by the kernel before this one
        return;
    const u64 mask = t.capacity - 1;
    bool any_retry = false;
    for (u64 i = (u64)blockIdx.x * KD_T + threadIdx.x; i < n; i += (u64)gridDim.x * KD_T)
    {
        const u64 slot = cand[i];
        if (slot == KD_SETTLED || slot == KD_DEFERRED)
            continue;
        const ulonglong2 * cp = t.cells + slot * t.cs;
        const u64 id = cp[0].y - 1;
        u64 w[4];
        kd_pack_row(kc, row_begin + i, w);
        const ulonglong2 k01 = cp[1];
        bool same = k01.x == w[0] && k01.y == w[1];
        if (t.W == 4)
        {
            const ulonglong2 k23 = cp[2];
            same = same && k23.x == w[2] && k23.y == w[3];
        }
        if (same)
        {
            rid[i] = (u32)id;
            cand[i] = KD_SETTLED;
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
__global__ __launch_bounds__(KD_T) void k_kd_rehash(const ulonglong2 * __restrict__ old_cells, u64 old_cap, KdTable t)
{
    const u64 mask = t.capacity - 1;
    for (u64 s = (u64)blockIdx.x * KD_T + threadIdx.x; s < old_cap; s += (u64)gridDim.x * KD_T)
    {
        const ulonglong2 * oc = old_cells + s * t.cs;
        const ulonglong2 c = oc[0];
        if (c.x == 0)
            continue;
        u64 slot = ((c.x >> 1) * 0x9E3779B97F4A7C15ull >> 20) & mask;
        for (u64 step = 0; step <= t.capacity; ++step)
        {
            ulonglong2 * nc = t.cells + slot * t.cs;
            if (nc[0].x == 0 && atomicCAS((unsigned long long *)&nc[0].x, 0ull, (unsigned long long)c.x) == 0)
            {
                nc[0].y = c.y;
                for (u32 q = 1; q < t.cs; ++q)
                    nc[q] = oc[q];
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
            v = t.store[(u64)id * t.W + word] >> shift;
            if (shift && shift + size * 8 > 64)
                v |= t.store[(u64)id * t.W + word + 1] << (64 - shift);
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
            const u64 x = id != KD_NO_ID ? t.store[(u64)id * t.W + q] : 0;
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
    t->cs = 1 + d->W / 2;
    CHGPU_TRY(chgpu_pool_alloc(d->ctx, cap * 16 * t->cs + 256, mem, cls));
    t->cells = (ulonglong2 *)*mem;
    t->capacity = cap;
    t->limit = cap / 2;
    CHGPU_HIP(hipMemsetAsync(t->cells, 0, cap * 16 * t->cs, d->ctx->stream));
    return CHGPU_OK;
}

// Cells and store for `want` keys.  Inserts stop at limit = capacity / 2; `overshoot` = the lanes of one k_kd_claim grid, which may pass
// the limit check together before any of them has counted itself: cells stay at most 3/4 full, the store holds limit + overshoot keys.
static int kd_reserve(chgpu_keydict * d, u64 want, u64 overshoot)
{
    chgpu_ctx * ctx = d->ctx;
    if (want < d->n_ids)
        want = d->n_ids;
    if (!d->ctrl_mem)
    {
        CHGPU_TRY(chgpu_pool_alloc(ctx, 256, &d->ctrl_mem, &d->ctrl_class));
        CHGPU_HIP(hipMemsetAsync(d->ctrl_mem, 0, 256, ctx->stream));
        d->t.ctrl = (KdCtrl *)d->ctrl_mem;
    }
    u64 cap = d->t.capacity ? d->t.capacity : 2048;
    while (cap < 2 * want || cap < 4 * overshoot)
        cap *= 2;
    CHGPU_REQUIRE(cap / 2 + overshoot < KD_NO_ID, CHGPU_ERR_NOT_IMPLEMENTED, "more than 2^32 - 1 distinct wide keys: CPU path");
    if (cap > d->t.capacity)
    {
        KdTable nt = d->t;
        void * m = nullptr;
        size_t cls = 0;
        CHGPU_TRY(kd_alloc_table(d, cap, &nt, &m, &cls));
        if (d->table_mem)
        {
            if (d->n_ids)
            {
                hipLaunchKernelGGL(k_kd_rehash, dim3(chgpu_grid_for(ctx, d->t.capacity, KD_T, 8)), dim3(KD_T), 0, ctx->stream, (const ulonglong2 *)d->t.cells, d->t.capacity, nt);
                ctx->counters[6] += 1;
                ctx->counters[7] += 1;
            }
            chgpu_pool_free(ctx, d->table_mem, d->table_class);
        }
        d->table_mem = m;
        d->table_class = cls;
        d->t = nt;
    }
    const u64 need = d->t.limit + overshoot;
    if (need > d->t.store_cap)
    {
        u64 scap = d->t.store_cap ? d->t.store_cap : 1024;
        while (scap < need)
            scap *= 2;
        void * m = nullptr;
        size_t cls = 0;
        CHGPU_TRY(chgpu_pool_alloc(ctx, (size_t)scap * 8 * d->W, &m, &cls));
        if (d->n_ids)
            CHGPU_HIP(hipMemcpyAsync(m, d->t.store, d->n_ids * 8 * d->W, hipMemcpyDeviceToDevice, ctx->stream));
        if (d->store_mem)
            chgpu_pool_free(ctx, d->store_mem, d->store_class); // reuse is ordered behind the copy above (same stream)
        d->store_mem = m;
        d->store_class = cls;
        d->t.store = (u64 *)m;
        d->t.store_cap = scap;
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
    const int rc = kd_reserve(d, size_hint ? size_hint : 1024, 0);
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
    kc.words8 = n_cols <= 4;
    for (u32 j = 0; j < n_cols; ++j)
        kc.words8 = kc.words8 && kc.size[j] == 8;
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
    // emplace: a short first chunk, then longer ones -- a key met in an earlier chunk is compared in place by k_kd_claim, only rows that meet
    // a key claimed in their own chunk go through k_kd_verify, and most keys of a block show up in its first few million rows
    u64 chunk = insert ? KD_FIRST_CHUNK_ROWS : KD_CHUNK_ROWS;
    for (u64 c0 = 0, m = 0; c0 < n && rc == CHGPU_OK; c0 += m, chunk = chunk * 4 < KD_CHUNK_ROWS ? chunk * 4 : KD_CHUNK_ROWS)
    {
        m = n - c0 < chunk + chunk / 2 ? n - c0 : chunk; // a short tail joins the last chunk
        const u32 grid = chgpu_grid_for(ctx, m, KD_T, 8);
        if (insert && (rc = kd_reserve(d, d->n_ids, (u64)grid * KD_T)) != CHGPU_OK)
            break;
        auto al = [](size_t b) { return (b + 255) / 256 * 256; };
        void * scratch = nullptr;
        if ((rc = chgpu_scratch(ctx, 2 * al(m * 8), &scratch)) != CHGPU_OK)
            break;
        u64 * cand = (u64 *)scratch;
        u64 * resume = (u64 *)((char *)cand + al(m * 8));
        u32 * rid = (u32 *)out->data + c0;
        int restart = 0;
        // cells at most 1/8 full and a dictionary that holds something: most rows find their key in their home cell -- the loop-free
        // look-up settles those, k_kd_claim then sees the chunk as a later round does (only the rows left over; none: it returns at once)
        const bool lookup_first = d->n_ids != 0 && d->n_ids * 8 <= d->t.capacity;
        if (lookup_first)
        {
            d->t.ids_before = (u32)d->n_ids;
            if (d->W == 2)
                hipLaunchKernelGGL((k_kd_lookup<2, 4>), dim3(grid), dim3(KD_T), 0, ctx->stream, d->t, kc, row_begin + c0, m, insert ? 1 : 0, d->weak_tags, rid, cand);
            else
                hipLaunchKernelGGL((k_kd_lookup<4, 2>), dim3(grid), dim3(KD_T), 0, ctx->stream, d->t, kc, row_begin + c0, m, insert ? 1 : 0, d->weak_tags, rid, cand);
            ctx->counters[6] += 1;
        }
        for (int round = 0; round < 96; ++round)
        {
            d->t.ids_before = (u32)d->n_ids;
            hipLaunchKernelGGL(k_kd_claim, dim3(grid), dim3(KD_T), 0, ctx->stream, d->t, kc, row_begin + c0, m, insert ? 1 : 0, round + (lookup_first ? 1 : 0), restart, d->weak_tags,
                               lookup_first && round == 0 ? 1 : 0, rid, cand, resume);
            hipLaunchKernelGGL(k_kd_verify, dim3(grid), dim3(KD_T), 0, ctx->stream, d->t, kc, row_begin + c0, m, rid, cand, resume);
            ctx->counters[6] += 2;
            KdCtrl c;
            if ((rc = chgpu_read_back(ctx, d->t.ctrl, &c, sizeof(c))) != CHGPU_OK)
                break;
            d->n_ids = c.n_ids;
            if (c.retry & 4)
            {
                rc = chgpu_set_error(CHGPU_ERR_LOGICAL, "wide-key dictionary: ids ran past the store");
                break;
            }
            if (!c.retry)
            {
                if ((c.need_verify || c.need_walk) && hipMemsetAsync(&d->t.ctrl->need_verify, 0, 8, ctx->stream) != hipSuccess)
                    rc = CHGPU_ERR_DEVICE;
                break;
            }
            // bit 0: some rows met another key under their tag and walk on from the next cell; bit 1: rows deferred at the limit -- the
            // table grows fourfold and every unsettled row starts its walk again in the new cells
            if (hipMemsetAsync(&d->t.ctrl->retry, 0, 12, ctx->stream) != hipSuccess)
                rc = CHGPU_ERR_DEVICE;
            restart = 0;
            if (rc == CHGPU_OK && (c.retry & 2))
            {
                rc = kd_reserve(d, 4 * d->t.limit, (u64)grid * KD_T);
                restart = 1;
            }
            if (rc == CHGPU_OK && round == 95)
                rc = chgpu_set_error(CHGPU_ERR_LOGICAL, "wide-key dictionary did not settle in 96 rounds");
            if (rc != CHGPU_OK)
                break;
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
