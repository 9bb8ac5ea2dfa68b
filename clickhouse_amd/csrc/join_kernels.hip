// join_kernels.hip — hash join build and probe on the device for one numeric key (HashJoin key64 family).
//
// Reference loops replaced (file:line in the reference checkout):
//   k_join_insert / k_join_fill   HashJoin::addBlockToJoin -> insertFromBlockImplTypeCase, Inserter::insertOne/insertAll,
//                                 RowRefList::insert            src/Interpreters/HashJoin/HashJoin.cpp:556-768,
//                                                               HashJoinMethodsImpl.h:220-281, HashJoinMethods.h:16-62, RowRefs.h:16-141
//   k_join_probe_* / k_join_emit  HashJoin::joinBlock -> joinRightColumns (findKey, addFoundRowAll, addNotFoundRow,
//                                 offsets_to_replicate, filter, max_joined_block_rows)
//                                                               HashJoin.cpp:1036-1101, HashJoinMethodsImpl.h:402-549,
//                                                               KnownRowsHolder.h:88-153, JoinFeatures.h:9-40
//
// Design (not a translation): RowRef{block*,row} becomes a 64-bit global row id (block_index << 32 | row); the
// RowRefList's arena-allocated 7-slot batches become one CSR array (per-key count -> exclusive scan -> fill); the
// serial "first row owns the cell" rule becomes atomicMin (atomicMax for any_take_last_row) on the cell's row id; the
// INNER ANY "each right row joins its first left row only" flag (setUsedOnce) becomes atomicMin of a running left-row
// sequence number.  The table is built once in chgpu_join_finish_build (IJoin::onBuildPhaseFinish), sized from the
// number of build rows, so no kernel ever has to grow it.
#include "chgpu_internal.h"
#include "radix_partition.h"

#include <cstdlib>

#include <vector>

static constexpr u32 JT = 256;
static constexpr u64 NO_ROW = ~0ull;
static constexpr u32 NO_SLOT = ~0u;
// packed probe value of a cell with several rows: bit 63 | min(count, SAT) << 40 | CSR start (40 bits)
static constexpr u64 JV_MULTI = 1ull << 63;
static constexpr u64 JV_CNT_SAT = (1ull << 23) - 1;
static constexpr u64 JV_START_MASK = (1ull << 40) - 1;

struct JoinCtrl
{
    unsigned long long n_keys;
    u32 has_zero;
    u32 pad;
    u64 consumed; // probe: left rows consumed
    u64 n_out;    // probe: appended rows
    unsigned long long max_key; // largest non-zero build key (picks the prefilter's mode)
};

struct JoinTable
{
    u64 * kv;        // [cap+1][2] interleaved {key (0 = empty; cell cap = the zero key), value}: a probe's random access
                     // fetches the key and the answer from ONE 16-byte cell (separate arrays cost two HBM transactions)
    u64 * first_row; // [cap+1] global row id of the first (ANY: min or max) inserted row
    u32 * cnt;       // [cap+1] rows per key (ALL)
    u64 * start;     // [cap+1] CSR start (ALL)
    u64 * used_by;   // [cap+1] left-row sequence that consumed this cell (INNER ANY)
    u64 * rowids;    // [inserted] CSR payload (ALL)
                     // value = the row id itself (unique key / ANY), or MULTI | count | CSR start
    u64 capacity;
    JoinCtrl * ctrl;
    // Probe prefilter: a bitmap of pf_mask+1 bits (16 per build row, 64 Ki..32 Mi bits) small enough to live in L2, tested
    // before the hash table, which does not fit L2 for any but tiny build sides.  Dense keys (max key <= pf_mask: dimension
    // surrogate keys) index it directly -- exact; otherwise one multiplicative hash -- a k=1 Bloom filter (~6 % false
    // positives).  It never produces a false negative, so results are unchanged; misses stop at an L2 hit.
    u32 * pf;
    u64 pf_mask;
};

struct PfView
{
    const u32 * words;
    u64 mask;
    bool dense;
};
__device__ __forceinline__ PfView jt_pf_view(const JoinTable & t)
{
    PfView v;
    v.words = t.pf;
    v.mask = t.pf_mask;
    v.dense = t.ctrl->max_key <= t.pf_mask;
    return v;
}
__device__ __forceinline__ u64 jt_pf_pos(const PfView & v, u64 key) { return v.dense ? key : ((key * 0x9E3779B97F4A7C15ull) >> 32) & v.mask; }
// false = certainly absent.  key != 0 (the zero key lives out of line and is answered by has_zero).
__device__ __forceinline__ bool jt_pf_maybe(const PfView & v, u64 key)
{
    // (build sides beyond 2 Mi rows get no prefilter: it would need more than L2 can hold -- C4's 1e7 keys with a 4 MB,
    //  27 %-false-positive filter probed 18 % slower than without)
    if (v.dense && key > v.mask)
        return false;
    const u64 pos = jt_pf_pos(v, key);
    return (v.words[pos >> 5] >> (pos & 31)) & 1;
}

struct BuildBlock
{
    u64 * keys = nullptr; // zero-extended keys on device (context pool)
    u8 * valid = nullptr; // NULL = all rows valid
    size_t keys_class = 0, valid_class = 0;
    u64 rows = 0;
    u64 base = 0;         // running position of this block's first row in slot_of_row
};

struct chgpu_join
{
    chgpu_ctx * ctx = nullptr;
    int key_type = CHGPU_U64;
    int kind = CHGPU_JOIN_INNER, strictness = CHGPU_STRICT_ALL, any_take_last_row = 0;
    std::vector<BuildBlock> blocks;
    u64 total_rows = 0;
    bool finished = false;     // the hash table exists (join_build_table)
    bool build_closed = false; // onBuildPhaseFinish was called: no more right blocks; the table itself is built by the first consumer that needs it
    JoinTable t{};
    void * table_mem = nullptr;
    size_t table_class = 0;
    u64 n_keys = 0;
    u64 inserted = 0;
    bool unique_keys = false; // no key has more than one build row (the CSR arrays are then untouched)
    u64 max_key = 0;          // largest non-zero build key / whether the zero key is present (read back once by finish_build)
    bool has_zero = false;
    u64 left_seq = 0; // running left-row sequence across joinBlock calls (INNER ANY)
    // RIGHT / FULL: JoinUsedFlags (src/Interpreters/HashJoin/JoinUsedFlags.h) -- one byte per build row in insertion order
    u8 * used = nullptr;
    size_t used_class = 0;
    u64 * block_base_dev = nullptr; // [n_blocks] first flat row of every build block
    size_t base_class = 0;
    // Key set only (join_build_keyset): the exact bitmap over [0, max_key] of a build side of <= 4-byte keys, built WITHOUT the hash table
    // for consumers that only ask "is the key present" (the SEMI / ANTI steps of chgpu_join_probe_chain).  Superseded by t.pf once the
    // table exists.
    u32 * ks_pf = nullptr;
    size_t ks_class = 0;
    u64 ks_bits = 0;
    bool ks_ready = false;
    // the same key set with the build ROW of every key beside it (dense surrogate keys, unique, one build block): dm_rows[key] = row or
    // 0xFFFFFFFF -- what a chain step that adds right columns needs instead of the hash table (join_chain.h, join_build_dense)
    u32 * dm_rows = nullptr;
    size_t dm_class = 0;
    bool dm_ready = false;
    unsigned long long * key_stats = nullptr; // device: {largest inserted key, zero key inserted}, kept up to date by k_join_stage_keys;
                                              // word [2]: the duplicate-key flag of a row-map build (k_join_dense_fill)
    size_t key_stats_class = 0;
    u64 stats_host[2] = {0, 0}; // the two words on the host, when a caller has fetched them for several joins at once (a chain)
    bool stats_known = false;
    bool dm_pending = false;    // the row map is filled but its duplicate flag has not been looked at yet
};

// the left-side behaviour of the four kinds: RIGHT probes like INNER, FULL like LEFT (JoinFeatures.h:20-40: add_missing for LEFT / FULL)
static inline int jf_left_kind(const chgpu_join * j) { return (j->kind == CHGPU_JOIN_LEFT || j->kind == CHGPU_JOIN_FULL) ? CHGPU_JOIN_LEFT : CHGPU_JOIN_INNER; }
static inline bool jf_track_used(const chgpu_join * j) { return j->kind == CHGPU_JOIN_RIGHT || j->kind == CHGPU_JOIN_FULL; }

// RIGHT ANY / RIGHT SEMI: the first left row to find a key takes ALL the right rows of that key, so the left row is replicated (JoinFeatures.h:28)
static inline bool jf_right_once(const chgpu_join * j) { return j->kind == CHGPU_JOIN_RIGHT && (j->strictness == CHGPU_STRICT_ANY || j->strictness == CHGPU_STRICT_SEMI); }
static inline bool jf_need_replication(const chgpu_join * j) { return j->strictness == CHGPU_STRICT_ALL || jf_right_once(j); } // JoinFeatures.h:28
static inline bool jf_need_filter(const chgpu_join * j)
{
    return !jf_need_replication(j)
        && (jf_left_kind(j) == CHGPU_JOIN_INNER || j->strictness == CHGPU_STRICT_SEMI || j->strictness == CHGPU_STRICT_ANTI); // JoinFeatures.h:32
}
static inline bool jf_add_missing(const chgpu_join * j) { return jf_left_kind(j) == CHGPU_JOIN_LEFT && j->strictness != CHGPU_STRICT_SEMI; } // :35

// ---------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 jload_key(const void * keys, int type, u64 i)
{
    switch (type)
    {
        case CHGPU_U32: case CHGPU_I32: return ((const u32 *)keys)[i];
        case CHGPU_U16: case CHGPU_I16: return ((const u16 *)keys)[i];
        case CHGPU_U8: case CHGPU_I8: return ((const u8 *)keys)[i];
        default: return ((const u64 *)keys)[i];
    }
}

// stats[0] = the largest inserted key, stats[1] = 1 when the zero key is inserted: what the key-set / row-map builds of join_chain.h size
// themselves by, gathered while the keys pass through anyway.  One atomic per WORKGROUP at most, and only when it would change the
// word (thousands of waves on two addresses cost more than the pass itself).  4-byte keys whose column and staging buffer are 16-byte
// aligned move four rows per lane (the dimension tables of a star join: 30 M customer keys at the rate of a copy).
__global__ __launch_bounds__(JT) void k_join_stage_keys(const void * __restrict__ keys, int type, const u8 * __restrict__ null_map,
                                                        const u8 * __restrict__ join_mask, u64 n, u64 * __restrict__ out_keys, u8 * __restrict__ out_valid,
                                                        unsigned long long * __restrict__ stats)
{
    u64 m = 0;
    u32 zero = 0;
    auto row = [&](u64 i, u64 k) {
        const bool valid = !(null_map && null_map[i]) && !(join_mask && !join_mask[i]); // HashJoinMethodsImpl.h:261-272
        if (out_valid)
            out_valid[i] = valid;
        if (valid)
        {
            m = k > m ? k : m;
            zero |= k == 0 ? 1u : 0u;
        }
    };
    const bool wide = (type == CHGPU_U32 || type == CHGPU_I32) && (((uintptr_t)keys | (uintptr_t)out_keys) & 15) == 0;
    if (wide)
    {
        typedef u32 v4d __attribute__((ext_vector_type(4)));
        typedef u64 v2q __attribute__((ext_vector_type(2)));
        const u64 quads = n / 4;
        for (u64 q = (u64)blockIdx.x * JT + threadIdx.x; q < quads; q += (u64)gridDim.x * JT)
        {
            const v4d k4 = __builtin_nontemporal_load((const v4d *)keys + q);
            __builtin_nontemporal_store(v2q{k4.x, k4.y}, (v2q *)out_keys + 2 * q);
            __builtin_nontemporal_store(v2q{k4.z, k4.w}, (v2q *)out_keys + 2 * q + 1);
            row(4 * q, k4.x), row(4 * q + 1, k4.y), row(4 * q + 2, k4.z), row(4 * q + 3, k4.w);
        }
        for (u64 i = quads * 4 + (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
        {
            const u64 k = ((const u32 *)keys)[i];
            out_keys[i] = k;
            row(i, k);
        }
    }
    else
        for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
        {
            const u64 k = jload_key(keys, type, i);
            out_keys[i] = k;
            row(i, k);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
        const u64 x = __shfl_xor(m, o);
        m = x > m ? x : m;
        zero |= __shfl_xor(zero, o);
    }
    __shared__ unsigned long long s_m;
    __shared__ u32 s_zero;
    if (threadIdx.x == 0)
        s_m = 0, s_zero = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
    {
        if (m)
            atomicMax(&s_m, (unsigned long long)m);
        if (zero)
            atomicOr(&s_zero, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        if (s_m > __hip_atomic_load(stats, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(stats, s_m);
        if (s_zero && !__hip_atomic_load(stats + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicOr(stats + 1, 1ull);
    }
}

// `claimed` tells the caller a new cell was taken; callers count claims in registers and add them to ctrl->n_keys once
// per wave when the kernel ends (a per-lane atomicAdd on that single address serialised the whole build: 6.3 ms for 1e7 rows).
__device__ __forceinline__ u32 jt_emplace(const JoinTable & t, u64 key, bool & claimed)
{
    claimed = false;
    if (key == 0)
    {
        if (t.ctrl->has_zero == 0 && atomicExch(&t.ctrl->has_zero, 1u) == 0)
            claimed = true;
        return (u32)t.capacity;
    }
    const u64 mask = t.capacity - 1;
    u64 slot = dev_intHash64(key) & mask;
    for (u64 step = 0; step < t.capacity; ++step)
    {
        u64 k = t.kv[2 * slot];
        if (k == 0)
        {
            k = atomicCAS((unsigned long long *)&t.kv[2 * slot], 0ull, (unsigned long long)key);
            if (k == 0)
            {
                claimed = true;
                return (u32)slot;
            }
        }
        if (k == key)
            return (u32)slot;
        slot = (slot + 1) & mask;
    }
    return NO_SLOT; // unreachable: capacity >= 2 * rows
}

// PF: the table has a prefilter (kernels are instantiated both ways and the host picks: with the test compiled in but
// disabled at run time, the C4 probe -- whose 1e7-row build side has no prefilter -- ran 37 % slower)
template <bool PF>
__device__ __forceinline__ u32 jt_find(const JoinTable & t, const PfView & pf, u64 key)
{
    if (key == 0)
        return t.ctrl->has_zero ? (u32)t.capacity : NO_SLOT;
    if constexpr (PF)
        if (!jt_pf_maybe(pf, key))
            return NO_SLOT;
    const u64 mask = t.capacity - 1;
    u64 slot = dev_intHash64(key) & mask;
    for (u64 step = 0; step < t.capacity; ++step)
    {
        const u64 k = t.kv[2 * slot];
        if (k == key)
            return (u32)slot;
        if (k == 0)
            return NO_SLOT;
        slot = (slot + 1) & mask;
    }
    return NO_SLOT;
}

// build pass 1: claim cells, count rows per key, record the owning row.
// The row that CLAIMS a cell (wins the compare-and-swap on its key) stores its row id with a plain store into the cell's value word --
// the same 16 bytes it has just touched -- and issues no further atomic.  Only later rows of the same key pay atomics (rows-per-key
// count, min / max row id).  A primary-key build side (every row claims) thus costs ONE device-scope atomic per row instead of
// three: the pass is bound by the ~2e10/s rate of scattered memory-side atomics, not by bandwidth.  ctrl->pad is set when any row found
// its key already present: without duplicates the CSR passes below (scan, fill, root-first) are skipped altogether.
__global__ __launch_bounds__(JT) void k_join_insert(JoinTable t, const u64 * __restrict__ keys, const u8 * __restrict__ valid, u64 n,
                                                    u64 block_index, int maps_all, int take_last, u32 * __restrict__ slot_of_row)
{
    u32 my_claims = 0; // nobody reads n_keys before the kernel ends: count in registers, one atomic per wave at the end
    u32 my_dups = 0;
    u64 my_max = 0;
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
    {
        u32 slot = NO_SLOT;
        bool claimed = false;
        if (!valid || valid[i])
        {
            my_max = keys[i] > my_max ? keys[i] : my_max;
            slot = jt_emplace(t, keys[i], claimed);
            if (slot != NO_SLOT)
            {
                const u64 rowid = (block_index << 32) | i;
                if (claimed)
                    t.kv[2 * (u64)slot + 1] = rowid; // nobody else writes this word before k_join_merge_claims / finalize read it
                else
                {
                    ++my_dups;
                    if (maps_all)
                    {
                        atomicAdd(&t.cnt[slot], 1u);                                              // RowRefList::rows beyond the claimer's
                        atomicMin((unsigned long long *)&t.first_row[slot], (unsigned long long)rowid); // the root RowRef = first inserted
                    }
                    else if (take_last)
                        atomicMax((unsigned long long *)&t.first_row[slot], (unsigned long long)(rowid + 1)); // stored +1 so 0 == unset
                    else
                        atomicMin((unsigned long long *)&t.first_row[slot], (unsigned long long)rowid);       // insertOne: first row wins
                }
            }
        }
        slot_of_row[i] = slot;
        my_claims += claimed;
    }
    // (a same-address atomic per wave and iteration costs ~10 ns each: 1.5 ms of a 1e7-row build)
    u32 tot = my_claims, dups = my_dups;
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1)
    {
        tot += __shfl_xor(tot, dlt, 64);
        dups += __shfl_xor(dups, dlt, 64);
    }
    if ((threadIdx.x & 63) == 0 && tot)
        atomicAdd(&t.ctrl->n_keys, (unsigned long long)tot);
    if ((threadIdx.x & 63) == 0 && dups && t.ctrl->pad == 0)
        atomicOr(&t.ctrl->pad, 1u);
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1)
    {
        const u64 o = __shfl_xor(my_max, dlt, 64);
        my_max = o > my_max ? o : my_max;
    }
    if ((threadIdx.x & 63) == 0 && my_max)
        atomicMax(&t.ctrl->max_key, (unsigned long long)my_max);
}

// build pass 1b (only when some key has several rows): fold the claimers' row ids (value words) into first_row / cnt so the CSR passes
// see what three atomics per row used to leave there
__global__ __launch_bounds__(JT) void k_join_merge_claims(JoinTable t, int maps_all, int take_last)
{
    for (u64 s = (u64)blockIdx.x * JT + threadIdx.x; s <= t.capacity; s += (u64)gridDim.x * JT)
    {
        const bool occupied = s == t.capacity ? (t.ctrl->has_zero != 0) : (t.kv[2 * s] != 0);
        if (!occupied)
            continue;
        const u64 claim = t.kv[2 * s + 1];
        if (maps_all)
        {
            t.cnt[s] += 1;
            t.first_row[s] = claim < t.first_row[s] ? claim : t.first_row[s];
        }
        else if (take_last)
            t.first_row[s] = claim + 1 > t.first_row[s] ? claim + 1 : t.first_row[s];
        else
            t.first_row[s] = claim < t.first_row[s] ? claim : t.first_row[s];
    }
}

// build pass 3: CSR fill
__global__ __launch_bounds__(JT) void k_join_fill(JoinTable t, const u32 * __restrict__ slot_of_row, u64 n, u64 block_index, u32 * __restrict__ cursor)
{
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
    {
        const u32 slot = slot_of_row[i];
        if (slot == NO_SLOT)
            continue;
        const u32 k = atomicAdd(&cursor[slot], 1u);
        t.rowids[t.start[slot] + k] = (block_index << 32) | i;
    }
}

// build pass 4: put the first-inserted row at the head of each key's run (RowRefList iteration starts at the root)
__global__ __launch_bounds__(JT) void k_join_root_first(JoinTable t)
{
    for (u64 s = (u64)blockIdx.x * JT + threadIdx.x; s <= t.capacity; s += (u64)gridDim.x * JT)
    {
        const u32 c = t.cnt[s];
        if (c < 2)
            continue;
        u64 * run = t.rowids + t.start[s];
        const u64 root = t.first_row[s];
        for (u32 k = 0; k < c; ++k)
            if (run[k] == root)
            {
                run[k] = run[0];
                run[0] = root;
                break;
            }
    }
}

// build pass 5: one 8-byte word per cell that answers a probe without touching cnt/start/rowids for unique keys.
// unique != 0: no key had a second row -- the value words already hold the claimers' row ids; only the prefilter is filled.
__global__ __launch_bounds__(JT) void k_join_finalize_values(JoinTable t, int maps_all, int take_last, int unique)
{
    const PfView pf = jt_pf_view(t); // every insert kernel has finished: max_key is final
    for (u64 s = (u64)blockIdx.x * JT + threadIdx.x; s <= t.capacity; s += (u64)gridDim.x * JT)
    {
        const bool occupied = s == t.capacity ? (t.ctrl->has_zero != 0) : (t.kv[2 * s] != 0);
        u64 v = NO_ROW;
        if (occupied && s != t.capacity && t.pf)
        {
            const u64 pos = jt_pf_pos(pf, t.kv[2 * s]);
            atomicOr(&t.pf[pos >> 5], 1u << (pos & 31));
        }
        if (unique)
        {
            if (!occupied)
                t.kv[2 * s + 1] = NO_ROW;
            continue;
        }
        if (occupied)
        {
            if (!maps_all)
                v = take_last ? t.first_row[s] - 1 : t.first_row[s];
            else
            {
                const u32 c = t.cnt[s];
                if (c == 1)
                    v = t.rowids[t.start[s]];
                else
                    v = JV_MULTI | ((u64)(c < JV_CNT_SAT ? c : JV_CNT_SAT) << 40) | (t.start[s] & JV_START_MASK);
            }
        }
        t.kv[2 * s + 1] = v;
    }
}

// prefilter bits straight from the build keys (after a slice build: unique keys, no NULLs, nothing else to finalise -- a pass over
// the build rows instead of k_join_finalize_values' pass over every cell)
__global__ __launch_bounds__(JT) void k_join_pf_fill(JoinTable t, const u64 * __restrict__ keys, u64 n)
{
    const PfView pf = jt_pf_view(t);
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
    {
        const u64 key = keys[i];
        if (key)
        {
            const u64 pos = jt_pf_pos(pf, key);
            atomicOr(&t.pf[pos >> 5], 1u << (pos & 31));
        }
    }
}

// largest staged build key (NULL rows staged as 0), for the dense-prefilter decision of build sides beyond 2 Mi rows
__global__ __launch_bounds__(JT) void k_join_max_key(const u64 * __restrict__ keys, u64 n, unsigned long long * __restrict__ out)
{
    u64 m = 0;
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
        m = keys[i] > m ? keys[i] : m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
        const u64 x = __shfl_xor(m, o);
        m = x > m ? x : m;
    }
    if ((threadIdx.x & 63) == 0 && m)
        atomicMax(out, (unsigned long long)m);
}

__global__ __launch_bounds__(JT) void k_fill_u64(u64 * p, u64 n, u64 v)
{
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
        p[i] = v;
}

enum { PV_ALL_INNER, PV_ALL_LEFT, PV_ANY_LEFT, PV_SEMI_LEFT, PV_ANTI_LEFT, PV_ANY_INNER, PV_ONCE_RIGHT, PV_ANTI_RIGHT };

// probe pass 0 (INNER ANY only): every matching left row bids for its right cell with its sequence number
template <bool PF>
__global__ __launch_bounds__(JT) void k_join_probe_bid(JoinTable t, const void * __restrict__ keys, int key_type, const u8 * __restrict__ null_map,
                                                       u64 n, u64 seq_base, u32 * __restrict__ slot_of_left)
{
    PfView pf{};
    if constexpr (PF)
        pf = jt_pf_view(t);
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
    {
        u32 slot = NO_SLOT;
        if (!(null_map && null_map[i]))
            slot = jt_find<PF>(t, pf, jload_key(keys, key_type, i));
        if (slot != NO_SLOT)
            atomicMin((unsigned long long *)&t.used_by[slot], (unsigned long long)(seq_base + i));
        slot_of_left[i] = slot;
    }
}

// probe pass 1: per left row, how many right rows get appended (and the filter byte).  A hit costs one random read (the 16-byte cell holds
// the key and its packed value word); the value is kept per left row so the emit pass streams instead of re-probing.
// Four rows per lane, stage by stage, every load of a stage unconditional (a row that is not looked up reads word / cell 0): a lane has four
// key loads, then four prefilter words, then four home cells in flight.  One row per lane and a probe loop per row left the kernel
// parked on s_waitcnt (2.55 ms for 1e8 probes of a 1e6-key table that needs 0.5 ms of traffic); only a row whose home cell holds
// another key walks on, by itself.
typedef u64 jpc_v2 __attribute__((ext_vector_type(2)));
template <bool PF>
__global__ __launch_bounds__(JT) void k_join_probe_count(JoinTable t, int variant, const void * __restrict__ keys, int key_type,
                                                         const u8 * __restrict__ null_map, u64 n, u64 seq_base, int slots_known,
                                                         u32 * __restrict__ slot_of_left, u64 * __restrict__ val_of_left,
                                                         u32 * __restrict__ counts, u8 * __restrict__ filter)
{
    PfView pf{};
    if constexpr (PF)
        pf = jt_pf_view(t);
#ifndef JPC_R
#define JPC_R 4
#endif
    constexpr int R = JPC_R;
    const u64 stride = (u64)gridDim.x * JT;
    const u64 mask = t.capacity - 1;
    for (u64 i0 = (u64)blockIdx.x * JT + threadIdx.x; i0 < n; i0 += stride * R)
    {
        u64 key[R], val[R];
        u32 slot[R];
        bool in[R], look[R];
#pragma unroll
        for (int q = 0; q < R; ++q)
        {
            const u64 i = i0 + (u64)q * stride;
            in[q] = i < n;
            const u64 ic = in[q] ? i : n - 1;
            key[q] = 0;
            slot[q] = NO_SLOT;
            if (slots_known)
                slot[q] = slot_of_left[ic];
            else
                key[q] = jload_key(keys, key_type, ic);
            const bool is_null = null_map && null_map[ic]; // HashJoinMethodsImpl.h:451-452
            look[q] = in[q] && !slots_known && !is_null;
        }
        if (!slots_known)
        {
            bool zero[R];
            u64 home[R];
#pragma unroll
            for (int q = 0; q < R; ++q)
            {
                zero[q] = look[q] && key[q] == 0; // the zero key lives out of line (HashTable.h:874-898)
                look[q] = look[q] && key[q] != 0;
            }
            if constexpr (PF)
            {
                u32 pw[R];
                u64 pos[R];
#pragma unroll
                for (int q = 0; q < R; ++q)
                {
                    look[q] = look[q] && !(pf.dense && key[q] > pf.mask);
                    pos[q] = look[q] ? jt_pf_pos(pf, key[q]) : 0;
                    pw[q] = pf.words[pos[q] >> 5];
                }
#pragma unroll
                for (int q = 0; q < R; ++q)
                    look[q] = look[q] && ((pw[q] >> (pos[q] & 31)) & 1u);
            }
            jpc_v2 c[R], c2[R];
#pragma unroll
            for (int q = 0; q < R; ++q)
            {
                home[q] = dev_intHash64(key[q]) & mask;
                c[q] = *(const jpc_v2 *)(t.kv + 2 * (look[q] ? home[q] : 0));
                c2[q] = *(const jpc_v2 *)(t.kv + 2 * (look[q] ? (home[q] + 1) & mask : 0)); // the next cell too: nearly always the same 64-byte sector
            }
#pragma unroll
            for (int q = 0; q < R; ++q)
            {
                val[q] = NO_ROW;
                if (look[q])
                {
                    if (c[q].x == key[q])
                    {
                        slot[q] = (u32)home[q];
                        val[q] = c[q].y;
                    }
                    else if (c[q].x == 0)
                        ;
                    else if (c2[q].x == key[q])
                    {
                        slot[q] = (u32)((home[q] + 1) & mask);
                        val[q] = c2[q].y;
                    }
                    else if (c2[q].x != 0)
                    {
                        u64 sl = (home[q] + 2) & mask; // other keys in both cells: the rest of the walk
                        for (u64 step = 2; step < t.capacity; ++step)
                        {
                            const jpc_v2 cc = *(const jpc_v2 *)(t.kv + 2 * sl);
                            if (cc.x == key[q])
                            {
                                slot[q] = (u32)sl;
                                val[q] = cc.y;
                                break;
                            }
                            if (cc.x == 0)
                                break;
                            sl = (sl + 1) & mask;
                        }
                    }
                }
                else if (zero[q] && t.ctrl->has_zero)
                {
                    slot[q] = (u32)t.capacity;
                    val[q] = t.kv[2 * t.capacity + 1];
                }
            }
        }
        else
        {
#pragma unroll
            for (int q = 0; q < R; ++q)
            {
                const u64 v = t.kv[2 * (u64)(slot[q] != NO_SLOT ? slot[q] : 0) + 1];
                val[q] = slot[q] != NO_SLOT ? v : NO_ROW;
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q)
        {
            if (!in[q])
                continue;
            const u64 i = i0 + (u64)q * stride;
            const bool found = slot[q] != NO_SLOT;
            const u64 v = val[q];
            u32 rows_here = 0; // RowRefList::rows of the matched cell
            if (found)
            {
                if (!(v & JV_MULTI))
                    rows_here = 1;
                else
                {
                    const u64 cn = (v >> 40) & JV_CNT_SAT;
                    rows_here = cn < JV_CNT_SAT ? (u32)cn : t.cnt[slot[q]];
                }
            }
            u32 c = 0;
            u8 f = 0;
            switch (variant)
            {
                case PV_ALL_INNER: c = rows_here; break;
                case PV_ALL_LEFT: c = found ? rows_here : 1; break;                   // addNotFoundRow<add_missing>: ++current_offset
                case PV_ANY_LEFT: c = 1; break;                                       // found row or default row
                case PV_SEMI_LEFT: c = found ? 1 : 0; f = found; break;
                case PV_ANTI_LEFT: c = found ? 0 : 1; f = !found; break;              // :515-519, :535-536
                case PV_ANY_INNER: c = (found && t.used_by[slot[q]] == seq_base + i) ? 1 : 0; f = (u8)c; break; // setUsedOnce, :498-510
                case PV_ONCE_RIGHT: c = (found && t.used_by[slot[q]] == seq_base + i) ? rows_here : 0; break;     // RIGHT ANY / SEMI: setUsedOnce + addFoundRowAll, :487-497
                case PV_ANTI_RIGHT: c = 0; f = 0; break;                              // RIGHT ANTI: found rows only set the flags (:515-519), nothing is emitted
            }
            counts[i] = c;
            val_of_left[i] = v;
            if (filter)
                filter[i] = f;
        }
    }
}

// LEFT SEMI / LEFT ANTI when the right side contributes no columns (the caller passed right_rowid == NULL; the reference's
// AddedColumns is empty then): only the filter byte and the number of kept rows are produced -- 1 byte written per left row
// instead of 21 (counts, packed values, offsets) plus a scan and an emit pass.  Four rows per lane are in flight.
template <bool PF>
__global__ __launch_bounds__(JT) void k_join_probe_filter(JoinTable t, int anti, const void * __restrict__ keys, int key_type,
                                                          const u8 * __restrict__ null_map, u64 n, u8 * __restrict__ filter, JoinCtrl * __restrict__ ctrl)
{
    PfView pf{};
    if constexpr (PF)
        pf = jt_pf_view(t);
#ifndef JPF_R
#define JPF_R 4
#endif
    constexpr int R = JPF_R;
    u32 kept = 0;
    const u64 stride = (u64)gridDim.x * JT;
    for (u64 i0 = (u64)blockIdx.x * JT + threadIdx.x; i0 < n; i0 += stride * R)
    {
        u64 key[R];
        bool in[R], ok[R];
#pragma unroll
        for (int q = 0; q < R; ++q)
        {
            const u64 i = i0 + (u64)q * stride;
            in[q] = i < n;
            ok[q] = in[q] && !(null_map && null_map[i]); // HashJoinMethodsImpl.h:451-452
            key[q] = in[q] ? jload_key(keys, key_type, i) : 0;
        }
#pragma unroll
        for (int q = 0; q < R; ++q)
        {
            if (!in[q])
                continue;
            bool found;
            if (PF && pf.dense && key[q] != 0)
                found = ok[q] && jt_pf_maybe(pf, key[q]); // a dense prefilter is exact: membership needs no table access
            else
                found = ok[q] && jt_find<PF>(t, pf, key[q]) != NO_SLOT;
            const u8 f = anti ? !found : found;             // :515-519, :535-536
            filter[i0 + (u64)q * stride] = f;
            kept += f;
        }
    }
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1)
        kept += __shfl_xor(kept, dlt, 64);
    if ((threadIdx.x & 63) == 0 && kept)
        atomicAdd((unsigned long long *)&ctrl->n_out, (unsigned long long)kept);
}

// The same filter-only probe for DENSE 4-byte keys (dimension surrogate keys: the prefilter bitmap IS the key set) with the bitmap
// staged in LDS.  From L2 the look-ups run at ~1.6e11/s chip-wide -- 64 lanes = 64 separate L2 requests -- i.e. 4.8 ms for the
// 750 M lineorder rows of SSB against the 0.6 ms the keys and filter bytes take to stream; LDS serves the same random bit reads an
// order of magnitude faster.  A slice of JPL_SLICE_BITS bits (150 KiB) fits; a larger key domain takes one pass per slice, pass d > 0
// OR-ing into the filter bytes of the passes before it (the anti inversion and the count of kept rows belong to the last pass).
// Four rows per lane and load: 16 bytes of keys, 4 null-map bytes, 4 filter bytes.
static constexpr u32 JPL_SLICE_BITS = 150u * 1024u * 8u;
template <bool HAS_NULL>
__global__ __launch_bounds__(1024) void k_join_probe_filter_lds(const u32 * __restrict__ pf_words, u32 slice_lo, u32 slice_bits, int anti, int first_pass, int last_pass,
                                                                int has_zero, const u32 * __restrict__ keys, const u8 * __restrict__ null_map, u64 n,
                                                                u8 * __restrict__ filter, JoinCtrl * __restrict__ ctrl)
{
    extern __shared__ __attribute__((aligned(16))) u32 jpl_bits[];
    const u32 n_words = (slice_bits + 31) / 32;
    for (u32 w = threadIdx.x; w < n_words; w += 1024)
        jpl_bits[w] = pf_words[slice_lo / 32 + w]; // slice_lo is a multiple of 32
    __syncthreads();
    auto found_in_slice = [&](u32 k) -> u32 {
        const u32 rel = k - slice_lo;
        const bool in = k != 0 && rel < slice_bits;
        const u32 r = in ? rel : 0;
        const u32 bit = (jpl_bits[r >> 5] >> (r & 31)) & 1u;
        return (in ? bit : 0u) | ((first_pass && k == 0 && has_zero) ? 1u : 0u); // the zero key lives out of line (HashTable.h:874-898)
    };
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const u64 nq = n / 4; // whole groups of four rows; the last n % 4 rows are done at the end
    const u64 q_per_wg = (nq + gridDim.x - 1) / gridDim.x;
    const u64 q0 = (u64)blockIdx.x * q_per_wg, q1 = q0 + q_per_wg < nq ? q0 + q_per_wg : nq;
    u32 kept = 0;
    constexpr int U = 4;
    for (u64 qb = q0 + threadIdx.x; qb < q1; qb += (u64)U * 1024)
    {
        v4u kk[U];
        u32 nm[U], old[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const u64 q = qb + (u64)u * 1024;
            const u64 qc = q < q1 ? q : q1 - 1;
            kk[u] = __builtin_nontemporal_load((const v4u *)keys + qc);
            nm[u] = HAS_NULL ? __builtin_nontemporal_load((const u32 *)null_map + qc) : 0u;
            old[u] = first_pass ? 0u : ((const u32 *)filter)[qc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const u64 q = qb + (u64)u * 1024;
            u32 acc = old[u];
            const u32 k4[4] = {kk[u].x, kk[u].y, kk[u].z, kk[u].w};
#pragma unroll
            for (int b = 0; b < 4; ++b)
            {
                const bool ok = !HAS_NULL || ((nm[u] >> (8 * b)) & 0xffu) == 0; // HashJoinMethodsImpl.h:451-452
                acc |= (ok ? found_in_slice(k4[b]) : 0u) << (8 * b);
            }
            if (last_pass)
            {
                acc = anti ? acc ^ 0x01010101u : acc; // :515-519, :535-536
                kept += (u32)__popc(acc);
            }
            if (q < q1)
                ((u32 *)filter)[q] = acc;
            else if (last_pass)
                kept -= (u32)__popc(acc); // (a clamped duplicate of the last group)
        }
    }
    if (last_pass && blockIdx.x == 0 && threadIdx.x < (u32)(n & 3))
    {
        // the last n % 4 rows, against the whole bitmap in global memory
        const u64 i = nq * 4 + threadIdx.x;
        const u32 k = keys[i];
        const bool ok = !(HAS_NULL && null_map[i]);
        const bool found = ok && (k == 0 ? has_zero != 0 : (k < slice_lo + slice_bits && ((pf_words[k >> 5] >> (k & 31)) & 1u) != 0)); // (the last slice ends the bitmap)
        const u8 f = anti ? !found : found;
        filter[i] = f;
        kept += f;
    }
    if (last_pass)
    {
#pragma unroll
        for (int dlt = 32; dlt >= 1; dlt >>= 1)
            kept += __shfl_xor(kept, dlt, 64);
        if ((threadIdx.x & 63) == 0 && kept)
            atomicAdd((unsigned long long *)&ctrl->n_out, (unsigned long long)kept);
    }
}

// The same probe for a key set of SEVERAL slices in ONE sweep over the rows.  k_join_probe_filter_lds takes one pass over all rows per
// slice (keys read again from HBM each time, the filter bytes of the passes before read back and rewritten); here a workgroup takes a PART
// of 64 Ki rows through all slices before it moves on: the part's keys come from HBM once and from L2 / Infinity Cache for the other
// slices, the hit bits of a thread's 64 rows wait in two registers between slices (a row's key lies in exactly one slice), and the filter
// bytes are written once, after the last slice.  Between slices the workgroup reloads its 150 KiB of LDS from the bitmap (L2-resident);
// consecutive parts walk the slices in opposite directions, so the slice a part ends with is the one the next part starts with.
// JPM_QPT: quads (of four rows) per thread and part
template <bool HAS_NULL, u32 JPM_QPT = 16>
__global__ __launch_bounds__(1024) void k_join_probe_filter_lds_multi(const u32 * __restrict__ pf_words, u32 dense_bits, u32 n_slices, int anti, int has_zero,
                                                                      const u32 * __restrict__ keys, const u8 * __restrict__ null_map, u64 n,
                                                                      u8 * __restrict__ filter, JoinCtrl * __restrict__ ctrl)
{
    extern __shared__ __attribute__((aligned(16))) u32 jpl_bits[];
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    constexpr u32 JPM_PART_Q = 1024 * JPM_QPT; // quads per part
    static_assert(JPM_QPT % 4 == 0 && JPM_QPT <= 16, "a thread's hit bits live in one 64-bit register");
    const u64 nq = n / 4; // whole groups of four rows; the last n % 4 rows are done at the end
    const u64 n_parts = (nq + JPM_PART_Q - 1) / JPM_PART_Q;
    u32 kept = 0;
    u32 loaded = ~0u;
    bool up = true;
    for (u64 part = blockIdx.x; part < n_parts; part += gridDim.x, up = !up)
    {
        const u64 q0 = part * JPM_PART_Q;
        u64 hits = 0; // bit 4 * k + b: row b of this thread's k-th quad of the part
        for (u32 si = 0; si < n_slices; ++si)
        {
            const u32 sl = up ? si : n_slices - 1 - si;
            const u32 slice_lo = sl * JPL_SLICE_BITS;
            const u32 slice_bits = dense_bits - slice_lo < JPL_SLICE_BITS ? dense_bits - slice_lo : JPL_SLICE_BITS;
            if (loaded != sl)
            {
                __syncthreads(); // every wave has finished probing the slice that is about to be replaced
                const u32 n_words = (slice_bits + 31) / 32;
                const u32 * src = pf_words + slice_lo / 32;
                for (u32 w = threadIdx.x * 4; w < n_words; w += 1024 * 4) // (slices start on 16-byte boundaries: JPL_SLICE_BITS % 128 == 0)
                {
                    if (w + 4 <= n_words)
                        *(v4u *)(jpl_bits + w) = *(const v4u *)(src + w);
                    else
                        for (u32 x = w; x < n_words; ++x)
                            jpl_bits[x] = src[x];
                }
                __syncthreads();
                loaded = sl;
            }
            const bool last = si + 1 == n_slices;
            auto found_in_slice = [&](u32 k) -> u32 {
                const u32 rel = k - slice_lo;
                const bool in = k != 0 && rel < slice_bits;
                const u32 r = in ? rel : 0;
                const u32 bit = (jpl_bits[r >> 5] >> (r & 31)) & 1u;
                return in ? bit : 0u;
            };
            constexpr int U = 4;
#pragma unroll 1
            for (u32 kb = 0; kb < JPM_QPT; kb += U) // (unrolled, the sixteen loads are hoisted together and spill)
            {
                v4u kk[U];
                u32 nm[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                {
                    const u64 q = q0 + (u64)(kb + u) * 1024 + threadIdx.x;
                    const u64 qc = q < nq ? q : nq - 1;
                    kk[u] = *((const v4u *)keys + qc); // plain loads: the other slices of this part find the lines in L2 / Infinity Cache
                    nm[u] = (HAS_NULL && last) ? *((const u32 *)null_map + qc) : 0u;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                {
                    const u64 q = q0 + (u64)(kb + u) * 1024 + threadIdx.x;
                    const u32 k4[4] = {kk[u].x, kk[u].y, kk[u].z, kk[u].w};
                    u32 h = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        h |= found_in_slice(k4[b]) << b;
                    hits |= (u64)h << (4 * (kb + u));
                    if (last)
                    {
                        const u32 hb = (u32)(hits >> (4 * (kb + u))) & 15u;
                        u32 acc = 0;
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                        {
                            const bool ok = !HAS_NULL || ((nm[u] >> (8 * b)) & 0xffu) == 0; // HashJoinMethodsImpl.h:451-452
                            const u32 f = ok ? (((hb >> b) & 1u) | ((k4[b] == 0 && has_zero) ? 1u : 0u)) : 0u; // the zero key lives out of line (HashTable.h:874-898)
                            acc |= f << (8 * b);
                        }
                        acc = anti ? acc ^ 0x01010101u : acc; // :515-519, :535-536
                        if (q < nq)
                        {
                            kept += (u32)__popc(acc);
                            __builtin_nontemporal_store(acc, (u32 *)filter + q);
                        }
                    }
                }
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (u32)(n & 3))
    {
        // the last n % 4 rows, against the whole bitmap in global memory
        const u64 i = nq * 4 + threadIdx.x;
        const u32 k = keys[i];
        const bool ok = !(HAS_NULL && null_map[i]);
        const bool found = ok && (k == 0 ? has_zero != 0 : (k < dense_bits && ((pf_words[k >> 5] >> (k & 31)) & 1u) != 0));
        const u8 f = anti ? !found : found;
        filter[i] = f;
        kept += f;
    }
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1)
        kept += __shfl_xor(kept, dlt, 64);
    if ((threadIdx.x & 63) == 0 && kept)
        atomicAdd((unsigned long long *)&ctrl->n_out, (unsigned long long)kept);
}

// probe pass 2b: where does max_joined_block_rows cut?  offsets are inclusive cumulative counts.
__global__ void k_join_cut(const u64 * __restrict__ offsets, u64 n, u64 max_rows, JoinCtrl * __restrict__ ctrl)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    u64 consumed = n;
    if (max_rows != 0 && n > 0)
    {
        // the loop stops BEFORE row i when the offset after row i-1 is already >= max (HashJoinMethodsImpl.h:436-444):
        // consumed = 1 + (first index whose inclusive offset >= max), capped at n
        u64 lo = 0, hi = n; // first idx in [0,n) with offsets[idx] >= max_rows
        while (lo < hi)
        {
            const u64 mid = (lo + hi) / 2;
            if (offsets[mid] >= max_rows)
                hi = mid;
            else
                lo = mid + 1;
        }
        if (lo < n)
            consumed = lo + 1;
    }
    ctrl->consumed = consumed;
    ctrl->n_out = consumed ? offsets[consumed - 1] : 0;
}

// probe pass 3: write the appended right row ids (streams counts/offsets/values; touches rowids only for duplicate keys)
__global__ __launch_bounds__(JT) void k_join_emit(JoinTable t, int variant, const u64 * __restrict__ val_of_left,
                                                  const u32 * __restrict__ counts, const u64 * __restrict__ offsets, u64 consumed,
                                                  u64 * __restrict__ right_rowid)
{
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < consumed; i += (u64)gridDim.x * JT)
    {
        const u32 c = counts[i];
        if (c == 0)
            continue;
        const u64 base = offsets[i] - c;
        const u64 v = val_of_left[i];
        if (v == NO_ROW || variant == PV_ANTI_LEFT)
        {
            right_rowid[base] = NO_ROW; // default row (addNotFoundRow -> insertDefault)
            continue;
        }
        if (!(v & JV_MULTI))
        {
            right_rowid[base] = v;
            continue;
        }
        const u64 * run = t.rowids + (v & JV_START_MASK);
        for (u32 k = 0; k < c; ++k)
            right_rowid[base + k] = run[k];
    }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
// Table capacity for `rows` build rows: a power of two with load factor in (0.35, 0.7].  Denser than the reference's
// 0.5 cap on purpose: the table is immutable after the build and a 1e7-row build then needs 2^24 cells (134 MB of keys),
// which stays resident in the 256 MB Infinity Cache, where 2^25 cells would not.
__global__ __launch_bounds__(JT) void k_join_mark_used(const u64 * __restrict__ rowid, u64 n, const u64 * __restrict__ block_base, u64 n_blocks, u64 total_rows,
                                                       u8 * __restrict__ used)
{
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
    {
        const u64 r = rowid[i];
        const u64 b = r >> 32;
        if (r == NO_ROW || b >= n_blocks)
            continue; // the default row of a LEFT / FULL miss
        const u64 f = block_base[b] + (r & 0xFFFFFFFFull);
        if (f < total_rows)
            used[f] = 1; // plain store: every writer stores the same value
    }
}

// RIGHT ANTI: a probe emits nothing; the left row that won a key's bid (the first to find it, over all blocks) marks every right row of
// that key as used -- once per key, however many left rows find it (used_flags.setUsed(find_result), HashJoinMethodsImpl.h:515-519)
__global__ __launch_bounds__(JT) void k_join_mark_used_keys(JoinTable t, const u32 * __restrict__ slot_of_left, const u64 * __restrict__ val_of_left, u64 n, u64 seq_base,
                                                            const u64 * __restrict__ block_base, u64 n_blocks, u64 total_rows, u8 * __restrict__ used)
{
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
    {
        const u32 slot = slot_of_left[i];
        if (slot == NO_SLOT || t.used_by[slot] != seq_base + i)
            continue;
        const u64 v = val_of_left[i];
        auto mark = [&](u64 r) {
            const u64 b = r >> 32;
            if (b < n_blocks)
            {
                const u64 f = block_base[b] + (r & 0xFFFFFFFFull);
                if (f < total_rows)
                    used[f] = 1;
            }
        };
        if (!(v & JV_MULTI))
        {
            mark(v);
            continue;
        }
        const u64 c0 = (v >> 40) & JV_CNT_SAT;
        const u32 c = c0 < JV_CNT_SAT ? (u32)c0 : t.cnt[slot];
        const u64 * run = t.rowids + (v & JV_START_MASK);
        for (u32 k = 0; k < c; ++k)
            mark(run[k]);
    }
}

// NotJoinedHash (src/Interpreters/HashJoin/HashJoin.cpp:1280-1420): the build rows no probe row ever matched -- rows whose key was
// NULL or whose ON mask was 0 included (they were never inserted, so they are never used) -- as (block << 32 | row) ids
__global__ __launch_bounds__(JT) void k_join_unused_flags(const u8 * __restrict__ used, u64 n, u32 * __restrict__ flag)
{
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
        flag[i] = used[i] ? 0u : 1u;
}

__global__ __launch_bounds__(JT) void k_join_unused_emit(const u32 * __restrict__ flag, const u64 * __restrict__ pos, u64 n, const u64 * __restrict__ block_base,
                                                         u64 n_blocks, u64 * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
    {
        if (!flag[i])
            continue;
        u64 lo = 0, hi = n_blocks - 1; // largest block with base <= i
        while (lo < hi)
        {
            const u64 mid = (lo + hi + 1) >> 1;
            if (block_base[mid] <= i)
                lo = mid;
            else
                hi = mid - 1;
        }
        out[pos[i]] = (lo << 32) | (i - block_base[lo]);
    }
}

static u64 jpow2_ceil(u64 x)
{
    u64 p = 256;
    while (p < x)
        p <<= 1;
    return p;
}

extern "C" int chgpu_join_create(chgpu_ctx * ctx, int key_type, int kind, int strictness, int any_take_last_row,
                                 uint64_t size_hint, chgpu_join ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    (void)size_hint;
    CHGPU_REQUIRE(ctx && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(chgpu_type_is_int(key_type),
                  CHGPU_ERR_NOT_IMPLEMENTED, "join key type %d: CPU path", key_type);
    CHGPU_REQUIRE(kind >= CHGPU_JOIN_INNER && kind <= CHGPU_JOIN_FULL, CHGPU_ERR_NOT_IMPLEMENTED, "join kind %d: CPU path", kind);
    CHGPU_REQUIRE(kind != CHGPU_JOIN_FULL || strictness == CHGPU_STRICT_ALL, CHGPU_ERR_NOT_IMPLEMENTED,
                  "FULL joins are carried for strictness ALL only (FULL ANY is a TODO in the reference too, HashJoinMethodsImpl.h:511-514): CPU path");
    CHGPU_REQUIRE(strictness >= CHGPU_STRICT_ANY && strictness <= CHGPU_STRICT_ANTI, CHGPU_ERR_NOT_IMPLEMENTED, "join strictness %d: CPU path", strictness);
    CHGPU_REQUIRE(!((strictness == CHGPU_STRICT_SEMI || strictness == CHGPU_STRICT_ANTI) && kind != CHGPU_JOIN_LEFT && kind != CHGPU_JOIN_RIGHT), CHGPU_ERR_NOT_IMPLEMENTED,
                  "only SEMI / ANTI LEFT and RIGHT are valid (joinDispatch.h:52-64)");
    chgpu_join * j = new chgpu_join();
    j->ctx = ctx;
    j->key_type = key_type;
    j->kind = kind;
    j->strictness = strictness;
    j->any_take_last_row = any_take_last_row ? 1 : 0;
    chgpu_ctx_retain(ctx);
    *out = j;
    return CHGPU_OK;
}

extern "C" int chgpu_join_free(chgpu_join * j)
{
    ChgpuDeviceGuard _dev_guard(j ? j->ctx : nullptr);
    if (!j)
        return CHGPU_OK;
    for (auto & b : j->blocks)
    {
        chgpu_pool_free(j->ctx, b.keys, b.keys_class);
        chgpu_pool_free(j->ctx, b.valid, b.valid_class);
    }
    if (j->table_mem)
        chgpu_pool_free(j->ctx, j->table_mem, j->table_class);
    if (j->used)
        chgpu_pool_free(j->ctx, j->used, j->used_class);
    if (j->block_base_dev)
        chgpu_pool_free(j->ctx, j->block_base_dev, j->base_class);
    if (j->ks_pf)
        chgpu_pool_free(j->ctx, j->ks_pf, j->ks_class);
    if (j->dm_rows)
        chgpu_pool_free(j->ctx, j->dm_rows, j->dm_class);
    if (j->key_stats)
        chgpu_pool_free(j->ctx, j->key_stats, j->key_stats_class);
    chgpu_ctx * ctx = j->ctx;
    delete j;
    chgpu_ctx_release(ctx);
    return CHGPU_OK;
}

extern "C" int chgpu_join_add_block(chgpu_join * j, const chgpu_col * key_col, const chgpu_col * null_map, const chgpu_col * join_mask,
                                    uint32_t * block_index_out)
{
    ChgpuDeviceGuard _dev_guard(j ? j->ctx : nullptr);
    CHGPU_REQUIRE(j && key_col, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(!j->finished && !j->build_closed, CHGPU_ERR_LOGICAL, "addBlockToJoin after onBuildPhaseFinish");
    CHGPU_REQUIRE(key_col->type == j->key_type, CHGPU_ERR_BAD_ARGUMENTS, "key column has type %d, expected %d", key_col->type, j->key_type);
    CHGPU_REQUIRE(key_col->rows < 0xFFFFFFFFull, CHGPU_ERR_TOO_MANY_ROWS, "Too many rows in right table block for HashJoin: %llu", (unsigned long long)key_col->rows); // HashJoin.cpp:563-564
    CHGPU_REQUIRE(j->blocks.size() < 0x7FFFFFFFull, CHGPU_ERR_TOO_MANY_ROWS, "too many right blocks"); // bit 63 of a row id tags packed multi-row values
    if (null_map)
        CHGPU_REQUIRE(null_map->type == CHGPU_U8 && null_map->rows == key_col->rows, CHGPU_ERR_SIZES_MISMATCH, "null map size mismatch");
    if (join_mask)
        CHGPU_REQUIRE(join_mask->type == CHGPU_U8 && join_mask->rows == key_col->rows, CHGPU_ERR_SIZES_MISMATCH, "join mask size mismatch");
    chgpu_ctx * ctx = j->ctx;
    BuildBlock b;
    b.rows = key_col->rows;
    b.base = j->total_rows;
    if (b.rows)
    {
        // the right block stays alive for the join's lifetime (data->blocks, HashJoin.cpp:656-658): keep its keys in HBM
        CHGPU_TRY(chgpu_pool_alloc(ctx, b.rows * sizeof(u64), (void **)&b.keys, &b.keys_class));
        if (null_map || join_mask)
        {
            const int rc = chgpu_pool_alloc(ctx, b.rows, (void **)&b.valid, &b.valid_class);
            if (rc != CHGPU_OK)
            {
                chgpu_pool_free(ctx, b.keys, b.keys_class);
                return rc;
            }
        }
        if (!j->key_stats)
        {
            int rc = chgpu_pool_alloc(ctx, 256, (void **)&j->key_stats, &j->key_stats_class);
            if (rc == CHGPU_OK && hipMemsetAsync(j->key_stats, 0, 32, ctx->stream) != hipSuccess)
                rc = chgpu_set_error(CHGPU_ERR_DEVICE, "memset failed");
            if (rc != CHGPU_OK)
            {
                chgpu_pool_free(ctx, b.keys, b.keys_class);
                if (b.valid)
                    chgpu_pool_free(ctx, b.valid, b.valid_class);
                return rc;
            }
        }
        hipLaunchKernelGGL(k_join_stage_keys, dim3(chgpu_grid_for(ctx, b.rows, JT, 8)), dim3(JT), 0, ctx->stream, (const void *)key_col->data, key_col->type,
                           null_map ? (const u8 *)null_map->data : nullptr, join_mask ? (const u8 *)join_mask->data : nullptr, b.rows, b.keys, b.valid, j->key_stats);
        ctx->counters[6] += 1;
        // the caller may free its columns right after this call returns.  Columns that own pool memory go back to this context's pool, whose
        // reuse is ordered on this stream behind the staging kernel; anything else (wrapped caller memory, views) may be released or
        // overwritten by means this stream does not order: wait.
        auto pooled_here = [&](const chgpu_col * c) { return !c || (c->owns && c->ctx == ctx); };
        if (!pooled_here(key_col) || !pooled_here(null_map) || !pooled_here(join_mask))
            CHGPU_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (block_index_out)
        *block_index_out = (u32)j->blocks.size();
    j->blocks.push_back(b);
    j->total_rows += b.rows;
    ctx->counters[2] += b.rows;
    return CHGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// Unique-key build at streaming speed (the primary-key build sides of star joins).  The generic build claims one cell per row with a
// device-scope CAS -- ~2e10/s whatever the bandwidth: 0.66 ms for 1e7 rows.  Here the build rows are partitioned down to table slices
// of 4096 cells with the same two passes as the LDS-staged probe (k_rp_hist_wide + k_rp_scatter into 64 partitions, k_rp_tilesort_keys
// inside them, the row id travelling as the word), every slice is then built by ONE workgroup in LDS (LDS compare-and-swap, linear
// probing inside the slice) and written out as one contiguous 64 KiB piece.  A row whose chain runs past its slice's end goes to a
// short overflow list that a last kernel inserts the generic way.  A duplicate key raises a flag and the generic build runs instead.
// ---------------------------------------------------------------------------------------------
struct JoinSliceFn1
{
    u64 mask;
    u32 shift;
    __device__ __forceinline__ u32 operator()(u64 key) const { return (u32)((dev_intHash64(key) & mask) >> shift); }
};
struct JoinSliceFn2
{
    u64 mask;
    u32 shift2, lg_p2;
    __device__ __forceinline__ u32 operator()(u64 key, u64 first) const
    {
        const u32 r = (u32)((dev_intHash64(key) & mask) >> shift2), r0 = (u32)((dev_intHash64(first) & mask) >> shift2) >> lg_p2 << lg_p2;
        return r - r0;
    }
};
static constexpr u32 JBS_LG_CELLS = 12, JBS_CELLS = 1u << JBS_LG_CELLS, JBS_TILE = 8192, JBS_LG_P1 = 6, JBS_THREADS = 512, JBS_MAX_OVERFLOW = 1u << 20;

__global__ __launch_bounds__(256) void k_join_iota(u64 * __restrict__ out, u64 n)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
        out[i] = i; // row id of block 0: (0 << 32) | row
}

// flags[0] = a duplicate key was met, [1] = overflow entries, [2] = the overflow list was too short
__global__ __launch_bounds__(JBS_THREADS) void k_join_build_slices(JoinTable t, const u64 * __restrict__ keys2, const u64 * __restrict__ rids2, u64 n, const u64 * __restrict__ off1, u32 G,
                                                                   u32 lg_p2, const unsigned short * __restrict__ tile_index, u32 * __restrict__ unit_ctr, u32 * __restrict__ flags,
                                                                   u64 * __restrict__ ovf_keys, u64 * __restrict__ ovf_rids)
{
    typedef u64 bv2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char jbs_lds[];
    u64 * ck = (u64 *)jbs_lds;       // [JBS_CELLS] keys
    u64 * cv = ck + JBS_CELLS;       // [JBS_CELLS] row ids
    constexpr u32 P1 = 1u << JBS_LG_P1, NW = JBS_THREADS / 64;
    __shared__ u64 s_off[P1 + 1];
    __shared__ u32 sh_unit;
    const u32 P2 = 1u << lg_p2, R2 = P1 << lg_p2, PB = 2 * P2;
    for (u32 p = threadIdx.x; p <= P1; p += JBS_THREADS)
        s_off[p] = p < P1 ? off1[(u64)p * G] : n;
    const u32 lane = threadIdx.x & 63;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u64 mask = t.capacity - 1;
    u32 inserted = 0;
    u64 my_max = 0;
    bool dup = false;
    for (;;)
    {
        __syncthreads(); // the previous slice has been written out
        if (threadIdx.x == 0)
            sh_unit = atomicAdd(unit_ctr, 1u);
        for (u32 c = threadIdx.x; c < JBS_CELLS; c += JBS_THREADS)
            ck[c] = 0;
        __syncthreads();
        const u32 r2 = sh_unit;
        if (r2 >= R2)
            break; // (every workgroup reaches this exit)
        const u32 p1 = r2 >> lg_p2, p2 = r2 & (P2 - 1);
        const u64 rb = s_off[p1], re = s_off[p1 + 1];
        const u64 slice = (u64)r2 * JBS_CELLS;
        if (rb != re)
        {
            const u32 t_lo = (u32)(rb / JBS_TILE), t_hi = (u32)((re - 1) / JBS_TILE);
            for (u32 tile = t_lo + wave; tile <= t_hi; tile += NW)
            {
                const u64 row0 = (u64)tile * JBS_TILE;
                u32 lo = 0, hi = P1 - 1; // the first-level partition that owns the tile's first row
                while (lo < hi)
                {
                    const u32 mid = (lo + hi + 1) >> 1;
                    if (s_off[mid] <= row0)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                const u32 bucket = (p1 - lo) * P2 + p2;
                if (bucket >= PB)
                    continue; // (k_rp_tilesort_keys raised the stray flag)
                const u32 a = tile_index[(u64)tile * (PB + 1) + bucket], b = tile_index[(u64)tile * (PB + 1) + bucket + 1];
                for (u32 o = lane; o < b - a; o += 64)
                {
                    const u64 key = keys2[row0 + a + o], rid = rids2[row0 + a + o];
                    my_max = key > my_max ? key : my_max;
                    if (key == 0)
                    {
                        // the zero key lives out of line (cell `capacity`)
                        if (atomicExch(&t.ctrl->has_zero, 1u) == 0)
                        {
                            t.kv[2 * t.capacity + 1] = rid;
                            ++inserted;
                        }
                        else
                            dup = true;
                        continue;
                    }
                    u32 c = (u32)((dev_intHash64(key) & mask) - slice);
                    for (;;)
                    {
                        if (c >= JBS_CELLS)
                        {
                            // the chain leaves the slice: the row goes to the overflow list
                            const u32 at = atomicAdd(&flags[1], 1u);
                            if (at < JBS_MAX_OVERFLOW)
                            {
                                ovf_keys[at] = key;
                                ovf_rids[at] = rid;
                            }
                            else
                                flags[2] = 1;
                            break;
                        }
                        const u64 old = atomicCAS((unsigned long long *)&ck[c], 0ull, (unsigned long long)key);
                        if (old == 0)
                        {
                            cv[c] = rid; // nobody reads it before the barrier below
                            ++inserted;
                            break;
                        }
                        if (old == key)
                        {
                            dup = true;
                            break;
                        }
                        ++c;
                    }
                }
            }
        }
        __syncthreads();
        // the slice goes out as it lies: 4096 {key, row id} cells = 64 KiB, contiguous
        for (u32 c = threadIdx.x; c < JBS_CELLS; c += JBS_THREADS)
        {
            const u64 k = ck[c];
            __builtin_nontemporal_store(bv2{k, k ? cv[c] : 0ull}, (bv2 *)(t.kv + 2 * (slice + c)));
        }
    }
    u32 tot = inserted;
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1)
    {
        tot += __shfl_xor(tot, dlt, 64);
        const u64 o = __shfl_xor(my_max, dlt, 64);
        my_max = o > my_max ? o : my_max;
    }
    if (lane == 0 && tot)
        atomicAdd(&t.ctrl->n_keys, (unsigned long long)tot);
    if (lane == 0 && my_max)
        atomicMax(&t.ctrl->max_key, (unsigned long long)my_max);
    if (dup)
        flags[0] = 1;
}

// the overflow list through the generic claim (after every slice has been written)
__global__ __launch_bounds__(JT) void k_join_insert_pairs(JoinTable t, const u64 * __restrict__ keys, const u64 * __restrict__ rids, const u32 * __restrict__ flags)
{
    const u32 n = flags[1] < JBS_MAX_OVERFLOW ? flags[1] : JBS_MAX_OVERFLOW;
    u32 claims = 0;
    bool dup = false;
    for (u32 i = blockIdx.x * JT + threadIdx.x; i < n; i += gridDim.x * JT)
    {
        bool claimed = false;
        const u32 slot = jt_emplace(t, keys[i], claimed);
        if (claimed)
        {
            t.kv[2 * (u64)slot + 1] = rids[i];
            ++claims;
        }
        else
            dup = true;
    }
    if (claims)
        atomicAdd(&t.ctrl->n_keys, (unsigned long long)claims);
    if (dup)
        ((u32 *)flags)[0] = 1;
}

// -> CHGPU_OK: the table is built (unique keys); NOT_IMPLEMENTED: shape does not fit or a duplicate key exists (the caller runs the
// generic build over a freshly zeroed table)
static int join_build_slices(chgpu_join * j, JoinTable & t)
{
    chgpu_ctx * ctx = j->ctx;
    const bool off = chgpu_opt(ctx, "tune_join_no_slice_build", 0) != 0;
    const u64 n = j->total_rows, cap = t.capacity;
    u32 lg_cap = 0;
    while ((1ull << lg_cap) < cap)
        ++lg_cap;
    if (off || j->blocks.size() != 1 || j->blocks[0].valid || n < (1u << 20) || n + JBS_TILE + RP_SCATTER_SLACK >= (1ull << 32)
        || lg_cap < JBS_LG_CELLS + JBS_LG_P1 || lg_cap > JBS_LG_CELLS + JBS_LG_P1 + 7 || ((uintptr_t)j->blocks[0].keys % 16) != 0)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    const u32 lg_p2 = lg_cap - JBS_LG_CELLS - JBS_LG_P1, P1 = 1u << JBS_LG_P1, PB = 2u << lg_p2;
    const JoinSliceFn1 fn1{cap - 1, lg_cap - JBS_LG_P1};
    const JoinSliceFn2 fn2{cap - 1, JBS_LG_CELLS, lg_p2};
    const u32 G = (u32)ctx->num_cus;
    const u64 rows_per_wg = ((n + G - 1) / G + 63) / 64 * 64;
    const u64 rows_per_wg2 = ((n + G - 1) / G + JBS_TILE - 1) / JBS_TILE * JBS_TILE;
    const u64 n_tiles = (n + JBS_TILE - 1) / JBS_TILE, n_pad = n_tiles * JBS_TILE;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const u64 m = (u64)P1 * G;
    const size_t cnt_b = al(m * 4), off_b = al(m * 8 + 8), tmp_b = chgpu_scan_tmp_bytes(m), a1_b = al((n + RP_SCATTER_SLACK) * 8), a2_b = al(n_pad * 8 + 64),
                 ix_b = al(n_tiles * (PB + 1) * 2 + 16), ov_b = al((size_t)JBS_MAX_OVERFLOW * 8);
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, cnt_b + off_b + 256 + tmp_b + 3 * a1_b + 2 * a2_b + ix_b + 2 * ov_b + 256, &scratch));
    char * p = (char *)scratch;
    u32 * counts = (u32 *)p; p += cnt_b;
    u64 * offsets = (u64 *)p; p += off_b;
    u64 * total_dev = (u64 *)p; p += 256; // [0] scan total, [2] unit counter | stray flag, [3..4] flags[0..3]
    void * tmp = p; p += tmp_b;
    u64 * rid0 = (u64 *)p; p += a1_b;
    u64 * keys1 = (u64 *)p; p += a1_b;
    u64 * rid1 = (u64 *)p; p += a1_b;
    u64 * keys2 = (u64 *)p; p += a2_b;
    u64 * rid2 = (u64 *)p; p += a2_b;
    unsigned short * tidx = (unsigned short *)p; p += ix_b;
    u64 * ovf_keys = (u64 *)p; p += ov_b;
    u64 * ovf_rids = (u64 *)p;
    u32 * unit_ctr = (u32 *)(total_dev + 2), * stray = unit_ctr + 1, * flags = (u32 *)(total_dev + 3);
    CHGPU_HIP(hipMemsetAsync(total_dev, 0, 64, ctx->stream));
    const u64 * keys0 = j->blocks[0].keys;
    hipLaunchKernelGGL(k_join_iota, dim3(chgpu_grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, rid0, n);
    hipLaunchKernelGGL((k_rp_hist_wide<u64, JoinSliceFn1>), dim3(G), dim3(RP_THREADS), 0, ctx->stream, keys0, n, rows_per_wg, P1, counts, fn1);
    CHGPU_TRY(chgpu_scan_exclusive_u32_u64(ctx, counts, offsets, m, total_dev, tmp, tmp_b));
    {
        const size_t lds = rp_scatter_lds_bytes(8192, P1, 8, true);
        auto scat = k_rp_scatter<8192, u64, true, JoinSliceFn1>;
        CHGPU_HIP(hipFuncSetAttribute((const void *)scat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(scat, dim3(G), dim3(RP_THREADS), lds, ctx->stream, keys0, (const u64 *)rid0, n, rows_per_wg, P1, (const u64 *)offsets, keys1, rid1, fn1);
    }
    {
        const size_t lds = (size_t)JBS_TILE * 16 + (size_t)(PB + 1) * 8 + 64;
        auto sortk = k_rp_tilesort_keys<JBS_TILE, JoinSliceFn2, RP_THREADS, true>;
        CHGPU_HIP(hipFuncSetAttribute((const void *)sortk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(sortk, dim3(G), dim3(RP_THREADS), lds, ctx->stream, (const u64 *)keys1, n, rows_per_wg2, PB, keys2, tidx, fn2, stray, (const u64 *)rid1, rid2);
    }
    {
        const size_t lds = (size_t)JBS_CELLS * 16;
        CHGPU_HIP(hipFuncSetAttribute((const void *)k_join_build_slices, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_join_build_slices, dim3(2 * G), dim3(JBS_THREADS), lds, ctx->stream, t, (const u64 *)keys2, (const u64 *)rid2, n, (const u64 *)offsets, G, lg_p2,
                           (const unsigned short *)tidx, unit_ctr, flags, ovf_keys, ovf_rids);
    }
    hipLaunchKernelGGL(k_join_insert_pairs, dim3(64), dim3(JT), 0, ctx->stream, t, (const u64 *)ovf_keys, (const u64 *)ovf_rids, (const u32 *)flags);
    ctx->counters[6] += 7;
    CHGPU_HIP(hipGetLastError());
    u64 back[3];
    CHGPU_TRY(chgpu_read_back(ctx, total_dev + 2, back, 24));
    const u32 stray_v = (u32)(back[0] >> 32), dup_v = (u32)back[1], too_long = (u32)back[2];
    if (stray_v || dup_v || too_long)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    return CHGPU_OK;
}

static int join_build_table(chgpu_join * j);
static u64 join_capacity_for(const chgpu_ctx * ctx, u64 rows)
{
    const u32 cap_shift = (u32)chgpu_opt(ctx, "tune_join_cap_shift", 1);
    return jpow2_ceil(rows + rows * 3 / 7 + 1) << cap_shift;
}

// IJoin::onBuildPhaseFinish: the right side is complete.  The hash table is built by the first consumer that needs it (joinBlock, the
// key count, non-joined rows ...): the fused probe of a large unique-key build side never does -- it joins partition by partition
// (join_probe_agg_radix).  CHGPU_TUNE_JOIN_EAGER_BUILD=1 builds here, as before.
extern "C" int chgpu_join_finish_build(chgpu_join * j)
{
    ChgpuDeviceGuard _dev_guard(j ? j->ctx : nullptr);
    CHGPU_REQUIRE(j, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    const bool eager = chgpu_opt(j->ctx, "tune_join_eager_build", 0) != 0;
    j->build_closed = true;
    return eager ? join_build_table(j) : CHGPU_OK;
}

static int join_build_table(chgpu_join * j)
{
    if (j->finished)
        return CHGPU_OK;
    j->build_closed = true;
    chgpu_ctx * ctx = j->ctx;
    // joinDispatch.h:30-68: MapsAll for every ALL join and for every RIGHT join (RIGHT ANY / SEMI / ANTI keep all rows of a key); a cell
    // remembers the left row that consumed it for INNER ANY and for RIGHT ANY / SEMI / ANTI (setUsedOnce / the per-key flag)
    const bool maps_all = j->strictness == CHGPU_STRICT_ALL || j->kind == CHGPU_JOIN_RIGHT;
    const bool flagged = (j->kind == CHGPU_JOIN_INNER && j->strictness == CHGPU_STRICT_ANY) || (j->kind == CHGPU_JOIN_RIGHT && j->strictness != CHGPU_STRICT_ALL);
    // load factor in (0.175, 0.35]: a probe then resolves at its home cell nearly always (1.1 cells per hit, 1.3 per miss, against 1.75 / 3.6
    // at 0.6).  Measured at C4: build 1.00 -> 0.90 ms (fewer retried claims), probe 3.40 -> 2.26 ms region-partitioned, 4.83 -> 3.39 ms
    // one-pass.  The table is immutable after the build and 288 GB of HBM make the doubled footprint (512 MB of cells for 1e7 rows) cheap.
    const u64 cap = join_capacity_for(j->ctx, j->total_rows);
    CHGPU_REQUIRE(cap + 1 < 0xFFFFFFFFull, CHGPU_ERR_NOT_IMPLEMENTED, "build side of %llu rows exceeds the 32-bit cell index", (unsigned long long)j->total_rows);
    const u64 cells = cap + 1;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t off_keys = 256, off_first = off_keys + al(cells * 16), off_cnt = off_first + al(cells * 8);
    size_t off_start = off_cnt + (maps_all ? al(cells * 4) : 0);
    size_t off_used = off_start + (maps_all ? al(cells * 8) : 0);
    size_t off_rowids = off_used + (flagged ? al(cells * 8) : 0);
    u64 pf_bits = 1ull << 16;
    while (pf_bits < 16 * j->total_rows && pf_bits < (1ull << 25))
        pf_bits <<= 1;
    bool use_pf = pf_bits >= 16 * j->total_rows && !chgpu_opt(ctx, "tune_join_no_prefilter", 0);
    // A larger build side of narrow DENSE keys (a filtered dimension table joined on its surrogate key: SSB's customer, 6 M of the keys
    // 1..30 M) still gets the exact bitmap if max_key + 1 bits fit the 4 MiB limit: the misses of the probe then stop at a bitmap that
    // lives in L2 / Infinity Cache instead of costing one HBM sector each.
    if (!use_pf && chgpu_type_size(j->key_type) <= 4 && !chgpu_opt(ctx, "tune_join_no_prefilter", 0) && !chgpu_opt(ctx, "tune_join_no_dense_prefilter", 0))
    {
        void * scratch0 = nullptr;
        CHGPU_TRY(chgpu_scratch(ctx, 256, &scratch0));
        CHGPU_HIP(hipMemsetAsync(scratch0, 0, 8, ctx->stream));
        for (const BuildBlock & b : j->blocks)
            if (b.rows)
                hipLaunchKernelGGL(k_join_max_key, dim3(chgpu_grid_for(ctx, b.rows, JT, 4)), dim3(JT), 0, ctx->stream, (const u64 *)b.keys, b.rows, (unsigned long long *)scratch0);
        u64 mk = 0;
        CHGPU_TRY(chgpu_read_back(ctx, scratch0, &mk, sizeof(mk)));
        if (mk < (1ull << 25))
        {
            pf_bits = 1ull << 16;
            while (pf_bits <= mk)
                pf_bits <<= 1;
            use_pf = true;
        }
    }
    size_t off_pf = off_rowids + (maps_all ? al(j->total_rows * 8) : 0);
    size_t total_b = off_pf + (use_pf ? al(pf_bits / 8) : 0) + 256;
    void * m = nullptr;
    CHGPU_TRY(chgpu_pool_alloc(ctx, total_b, &m, &j->table_class)); // pooled: no hipMalloc/hipFree per join
    j->table_mem = m;
    JoinTable & t = j->t;
    t.ctrl = (JoinCtrl *)m;
    t.kv = (u64 *)((char *)m + off_keys);
    t.first_row = (u64 *)((char *)m + off_first);
    t.cnt = maps_all ? (u32 *)((char *)m + off_cnt) : nullptr;
    t.start = maps_all ? (u64 *)((char *)m + off_start) : nullptr;
    t.used_by = flagged ? (u64 *)((char *)m + off_used) : nullptr;
    t.rowids = maps_all ? (u64 *)((char *)m + off_rowids) : nullptr;
    t.capacity = cap;
    t.pf = use_pf ? (u32 *)((char *)m + off_pf) : nullptr;
    t.pf_mask = pf_bits - 1;
    if (use_pf)
        CHGPU_HIP(hipMemsetAsync(t.pf, 0, pf_bits / 8, ctx->stream));
    // unique keys, one right block: the table slices are built in LDS and written out whole (join_build_slices; a prefilter is filled by
    // k_join_finalize_values afterwards); only the
    // control block and the zero key's cell need clearing first.  Anything else -- or a duplicate key met on the way -- takes the
    // generic build over a zeroed table.
    CHGPU_HIP(hipMemsetAsync(m, 0, off_keys, ctx->stream));
    CHGPU_HIP(hipMemsetAsync(t.kv + 2 * cap, 0, 16, ctx->stream));
    const int fast = join_build_slices(j, t);
    if (fast != CHGPU_OK && fast != CHGPU_ERR_NOT_IMPLEMENTED)
        return fast;
    const bool sliced = fast == CHGPU_OK;
    if (!sliced)
        CHGPU_HIP(hipMemsetAsync(m, 0, off_first, ctx->stream)); // ctrl + {key, value} cells
    const bool take_last = !maps_all && j->any_take_last_row;
    // first_row: ~0 for atomicMin, 0 for atomicMax(+1)
    // (unique keys: first_row / cnt are only read for cells that hold several rows -- there are none)
    if (!sliced)
    {
        CHGPU_HIP(hipMemsetAsync(t.first_row, take_last ? 0x00 : 0xFF, cells * 8, ctx->stream));
        if (maps_all)
            CHGPU_HIP(hipMemsetAsync(t.cnt, 0, cells * 4, ctx->stream));
    }
    if (flagged)
        CHGPU_HIP(hipMemsetAsync(t.used_by, 0xFF, cells * 8, ctx->stream));

    // scratch: slot_of_row u32[total_rows] | cursor u32[cells] | total u64 | scan tmp
    const size_t sor_b = al(j->total_rows * 4 + 4), cur_b = al(cells * 4), tmp_b = chgpu_scan_tmp_bytes(cells);
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, sor_b + cur_b + 256 + tmp_b, &scratch));
    u32 * slot_of_row = (u32 *)scratch;
    u32 * cursor = (u32 *)((char *)scratch + sor_b);
    u64 * total_dev = (u64 *)((char *)scratch + sor_b + cur_b);
    void * tmp = (char *)scratch + sor_b + cur_b + 256;

    for (size_t bi = 0; bi < j->blocks.size() && !sliced; ++bi)
    {
        const BuildBlock & b = j->blocks[bi];
        if (!b.rows)
            continue;
        hipLaunchKernelGGL(k_join_insert, dim3(chgpu_grid_for(ctx, b.rows, JT, 8)), dim3(JT), 0, ctx->stream, t, (const u64 *)b.keys, (const u8 *)b.valid, b.rows,
                           (u64)bi, maps_all ? 1 : 0, take_last ? 1 : 0, slot_of_row + b.base);
        ctx->counters[6] += 1;
    }
    // first flat row of every right block: row ids (block << 32 | row) -> position in payload columns glued over all blocks.  Uploaded
    // here so that the read-back below -- which waits for the stream -- also covers this copy out of a host temporary.
    const u64 nb_blocks = j->blocks.size();
    std::vector<u64> bases(nb_blocks ? nb_blocks : 1, 0);
    for (u64 b = 0; b < nb_blocks; ++b)
        bases[b] = j->blocks[b].base;
    {
        void * um = nullptr, * bm = nullptr;
        if (jf_track_used(j))
        {
            CHGPU_TRY(chgpu_pool_alloc(ctx, j->total_rows + 64, &um, &j->used_class));
            j->used = (u8 *)um;
            CHGPU_HIP(hipMemsetAsync(j->used, 0, j->total_rows + 64, ctx->stream));
        }
        CHGPU_TRY(chgpu_pool_alloc(ctx, bases.size() * sizeof(u64), &bm, &j->base_class));
        j->block_base_dev = (u64 *)bm;
        CHGPU_HIP(hipMemcpyAsync(j->block_base_dev, bases.data(), bases.size() * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    }
    // did any key get a second row?  (one small read-back; the common primary-key build then skips four passes over the table)
    JoinCtrl after_insert;
    CHGPU_TRY(chgpu_read_back(ctx, t.ctrl, &after_insert, sizeof(after_insert)));
    const bool unique = after_insert.pad == 0;
    j->unique_keys = unique;
    if (!unique)
    {
        hipLaunchKernelGGL(k_join_merge_claims, dim3(chgpu_grid_for(ctx, cells, JT, 8)), dim3(JT), 0, ctx->stream, t, maps_all ? 1 : 0, take_last ? 1 : 0);
        ctx->counters[6] += 1;
    }
    if (maps_all && !unique)
    {
        CHGPU_TRY(chgpu_scan_exclusive_u32_u64(ctx, t.cnt, t.start, cells, total_dev, tmp, tmp_b));
        CHGPU_HIP(hipMemsetAsync(cursor, 0, cells * 4, ctx->stream));
        for (size_t bi = 0; bi < j->blocks.size(); ++bi)
        {
            const BuildBlock & b = j->blocks[bi];
            if (!b.rows)
                continue;
            hipLaunchKernelGGL(k_join_fill, dim3(chgpu_grid_for(ctx, b.rows, JT, 8)), dim3(JT), 0, ctx->stream, t, (const u32 *)(slot_of_row + b.base), b.rows, (u64)bi, cursor);
            ctx->counters[6] += 1;
        }
        hipLaunchKernelGGL(k_join_root_first, dim3(chgpu_grid_for(ctx, cells, JT, 8)), dim3(JT), 0, ctx->stream, t);
        ctx->counters[6] += 1;
        CHGPU_TRY(chgpu_read_back(ctx, total_dev, &j->inserted, sizeof(u64)));
    }
    else if (maps_all)
        j->inserted = after_insert.n_keys;
    // (without duplicates and without a prefilter there is nothing to finalise: the value word of an empty cell is never read)
    if (sliced && unique)
    {
        if (t.pf)
        {
            hipLaunchKernelGGL(k_join_pf_fill, dim3(chgpu_grid_for(ctx, j->blocks[0].rows, JT, 8)), dim3(JT), 0, ctx->stream, t, (const u64 *)j->blocks[0].keys, j->blocks[0].rows);
            ctx->counters[6] += 1;
        }
    }
    else if (!unique || t.pf)
    {
        hipLaunchKernelGGL(k_join_finalize_values, dim3(chgpu_grid_for(ctx, cells, JT, 8)), dim3(JT), 0, ctx->stream, t, maps_all ? 1 : 0, take_last ? 1 : 0, unique ? 1 : 0);
        ctx->counters[6] += 1;
    }
    CHGPU_HIP(hipGetLastError());
    // unique keys: nothing after the inserts touches the control block -- the read-back above already holds the final key count, the
    // largest key and the zero-key flag (one host synchronisation per build instead of three)
    JoinCtrl c = after_insert;
    if (!unique)
        CHGPU_TRY(chgpu_read_back(ctx, t.ctrl, &c, sizeof(c)));
    j->n_keys = c.n_keys;
    j->max_key = c.max_key;
    j->has_zero = c.has_zero != 0;
    j->finished = true;
    return CHGPU_OK;
}

/* IJoin::getNonJoinedBlocks (src/Interpreters/IJoin.h:133-134; NotJoinedHash, HashJoin.cpp:1280-1420) for RIGHT / FULL joins: after the
   last joinBlock, the build rows no left row matched, as (block << 32 | row) ids in insertion order; the caller gathers the right
   columns there and pads the left columns with defaults. */
extern "C" int chgpu_join_non_joined_rows(chgpu_join * j, chgpu_col ** right_rowid_u64, uint64_t * rows_out)
{
    ChgpuDeviceGuard _dev_guard(j ? j->ctx : nullptr);
    CHGPU_REQUIRE(j && right_rowid_u64 && rows_out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(jf_track_used(j), CHGPU_ERR_LOGICAL, "non-joined rows exist for RIGHT / FULL joins only");
    if (!j->finished)
        CHGPU_TRY(join_build_table(j));
    chgpu_ctx * ctx = j->ctx;
    const u64 n = j->strictness == CHGPU_STRICT_SEMI ? 0 : j->total_rows; // JoinCommon::hasNonJoinedBlocks (JoinUtils.cpp:634-638): none for SEMI
    chgpu_col * out = nullptr;
    u64 total = 0;
    if (n)
    {
        auto al = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t b_flag = al(n * 4), b_pos = al(n * 8), b_tmp = chgpu_scan_tmp_bytes(n);
        void * mem = nullptr;
        size_t mem_class = 0;
        CHGPU_TRY(chgpu_pool_alloc(ctx, b_flag + b_pos + 256 + b_tmp, &mem, &mem_class));
        u32 * flag = (u32 *)mem;
        u64 * pos = (u64 *)((char *)mem + b_flag);
        u64 * total_dev = (u64 *)((char *)mem + b_flag + b_pos);
        void * tmp = (char *)mem + b_flag + b_pos + 256;
        const u32 grid = chgpu_grid_for(ctx, n, JT, 8);
        hipLaunchKernelGGL(k_join_unused_flags, dim3(grid), dim3(JT), 0, ctx->stream, (const u8 *)j->used, n, flag);
        int rc = chgpu_scan_exclusive_u32_u64(ctx, flag, pos, n, total_dev, tmp, b_tmp);
        if (rc == CHGPU_OK)
            rc = chgpu_read_back(ctx, total_dev, &total, sizeof(total));
        if (rc == CHGPU_OK)
            rc = chgpu_col_new(ctx, CHGPU_U64, total, &out);
        if (rc == CHGPU_OK && total)
            hipLaunchKernelGGL(k_join_unused_emit, dim3(grid), dim3(JT), 0, ctx->stream, (const u32 *)flag, (const u64 *)pos, n, (const u64 *)j->block_base_dev,
                               (u64)j->blocks.size(), (u64 *)out->data);
        ctx->counters[6] += 2;
        chgpu_pool_free(ctx, mem, mem_class);
        if (rc != CHGPU_OK)
            return rc;
        CHGPU_HIP(hipGetLastError());
    }
    else
        CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, 0, &out));
    *right_rowid_u64 = out;
    *rows_out = total;
    return CHGPU_OK;
}

__global__ __launch_bounds__(JT) void k_join_flatten(const u64 * __restrict__ rowid, u64 n, const u64 * __restrict__ block_base, u64 n_blocks, u64 * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
    {
        const u64 r = rowid[i];
        const u64 b = r >> 32;
        out[i] = (r == NO_ROW || b >= n_blocks) ? NO_ROW : block_base[b] + (r & 0xFFFFFFFFull);
    }
}

extern "C" int chgpu_join_flatten_rowids(chgpu_join * j, const chgpu_col * rowids, chgpu_col ** flat)
{
    ChgpuDeviceGuard _dev_guard(j ? j->ctx : nullptr);
    CHGPU_REQUIRE(j && rowids && flat, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(chgpu_type_size(rowids->type) == 8, CHGPU_ERR_BAD_ARGUMENTS, "row ids must be a 64-bit column");
    chgpu_ctx * ctx = j->ctx;
    const u64 nb = j->blocks.size();
    std::vector<u64> bases(nb ? nb : 1, 0);
    for (u64 b = 0; b < nb; ++b)
        bases[b] = j->blocks[b].base;
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, bases.size() * sizeof(u64), &scratch));
    CHGPU_HIP(hipMemcpyAsync(scratch, bases.data(), bases.size() * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    CHGPU_HIP(hipStreamSynchronize(ctx->stream)); // `bases` is a host temporary
    chgpu_col * out = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, rowids->rows, &out));
    if (rowids->rows)
    {
        hipLaunchKernelGGL(k_join_flatten, dim3(chgpu_grid_for(ctx, rowids->rows, JT, 8)), dim3(JT), 0, ctx->stream, (const u64 *)rowids->data, rowids->rows,
                           (const u64 *)scratch, nb, (u64 *)out->data);
        ctx->counters[6] += 1;
        // the scratch holding block_base is reused by the next call on this context
        CHGPU_HIP(hipStreamSynchronize(ctx->stream));
    }
    *flat = out;
    return CHGPU_OK;
}

extern "C" int chgpu_join_total_rows(chgpu_join * j, uint64_t * rows, uint64_t * keys)
{
    ChgpuDeviceGuard _dev_guard(j ? j->ctx : nullptr);
    CHGPU_REQUIRE(j, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    if (rows)
        *rows = j->total_rows; // IJoin::getTotalRowCount
    if (keys)
    {
        if (!j->finished)
            CHGPU_TRY(join_build_table(j));
        *keys = j->n_keys;
    }
    return CHGPU_OK;
}

extern "C" int chgpu_join_probe(chgpu_join * j, const chgpu_col * key_col, const chgpu_col * null_map, uint64_t max_joined_block_rows,
                                chgpu_col ** filter_out, chgpu_col ** offsets_out, chgpu_col ** right_rowid_out, uint64_t * n_out,
                                uint64_t * n_left_consumed)
{
    ChgpuDeviceGuard _dev_guard(j ? j->ctx : nullptr);
    CHGPU_REQUIRE(j && key_col && n_out && n_left_consumed, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(key_col->type == j->key_type, CHGPU_ERR_BAD_ARGUMENTS, "left key column has type %d, expected %d", key_col->type, j->key_type);
    if (null_map)
        CHGPU_REQUIRE(null_map->type == CHGPU_U8 && null_map->rows == key_col->rows, CHGPU_ERR_SIZES_MISMATCH, "null map size mismatch");
    if (!j->finished)
        CHGPU_TRY(join_build_table(j));
    chgpu_ctx * ctx = j->ctx;
    const u64 n = key_col->rows;
    const bool need_filter = jf_need_filter(j), need_repl = jf_need_replication(j);
    CHGPU_REQUIRE(!need_filter || filter_out, CHGPU_ERR_BAD_ARGUMENTS, "this join variant produces a filter: filter_u8 must not be NULL");
    CHGPU_REQUIRE(!need_repl || offsets_out, CHGPU_ERR_BAD_ARGUMENTS, "this join variant produces offsets_to_replicate: offsets_u64 must not be NULL");
    CHGPU_REQUIRE(right_rowid_out || (need_filter && !need_repl && j->kind == CHGPU_JOIN_LEFT && (j->strictness == CHGPU_STRICT_SEMI || j->strictness == CHGPU_STRICT_ANTI)),
                  CHGPU_ERR_BAD_ARGUMENTS, "right_rowid_u64 may only be NULL for LEFT SEMI / LEFT ANTI (filter-only probe)");
    if (filter_out) *filter_out = nullptr;
    if (offsets_out) *offsets_out = nullptr;
    if (right_rowid_out) *right_rowid_out = nullptr;
    int variant;
    if (j->strictness == CHGPU_STRICT_ALL) variant = jf_left_kind(j) == CHGPU_JOIN_LEFT ? PV_ALL_LEFT : PV_ALL_INNER;
    else if (jf_right_once(j)) variant = PV_ONCE_RIGHT;
    else if (j->kind == CHGPU_JOIN_RIGHT) variant = PV_ANTI_RIGHT;
    else if (j->strictness == CHGPU_STRICT_SEMI) variant = PV_SEMI_LEFT;
    else if (j->strictness == CHGPU_STRICT_ANTI) variant = PV_ANTI_LEFT;
    else variant = jf_left_kind(j) == CHGPU_JOIN_LEFT ? PV_ANY_LEFT : PV_ANY_INNER;
    if (!need_repl)
        max_joined_block_rows = 0; // the early stop only exists for need_replication (HashJoinMethodsImpl.h:434-444)

    if (n == 0)
    {
        if (need_filter) CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U8, 0, filter_out));
        if (need_repl) CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, 0, offsets_out));
        if (right_rowid_out) CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, 0, right_rowid_out));
        *n_out = 0;
        *n_left_consumed = 0;
        return CHGPU_OK;
    }

    if (!right_rowid_out)
    {
        // filter-only probe (SEMI / ANTI without right columns)
        chgpu_col * fcol = nullptr;
        CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U8, n, &fcol));
        hipError_t e = hipMemsetAsync(&j->t.ctrl->n_out, 0, sizeof(u64), ctx->stream);
        // dense 4-byte keys whose key set fits a few LDS slices: k_join_probe_filter_lds (the tail of a bitmap beyond max_key is zero)
        const bool no_lds_filter = chgpu_opt(ctx, "tune_join_no_lds_filter", 0) != 0;
        const u64 dense_bits = (j->max_key + 32) / 32 * 32;
        if (e == hipSuccess && !no_lds_filter && j->t.pf && j->max_key <= j->t.pf_mask && chgpu_type_size(j->key_type) == 4 && dense_bits <= 4ull * JPL_SLICE_BITS
            && n >= (1u << 20) && (uintptr_t)key_col->data % 16 == 0 && (!null_map || (uintptr_t)null_map->data % 4 == 0))
        {
            const u32 passes = (u32)((dense_bits + JPL_SLICE_BITS - 1) / JPL_SLICE_BITS);
            const bool no_multi = chgpu_opt(ctx, "tune_join_no_lds_filter_multi", 0) != 0;
            if (passes > 1 && !no_multi)
            {
                // several slices: one sweep, every part of the rows through all slices (k_join_probe_filter_lds_multi)
                const u32 qpt = (u32)chgpu_opt(ctx, "tune_join_lds_filter_qpt", 16);
                auto kern = qpt == 8 ? (null_map ? k_join_probe_filter_lds_multi<true, 8> : k_join_probe_filter_lds_multi<false, 8>)
                          : qpt == 4 ? (null_map ? k_join_probe_filter_lds_multi<true, 4> : k_join_probe_filter_lds_multi<false, 4>)
                                     : (null_map ? k_join_probe_filter_lds_multi<true, 16> : k_join_probe_filter_lds_multi<false, 16>);
                e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(JPL_SLICE_BITS / 8));
                if (e == hipSuccess)
                {
                    hipLaunchKernelGGL(kern, dim3((u32)ctx->num_cus), dim3(1024), (size_t)(JPL_SLICE_BITS / 8), ctx->stream, (const u32 *)j->t.pf, (u32)dense_bits, passes,
                                       variant == PV_ANTI_LEFT ? 1 : 0, j->has_zero ? 1 : 0, (const u32 *)key_col->data, null_map ? (const u8 *)null_map->data : nullptr, n,
                                       (u8 *)fcol->data, j->t.ctrl);
                    ctx->counters[6] += 1;
                    e = hipGetLastError();
                }
            }
            else
            for (u32 d = 0; d < passes && e == hipSuccess; ++d)
            {
                const u32 lo = d * JPL_SLICE_BITS;
                const u32 bits = (u32)(dense_bits - lo < JPL_SLICE_BITS ? dense_bits - lo : JPL_SLICE_BITS);
                const size_t lds_b = (size_t)(bits + 31) / 32 * 4;
                auto kern = null_map ? k_join_probe_filter_lds<true> : k_join_probe_filter_lds<false>;
                e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(JPL_SLICE_BITS / 8));
                if (e == hipSuccess)
                {
                    hipLaunchKernelGGL(kern, dim3((u32)ctx->num_cus), dim3(1024), lds_b, ctx->stream, (const u32 *)j->t.pf, lo, bits, variant == PV_ANTI_LEFT ? 1 : 0, d == 0 ? 1 : 0,
                                       d + 1 == passes ? 1 : 0, j->has_zero ? 1 : 0, (const u32 *)key_col->data, null_map ? (const u8 *)null_map->data : nullptr, n,
                                       (u8 *)fcol->data, j->t.ctrl);
                    ctx->counters[6] += 1;
                    e = hipGetLastError();
                }
            }
        }
        else if (e == hipSuccess)
        {
            auto kern = j->t.pf ? k_join_probe_filter<true> : k_join_probe_filter<false>;
            hipLaunchKernelGGL(kern, dim3(chgpu_grid_for(ctx, (n + JPF_R - 1) / JPF_R, JT, 8)), dim3(JT), 0, ctx->stream, j->t, variant == PV_ANTI_LEFT ? 1 : 0,
                               (const void *)key_col->data, j->key_type, null_map ? (const u8 *)null_map->data : nullptr, n, (u8 *)fcol->data, j->t.ctrl);
            ctx->counters[6] += 1;
            e = hipGetLastError();
        }
        JoinCtrl c;
        int rc = e == hipSuccess ? chgpu_read_back(ctx, j->t.ctrl, &c, sizeof(c)) : chgpu_set_error(CHGPU_ERR_DEVICE, "join probe launch: %s", hipGetErrorString(e));
        if (rc != CHGPU_OK)
        {
            chgpu_col_free(fcol);
            return rc;
        }
        j->left_seq += n;
        *filter_out = fcol;
        *n_out = c.n_out;
        *n_left_consumed = n;
        ctx->counters[3] += n;
        ctx->counters[4] += c.n_out;
        return CHGPU_OK;
    }

    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t sl_b = al(n * 4), cnt_b = al(n * 4), val_b = al(n * 8), tmp_b = chgpu_scan_tmp_bytes(n);
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, sl_b + cnt_b + val_b + 256 + tmp_b, &scratch));
    u32 * slot_of_left = (u32 *)scratch;
    u32 * counts = (u32 *)((char *)scratch + sl_b);
    u64 * val_of_left = (u64 *)((char *)scratch + sl_b + cnt_b);
    u64 * total_dev = (u64 *)((char *)scratch + sl_b + cnt_b + val_b);
    void * tmp = (char *)scratch + sl_b + cnt_b + val_b + 256;

    chgpu_col * filter = nullptr;
    chgpu_col * offsets = nullptr;
    chgpu_col * rowid = nullptr;
    int rc = CHGPU_OK;
    auto fail = [&](int code) {
        chgpu_col_free(filter);
        chgpu_col_free(offsets);
        chgpu_col_free(rowid);
        return code;
    };
    if (need_filter && (rc = chgpu_col_new(ctx, CHGPU_U8, n, &filter)) != CHGPU_OK)
        return fail(rc);
    if ((rc = chgpu_col_new(ctx, CHGPU_U64, n, &offsets)) != CHGPU_OK)
        return fail(rc);

    const u32 grid = chgpu_grid_for(ctx, n, JT, 8);
    const void * kp = key_col->data;
    const u8 * nm = null_map ? (const u8 *)null_map->data : nullptr;
    const u64 seq_base = j->left_seq;
    const bool bids = variant == PV_ANY_INNER || variant == PV_ONCE_RIGHT || variant == PV_ANTI_RIGHT;
    if (bids)
    {
        hipLaunchKernelGGL(j->t.pf ? k_join_probe_bid<true> : k_join_probe_bid<false>, dim3(grid), dim3(JT), 0, ctx->stream, j->t, kp, j->key_type, nm, n, seq_base, slot_of_left);
        ctx->counters[6] += 1;
    }
    hipLaunchKernelGGL(j->t.pf ? k_join_probe_count<true> : k_join_probe_count<false>, dim3(grid), dim3(JT), 0, ctx->stream, j->t, variant, kp, j->key_type, nm, n, seq_base,
                       bids ? 1 : 0, slot_of_left, val_of_left, counts, filter ? (u8 *)filter->data : nullptr);
    ctx->counters[6] += 1;
    if (variant == PV_ANTI_RIGHT)
    {
        hipLaunchKernelGGL(k_join_mark_used_keys, dim3(grid), dim3(JT), 0, ctx->stream, j->t, (const u32 *)slot_of_left, (const u64 *)val_of_left, n, seq_base,
                           (const u64 *)j->block_base_dev, (u64)j->blocks.size(), j->total_rows, j->used);
        ctx->counters[6] += 1;
    }
    if ((rc = chgpu_scan_inclusive_u32_u64(ctx, counts, (u64 *)offsets->data, n, total_dev, tmp, tmp_b)) != CHGPU_OK)
        return fail(rc);
    hipLaunchKernelGGL(k_join_cut, dim3(1), dim3(64), 0, ctx->stream, (const u64 *)offsets->data, n, (u64)max_joined_block_rows, j->t.ctrl);
    ctx->counters[6] += 1;
    JoinCtrl c;
    if ((rc = chgpu_read_back(ctx, j->t.ctrl, &c, sizeof(c))) != CHGPU_OK)
        return fail(rc);
    if ((rc = chgpu_col_new(ctx, CHGPU_U64, c.n_out, &rowid)) != CHGPU_OK)
        return fail(rc);
    if (c.n_out)
    {
        hipLaunchKernelGGL(k_join_emit, dim3(grid), dim3(JT), 0, ctx->stream, j->t, variant, (const u64 *)val_of_left, (const u32 *)counts,
                           (const u64 *)offsets->data, c.consumed, (u64 *)rowid->data);
        ctx->counters[6] += 1;
    }
    if (c.n_out && jf_track_used(j))
    {
        // used_flags.setUsed for every emitted right row (HashJoinMethodsImpl.h addFoundRowAll with flag_per_row)
        hipLaunchKernelGGL(k_join_mark_used, dim3(chgpu_grid_for(ctx, c.n_out, JT, 8)), dim3(JT), 0, ctx->stream, (const u64 *)rowid->data, (u64)c.n_out,
                           (const u64 *)j->block_base_dev, (u64)j->blocks.size(), j->total_rows, j->used);
        ctx->counters[6] += 1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        chgpu_set_error(CHGPU_ERR_DEVICE, "join probe launch: %s", hipGetErrorString(e));
        return fail(CHGPU_ERR_DEVICE);
    }
    // (the scratch buffers -- slot_of_left, counts -- are reused by the next call on this context, which launches on the same stream and so
    //  runs after the kernels above; chgpu_scratch drains the stream itself before it ever frees a buffer: no host synchronisation here)
    j->left_seq += c.consumed; // rows not consumed are resubmitted by the caller and bid again
    // outputs are cut to the consumed prefix (offsets_to_replicate->resize(i), filter.resize(i): :439-441)
    if (filter)
        filter->rows = c.consumed;
    offsets->rows = c.consumed;
    if (need_filter)
        *filter_out = filter;
    if (need_repl)
        *offsets_out = offsets;
    else
        chgpu_col_free(offsets);
    *right_rowid_out = rowid;
    *n_out = c.n_out;
    *n_left_consumed = c.consumed;
    ctx->counters[3] += c.consumed;
    ctx->counters[4] += c.n_out;
    return CHGPU_OK;
}


// ---------------------------------------------------------------------------------------------
// Fused probe -> payload gather -> aggregate without key: joinBlock (HashJoinMethodsImpl.h:402-549) followed by
// AddedColumns' lazy gather (AddedColumns.cpp:39-131) and executeWithoutKeyImpl (Aggregator.cpp:1276-1321) in ONE pass over the
// left keys.  Nothing per left row is written: no counts, no packed values, no offsets_to_replicate, no row ids, no gathered column --
// the plan `SELECT count(), sum(right.v) FROM left JOIN right USING k` only ever needed two numbers.
// Every lane keeps R rows in flight (the table and payload reads are dependent random accesses); the {key, value} cell is fetched
// with one 16-byte load.
// ---------------------------------------------------------------------------------------------
typedef u64 jv2 __attribute__((ext_vector_type(2)));

template <bool PF>
__device__ __forceinline__ bool jt_find_value(const JoinTable & t, const PfView & pf, u64 key, u64 & value, u32 & slot_out)
{
    if (key == 0)
    {
        if (!t.ctrl->has_zero)
            return false;
        slot_out = (u32)t.capacity;
        value = t.kv[2 * t.capacity + 1];
        return true;
    }
    if constexpr (PF)
        if (!jt_pf_maybe(pf, key))
            return false;
    const u64 mask = t.capacity - 1;
    u64 slot = dev_intHash64(key) & mask;
    for (u64 step = 0; step < t.capacity; ++step)
    {
        const jv2 c = *(const jv2 *)(t.kv + 2 * slot);
        if (c.x == key)
        {
            value = c.y;
            slot_out = (u32)slot;
            return true;
        }
        if (c.x == 0)
            return false;
        slot = (slot + 1) & mask;
    }
    return false;
}

// payload value at flat row f, widened to the 8-byte sum operand (integers sign/zero-extended, floats as Float64 bits)
__device__ __forceinline__ u64 jload_payload(const void * p, int type, u64 f)
{
    switch (type)
    {
        case CHGPU_I64: case CHGPU_U64: case CHGPU_F64: return ((const u64 *)p)[f];
        case CHGPU_U32: return ((const u32 *)p)[f];
        case CHGPU_I32: return (u64)(i64)((const i32 *)p)[f];
        case CHGPU_U16: return ((const u16 *)p)[f];
        case CHGPU_I16: return (u64)(i64)((const i16 *)p)[f];
        case CHGPU_U8: return ((const u8 *)p)[f];
        case CHGPU_I8: return (u64)(i64)((const i8 *)p)[f];
        default: return (u64)__double_as_longlong((double)((const float *)p)[f]); // CHGPU_F32
    }
}

template <bool PF, bool FLOAT>
__global__ __launch_bounds__(JT) void k_join_probe_agg(JoinTable t, int variant, const void * __restrict__ keys, int key_type, const u8 * __restrict__ null_map, u64 n,
                                                       const void * __restrict__ payload, int payload_type, const u64 * __restrict__ block_base, u64 n_blocks,
                                                       u64 * __restrict__ partials /* [grid][2] */)
{
    PfView pf{};
    if constexpr (PF)
        pf = jt_pf_view(t);
    constexpr int R = 4;
    u64 cnt = 0, isum = 0;
    double fsum = 0.0;
    auto flat_of = [&](u64 rowid) -> u64 { return n_blocks == 1 ? (rowid & 0xFFFFFFFFull) : block_base[rowid >> 32] + (rowid & 0xFFFFFFFFull); };
    auto add_row = [&](u64 rowid) {
        if (!payload)
            return;
        const u64 b = jload_payload(payload, payload_type, flat_of(rowid));
        if constexpr (FLOAT)
            fsum += __longlong_as_double((long long)b);
        else
            isum += b;
    };
    const u64 stride = (u64)gridDim.x * JT;
    for (u64 i0 = (u64)blockIdx.x * JT + threadIdx.x; i0 < n; i0 += stride * R)
    {
        u64 key[R];
        bool ok[R];
#pragma unroll
        for (int q = 0; q < R; ++q)
        {
            const u64 i = i0 + (u64)q * stride;
            ok[q] = i < n;
            key[q] = ok[q] ? jload_key(keys, key_type, i) : 0;
            ok[q] = ok[q] && !(null_map && null_map[i]);
        }
        u64 val[R];
        u32 slot[R];
        bool found[R];
#pragma unroll
        for (int q = 0; q < R; ++q)
            found[q] = ok[q] && jt_find_value<PF>(t, pf, key[q], val[q], slot[q]);
#pragma unroll
        for (int q = 0; q < R; ++q)
        {
            if (i0 + (u64)q * stride >= n)
                continue;
            // a row with a NULL key finds nothing (HashJoinMethodsImpl.h:451-452) but is still a left row of LEFT / ANTI joins
            if (!found[q])
            {
                // addNotFoundRow<add_missing>: a default right row (payload 0) for LEFT ALL / LEFT ANY, the kept row of ANTI
                cnt += (variant == PV_ALL_LEFT || variant == PV_ANY_LEFT || variant == PV_ANTI_LEFT) ? 1 : 0;
                continue;
            }
            if (variant == PV_ANTI_LEFT)
                continue;
            const u64 v = val[q];
            if (!(v & JV_MULTI))
            {
                cnt += 1;
                add_row(v);
                continue;
            }
            const u64 c0 = (v >> 40) & JV_CNT_SAT;
            const u32 c = c0 < JV_CNT_SAT ? (u32)c0 : t.cnt[slot[q]];
            const u64 * run = t.rowids + (v & JV_START_MASK);
            cnt += c;
            for (u32 k = 0; k < c; ++k)
                add_row(run[k]);
        }
    }
    // fixed-order reduction: lane partials -> wave (shuffle tree) -> workgroup (LDS, wave order) -> one partial per workgroup
    __shared__ u64 sh_c[JT / 64], sh_s[JT / 64];
    cnt = wave_reduce_add_u64(cnt);
    u64 sbits;
    if constexpr (FLOAT)
        sbits = (u64)__double_as_longlong(wave_reduce_add_f64(fsum));
    else
        sbits = wave_reduce_add_u64(isum);
    if ((threadIdx.x & 63) == 0)
    {
        sh_c[threadIdx.x >> 6] = cnt;
        sh_s[threadIdx.x >> 6] = sbits;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        u64 c = 0, si = 0;
        double sf = 0.0;
        for (u32 w = 0; w < JT / 64; ++w)
        {
            c += sh_c[w];
            if constexpr (FLOAT)
                sf += __longlong_as_double((long long)sh_s[w]);
            else
                si += sh_s[w];
        }
        partials[2 * (u64)blockIdx.x] = c;
        partials[2 * (u64)blockIdx.x + 1] = FLOAT ? (u64)__double_as_longlong(sf) : si;
    }
}

template <bool FLOAT>
__global__ __launch_bounds__(64) void k_join_probe_agg_finish(const u64 * __restrict__ partials, u32 n_parts, u64 * __restrict__ out2)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    u64 c = 0, si = 0;
    double sf = 0.0;
    for (u32 p = 0; p < n_parts; ++p) // workgroup order: run-to-run reproducible Float64 sums
    {
        c += partials[2 * (u64)p];
        if constexpr (FLOAT)
            sf += __longlong_as_double((long long)partials[2 * (u64)p + 1]);
        else
            si += partials[2 * (u64)p + 1];
    }
    out2[0] = c;
    out2[1] = FLOAT ? (u64)__double_as_longlong(sf) : si;
}

// ---------------------------------------------------------------------------------------------
// The same fused probe for build sides far beyond a cache: the probe keys are first radix-partitioned by TABLE REGION (the top bits
// of their home slot) with the carried-tail partition of radix_partition.h, then every region's keys are probed by the workgroups of
// ONE XCD at a time, so the region's slice of the table (~1 MB) is fetched from HBM once and every further lookup is an L2 hit:
// the unpartitioned probe pays one random 64-byte HBM / Infinity-Cache transaction per key (3.8e10/s chip-wide over a 256 MB table)
// whatever the bandwidth.  Workgroup -> XCD placement is read from HW_REG_XCC_ID and only steers which queue a workgroup drains
// first; every workgroup ends up draining all eight queues, so the result never depends on it.
// ---------------------------------------------------------------------------------------------
struct JoinRegionFn
{
    u64 mask;
    u32 shift;
    __device__ __forceinline__ u32 operator()(u64 key) const { return (u32)((dev_intHash64(key) & mask) >> shift); }
};

static constexpr u32 JPR_CHUNK = 4096; // keys per work item
static constexpr u32 JPR_XCDS = 8;
static constexpr u32 JPR_MAX_REGIONS = 512;

// per XCD x: the chunks of regions x, x + 8, ... as one queue; qstart[x * (R/8 + 1) + i] = first chunk of its i-th region
__global__ void k_jp_queues(const u64 * __restrict__ offsets, u32 G, u32 R, u64 n, u32 * __restrict__ qstart, u32 * __restrict__ qctr)
{
    const u32 x = threadIdx.x;
    if (x >= JPR_XCDS)
        return;
    const u32 per = R / JPR_XCDS;
    u32 acc = 0;
    for (u32 i = 0; i < per; ++i)
    {
        const u32 r = x + i * JPR_XCDS;
        const u64 b = offsets[(u64)r * G], e = r + 1 < R ? offsets[(u64)(r + 1) * G] : n;
        qstart[x * (per + 1) + i] = acc;
        acc += (u32)((e - b + JPR_CHUNK - 1) / JPR_CHUNK);
    }
    qstart[x * (per + 1) + per] = acc;
    qctr[x] = 0;
}

// A primary-key build side probed for sum(payload): the payload of every cell's row is gathered ONCE into the cell's value word
// (a copy of the table: {key, widened payload}), so a hit needs no second random access -- and no access outside its table region:
// the row-id indirection into an 80 MB payload column cost 3.2 GB of line fetches per 1e8 probes and evicted the region's slice
// from the XCD's L2 while it was being probed (PMC FETCH_SIZE 9.6 GB per launch before, for 0.8 GB of keys and 0.5 GB of table).
__global__ __launch_bounds__(JT) void k_join_fuse_payload(JoinTable t, const void * __restrict__ payload, int payload_type, const u64 * __restrict__ block_base, u64 n_blocks,
                                                          u64 * __restrict__ kvp)
{
    for (u64 s = (u64)blockIdx.x * JT + threadIdx.x; s <= t.capacity; s += (u64)gridDim.x * JT)
    {
        const u64 k = t.kv[2 * s];
        const bool occupied = s == t.capacity ? (t.ctrl->has_zero != 0) : (k != 0);
        u64 v = 0;
        if (occupied)
        {
            const u64 rowid = t.kv[2 * s + 1];
            const u64 flat = n_blocks == 1 ? (rowid & 0xFFFFFFFFull) : block_base[rowid >> 32] + (rowid & 0xFFFFFFFFull);
            v = jload_payload(payload, payload_type, flat);
        }
        *(jv2 *)(kvp + 2 * s) = jv2{k, v};
    }
}

// FUSED: t.kv points at such a {key, payload} copy (unique build keys): a hit adds the cell's second word
template <bool PF, bool FUSED = false>
__global__ __launch_bounds__(JT) void k_join_probe_agg_regions(JoinTable t, int variant, const u64 * __restrict__ keys, u64 n, const u64 * __restrict__ offsets, u32 G, u32 R,
                                                               const u32 * __restrict__ qstart, u32 * __restrict__ qctr, const void * __restrict__ payload, int payload_type,
                                                               const u64 * __restrict__ block_base, u64 n_blocks, unsigned long long * __restrict__ result2)
{
    PfView pf{};
    if constexpr (PF)
        pf = jt_pf_view(t);
    __shared__ u64 sh_begin, sh_end;
    __shared__ u32 sh_more;
    __shared__ u64 sh_c[JT / 64], sh_s[JT / 64];
    const u32 per = R / JPR_XCDS;
    // the queue tables and the region boundaries live in LDS: the lane that fetches a work item then makes no dependent global read
    __shared__ u32 s_qs[JPR_XCDS * (JPR_MAX_REGIONS / JPR_XCDS + 1)];
    __shared__ u64 s_roff[JPR_MAX_REGIONS + 1];
    for (u32 i = threadIdx.x; i < JPR_XCDS * (per + 1); i += JT)
        s_qs[i] = qstart[i];
    for (u32 r = threadIdx.x; r <= R; r += JT)
        s_roff[r] = r < R ? offsets[(u64)r * G] : n;
    const u32 xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & (JPR_XCDS - 1); // HW_REG_XCC_ID[3:0]
    u64 cnt = 0, isum = 0;
    auto flat_of = [&](u64 rowid) -> u64 { return n_blocks == 1 ? (rowid & 0xFFFFFFFFull) : block_base[rowid >> 32] + (rowid & 0xFFFFFFFFull); };
    for (u32 dx = 0; dx < JPR_XCDS; ++dx)
    {
        const u32 x = (xcc + dx) & (JPR_XCDS - 1);
        const u32 * qs = s_qs + x * (per + 1);
        for (;;)
        {
            __syncthreads(); // the previous item's sh_* have been read by everyone
            if (threadIdx.x == 0)
            {
                const u32 c = atomicAdd(&qctr[x], 1u);
                u32 more = c < qs[per];
                if (more)
                {
                    u32 i = 0;
                    while (i + 1 < per && qs[i + 1] <= c)
                        ++i;
                    const u32 r = x + i * JPR_XCDS;
                    const u64 rb = s_roff[r], re = s_roff[r + 1];
                    const u64 b = rb + (u64)(c - qs[i]) * JPR_CHUNK;
                    sh_begin = b;
                    sh_end = b + JPR_CHUNK < re ? b + JPR_CHUNK : re;
                }
                sh_more = more;
            }
            __syncthreads();
            if (!sh_more)
                break; // every thread of the workgroup leaves this queue together; the counter only ever grows
            const u64 begin = sh_begin, end = sh_end;
            constexpr int RR = 8; // keys per lane and batch: their first cell reads (and then their payload reads) are all in flight together --
                                  // a lane that walks its keys one after the other has ONE random access outstanding
            static_assert(JPR_CHUNK == JT * RR * 2, "chunk = 256 threads x 2 batches x 8 keys");
            const u64 mask = t.capacity - 1;
#pragma unroll 1
            for (u32 batch = 0; batch < 2; ++batch)
            {
                u64 key[RR];
                jv2 cell[RR];
                u64 slot0[RR];
                bool in[RR];
#pragma unroll
                for (int q = 0; q < RR; ++q)
                {
                    const u64 i = begin + (u64)(batch * RR + q) * JT + threadIdx.x;
                    in[q] = i < end;
                    key[q] = in[q] ? __builtin_nontemporal_load(&keys[i]) : 0;
                }
#pragma unroll
                for (int q = 0; q < RR; ++q)
                {
                    slot0[q] = dev_intHash64(key[q]) & mask;
                    bool go = in[q] && key[q] != 0;
                    if constexpr (PF)
                        go = go && jt_pf_maybe(pf, key[q]);
                    cell[q] = go ? *(const jv2 *)(t.kv + 2 * slot0[q]) : jv2{0, 0};
                }
                u64 pay_row[RR];
                bool pay[RR];
#pragma unroll
                for (int q = 0; q < RR; ++q)
                {
                    pay[q] = false;
                    pay_row[q] = 0;
                    if (!in[q])
                        continue;
                    bool found = false;
                    u64 v = 0;
                    u32 sl = 0;
                    if (key[q] == 0)
                    {
                        found = t.ctrl->has_zero != 0;
                        sl = (u32)t.capacity;
                        v = found ? t.kv[2 * t.capacity + 1] : 0;
                    }
                    else if (cell[q].x == key[q])
                    {
                        found = true;
                        v = cell[q].y;
                        sl = (u32)slot0[q];
                    }
                    else if (cell[q].x != 0)
                    {
                        // the home cell belongs to another key: walk on (rare at the table's load factor; misses end at the first empty cell)
                        u64 s2 = (slot0[q] + 1) & mask;
                        for (u64 step = 1; step < t.capacity; ++step)
                        {
                            const jv2 c = *(const jv2 *)(t.kv + 2 * s2);
                            if (c.x == key[q])
                            {
                                found = true;
                                v = c.y;
                                sl = (u32)s2;
                                break;
                            }
                            if (c.x == 0)
                                break;
                            s2 = (s2 + 1) & mask;
                        }
                    }
                    if (!found)
                    {
                        cnt += (variant == PV_ALL_LEFT || variant == PV_ANY_LEFT || variant == PV_ANTI_LEFT) ? 1 : 0;
                        continue;
                    }
                    if (variant == PV_ANTI_LEFT)
                        continue;
                    if constexpr (FUSED)
                    {
                        cnt += 1;
                        isum += v; // the value word IS the payload
                        continue;
                    }
                    if (!(v & JV_MULTI))
                    {
                        cnt += 1;
                        pay[q] = payload != nullptr;
                        pay_row[q] = flat_of(v);
                        continue;
                    }
                    const u64 c0 = (v >> 40) & JV_CNT_SAT;
                    const u32 c = c0 < JV_CNT_SAT ? (u32)c0 : t.cnt[sl];
                    const u64 * run = t.rowids + (v & JV_START_MASK);
                    cnt += c;
                    if (payload)
                        for (u32 k = 0; k < c; ++k)
                            isum += jload_payload(payload, payload_type, flat_of(run[k]));
                }
                // the single-row matches' payload reads, issued back to back
                if constexpr (!FUSED)
#pragma unroll
                    for (int q = 0; q < RR; ++q)
                        isum += pay[q] ? jload_payload(payload, payload_type, pay_row[q]) : 0;
            }
        }
    }
    // integer count and wrap-around sum: the order of the additions does not matter
    cnt = wave_reduce_add_u64(cnt);
    isum = wave_reduce_add_u64(isum);
    if ((threadIdx.x & 63) == 0)
    {
        sh_c[threadIdx.x >> 6] = cnt;
        sh_s[threadIdx.x >> 6] = isum;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        u64 c = 0, si = 0;
        for (u32 w = 0; w < JT / 64; ++w)
        {
            c += sh_c[w];
            si += sh_s[w];
        }
        atomicAdd(&result2[0], (unsigned long long)c);
        atomicAdd(&result2[1], (unsigned long long)si);
    }
}

// -> CHGPU_OK with res[] filled, or CHGPU_ERR_NOT_IMPLEMENTED when this plan does not apply (the caller then runs the one-pass probe)
static int join_probe_agg_regions(chgpu_join * j, const chgpu_col * key_col, const chgpu_col * right_payload, int variant, u64 res[2])
{
    chgpu_ctx * ctx = j->ctx;
    const u64 n = key_col->rows, cap = j->t.capacity;
    const bool off = chgpu_opt(ctx, "tune_join_no_regions", 0) != 0;
    const u64 min_rows = chgpu_opt(ctx, "tune_join_region_min_rows", (4ull << 20));
    // worth it when the table is far larger than the XCDs' L2s together (32 MB) and there are enough keys to pay two extra passes
    if (off || chgpu_type_size(j->key_type) != 8 || n < min_rows || n + RP_SCATTER_SLACK >= (1ull << 32) || cap * 16 < (64ull << 20) || ((uintptr_t)key_col->data % 16) != 0)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    if (right_payload && chgpu_type_is_float(right_payload->type))
        return CHGPU_ERR_NOT_IMPLEMENTED; // a Float64 sum keeps the one-pass probe's fixed reduction order
    const u32 region_kib = (u32)chgpu_opt(ctx, "tune_join_region_kib", 1024);
    u32 lg_cap = 0;
    while ((1ull << lg_cap) < cap)
        ++lg_cap;
    u32 R = 8;
    while (R < JPR_MAX_REGIONS && (cap * 16) / R > (u64)region_kib * 1024)
        R <<= 1;
    u32 lg_r = 0;
    while ((1u << lg_r) < R)
        ++lg_r;
    const JoinRegionFn fn{cap - 1, lg_cap - lg_r};
    const u32 G = (u32)ctx->num_cus;
    u64 rows_per_wg = ((n + G - 1) / G + 63) / 64 * 64;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const u64 m = (u64)R * G;
    const size_t cnt_b = al(m * 4), off_b = al(m * 8 + 8), tmp_b = chgpu_scan_tmp_bytes(m), q_b = al((size_t)JPR_XCDS * (R / JPR_XCDS + 1) * 4 + JPR_XCDS * 4 + 64);
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, cnt_b + off_b + 256 + tmp_b + q_b + al((n + RP_SCATTER_SLACK) * 8) + 256, &scratch));
    u32 * counts = (u32 *)scratch;
    u64 * offsets = (u64 *)((char *)scratch + cnt_b);
    u64 * total_dev = (u64 *)((char *)scratch + cnt_b + off_b); // [0] scan total, [2..3] the result
    void * tmp = (char *)scratch + cnt_b + off_b + 256;
    u32 * qstart = (u32 *)((char *)tmp + tmp_b);
    u32 * qctr = qstart + JPR_XCDS * (R / JPR_XCDS + 1);
    u64 * pkeys = (u64 *)((char *)qstart + q_b);
    unsigned long long * result2 = (unsigned long long *)(total_dev + 2);
    CHGPU_HIP(hipMemsetAsync(total_dev, 0, 64, ctx->stream));
    hipLaunchKernelGGL((k_rp_hist_wide<u64, JoinRegionFn>), dim3(G), dim3(RP_THREADS), 0, ctx->stream, (const u64 *)key_col->data, n, rows_per_wg, R, counts, fn);
    CHGPU_TRY(chgpu_scan_exclusive_u32_u64(ctx, counts, offsets, m, total_dev, tmp, tmp_b));
    {
        // keys only, plain runs (radix_partition.h k_rp_scatter; carried tails measured slower and wrote 1.26 GB for 0.8 GB of keys)
        const size_t lds = rp_scatter_lds_bytes(12288, R, 8, false);
        auto scat = k_rp_scatter<12288, u64, false, JoinRegionFn>;
        CHGPU_HIP(hipFuncSetAttribute((const void *)scat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(scat, dim3(G), dim3(RP_THREADS), lds, ctx->stream, (const u64 *)key_col->data, (const u64 *)nullptr, n, rows_per_wg, R, (const u64 *)offsets, pkeys,
                           (u64 *)nullptr, fn);
    }
    hipLaunchKernelGGL(k_jp_queues, dim3(1), dim3(64), 0, ctx->stream, (const u64 *)offsets, G, R, n, qstart, qctr);
    const void * pp = right_payload ? right_payload->data : nullptr;
    const int pt = right_payload ? right_payload->type : CHGPU_U64;
    const u32 grid = (u32)ctx->num_cus * 4;
    const bool no_fuse = chgpu_opt(ctx, "tune_join_no_fused_payload", 0) != 0;
    if (j->unique_keys && right_payload && !j->t.pf && !no_fuse)
    {
        // {key, payload} cells: one 16-byte read answers a hit completely (k_join_fuse_payload); rebuilt per call -- the payload column
        // is the caller's, and nothing ties its contents to its address between two calls
        void * fm = nullptr;
        size_t fcls = 0;
        CHGPU_TRY(chgpu_pool_alloc(ctx, (size_t)(cap + 1) * 16, &fm, &fcls));
        hipLaunchKernelGGL(k_join_fuse_payload, dim3(chgpu_grid_for(ctx, cap + 1, JT, 8)), dim3(JT), 0, ctx->stream, j->t, pp, pt, (const u64 *)j->block_base_dev, (u64)j->blocks.size(),
                           (u64 *)fm);
        JoinTable ft = j->t;
        ft.kv = (u64 *)fm;
        hipLaunchKernelGGL((k_join_probe_agg_regions<false, true>), dim3(grid), dim3(JT), 0, ctx->stream, ft, variant, (const u64 *)pkeys, n, (const u64 *)offsets, G, R,
                           (const u32 *)qstart, qctr, pp, pt, (const u64 *)j->block_base_dev, (u64)j->blocks.size(), result2);
        ctx->counters[6] += 5;
        const hipError_t e = hipGetLastError();
        chgpu_pool_free(ctx, fm, fcls); // reuse is stream-ordered behind the probe
        CHGPU_REQUIRE(e == hipSuccess, CHGPU_ERR_DEVICE, "join probe launch: %s", hipGetErrorString(e));
        return chgpu_read_back(ctx, result2, res, 16);
    }
    if (j->t.pf)
        hipLaunchKernelGGL(k_join_probe_agg_regions<true>, dim3(grid), dim3(JT), 0, ctx->stream, j->t, variant, (const u64 *)pkeys, n, (const u64 *)offsets, G, R,
                           (const u32 *)qstart, qctr, pp, pt, (const u64 *)j->block_base_dev, (u64)j->blocks.size(), result2);
    else
        hipLaunchKernelGGL(k_join_probe_agg_regions<false>, dim3(grid), dim3(JT), 0, ctx->stream, j->t, variant, (const u64 *)pkeys, n, (const u64 *)offsets, G, R,
                           (const u32 *)qstart, qctr, pp, pt, (const u64 *)j->block_base_dev, (u64)j->blocks.size(), result2);
    ctx->counters[6] += 4;
    CHGPU_HIP(hipGetLastError());
    return chgpu_read_back(ctx, result2, res, 16);
}

// ---------------------------------------------------------------------------------------------
// LDS-staged probe (unique build keys, integer payload): the probe keys are partitioned TWICE -- 64 contiguous first-level partitions
// by the top bits of the home slot (k_rp_hist_wide + k_rp_scatter, runs of 256 keys), then tile-sorted inside them by the next bits
// (k_rp_tilesort_keys) -- down to table slices of 8192 cells, which one workgroup stages in LDS as {key, payload} (the payload gathered
// through the row id while staging: once per build row, not once per probe row) and then answers every key of that slice from LDS.
// A probe key costs two streaming passes and one LDS look-up instead of an L2 gather; the {key, payload} copy of the table
// (k_join_fuse_payload) is not needed.  A linear-probing chain that runs past the slice end finds the next slice's first JPL2_TAIL cells
// in LDS too, and the global table behind them.
// ---------------------------------------------------------------------------------------------
static constexpr u32 JPL2_LG_CELLS = 12, JPL2_CELLS = 1u << JPL2_LG_CELLS, JPL2_TAIL = 256, JPL2_TILE = 16384, JPL2_LG_P1 = 6, JPL2_THREADS = 512;
// (4096-cell slices and 512-thread workgroups: two workgroups per CU, so one's staging round trips overlap the other's probing --
//  8192 cells x 1024 threads, one per CU: 0.68 ms for C4's 1e8 keys)

struct JoinBucket2Fn
{
    u64 mask;
    u32 shift2; // home slot -> second-level region (slice) number
    u32 lg_p2;  // slices per first-level partition
    __device__ __forceinline__ u32 operator()(u64 key, u64 first) const
    {
        const u32 r = (u32)((dev_intHash64(key) & mask) >> shift2), r0 = (u32)((dev_intHash64(first) & mask) >> shift2) >> lg_p2 << lg_p2;
        return r - r0; // 0 .. 2 * P2 - 1 for the tile's own and the next first-level partition (keys sort by partition, so r >= r0)
    }
};

// The radix join has no table whose placement it must match, so it places keys by ONE 64-bit multiply (top lg_cap bits of key x odd
// constant) instead of intHash64's two multiplies and three shift-xors: the hash is evaluated once per key in each of the four passes, and
// a 64-bit multiply is four quarter-rate instructions.  (A key set that clusters under it overflows a slice's window: the stray flag
// sends the call to the table paths.)
__device__ __forceinline__ u64 join_radix_slot(u64 key, u32 lg_cap) { return (key * 0x9E3779B97F4A7C15ull) >> (64 - lg_cap); }
struct JoinRadixFn1
{
    u32 lg_cap, shift;
    __device__ __forceinline__ u32 operator()(u64 key) const { return (u32)(join_radix_slot(key, lg_cap) >> shift); }
};
struct JoinRadixFn2
{
    u32 lg_cap, shift2, lg_p2;
    __device__ __forceinline__ u32 operator()(u64 key, u64 first) const
    {
        const u32 r = (u32)(join_radix_slot(key, lg_cap) >> shift2), r0 = (u32)(join_radix_slot(first, lg_cap) >> shift2) >> lg_p2 << lg_p2;
        return r - r0;
    }
};

// FROM_ROWS: there is no table -- the slice's LDS cells are filled from the BUILD ROWS of the slice, which were partitioned the same way
// (bkeys2 / bwords2 = keys and payloads tile-sorted inside 64 partitions, boff1 / btidx their offsets and run index; t only carries the
// capacity).  A duplicate build key raises dup_flag (the caller then builds the table and takes the other paths).
template <bool FROM_ROWS>
__global__ __launch_bounds__(JPL2_THREADS) void k_join_probe_lds(JoinTable t, int variant, const u64 * __restrict__ keys2, u64 n, const u64 * __restrict__ off1, u32 G, u32 lg_p2,
                                                         const unsigned short * __restrict__ tile_index, const u64 * __restrict__ payload,
                                                         const u64 * __restrict__ block_base, u64 n_blocks, u32 * __restrict__ unit_ctr, u32 * __restrict__ stray_flag,
                                                         unsigned long long * __restrict__ result2, const u64 * __restrict__ bkeys2, const u64 * __restrict__ bwords2,
                                                         const u64 * __restrict__ boff1, const unsigned short * __restrict__ btidx, u64 nb, u32 * __restrict__ dup_flag)
{
    // The hot loop touches global memory only to stream the keys in: everything a key can meet -- its slice, the cells behind it, the
    // zero key's cell -- is staged in LDS first, and the one case that is not (a chain longer than the staged window) raises the
    // stray flag instead of reading the table (the host then discards this run).  A conditional global load inside the loop would
    // make every wait for the prefetched keys a full drain.
    extern __shared__ __attribute__((aligned(16))) unsigned char jpl2_lds[];
    jv2 * cells = (jv2 *)jpl2_lds; // [JPL2_CELLS + JPL2_TAIL] the slice and the cells behind it, then [1] the zero key {present, payload}
    constexpr u32 P1 = 1u << JPL2_LG_P1, WIN = JPL2_CELLS + JPL2_TAIL;
    __shared__ u64 s_off[P1 + 1];
    __shared__ u32 sh_unit;
    const u32 P2 = 1u << lg_p2, R2 = P1 << lg_p2, PB = 2 * P2;
    __shared__ u64 s_boff[P1 + 1];
    for (u32 p = threadIdx.x; p <= P1; p += JPL2_THREADS)
    {
        s_off[p] = p < P1 ? off1[(u64)p * G] : n;
        if constexpr (FROM_ROWS)
            s_boff[p] = p < P1 ? boff1[(u64)p * G] : nb;
    }
    const u32 lane = threadIdx.x & 63;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u64 mask = t.capacity - 1;
    u32 lg_cap = 0;
    while ((1ull << lg_cap) < t.capacity)
        ++lg_cap;
    auto slot_of = [&](u64 key) -> u64 { return FROM_ROWS ? join_radix_slot(key, lg_cap) : (dev_intHash64(key) & mask); };
    auto flat_of = [&](u64 rowid) -> u64 { return n_blocks == 1 ? (rowid & 0xFFFFFFFFull) : block_base[rowid >> 32] + (rowid & 0xFFFFFFFFull); };
    const int experiment = CHGPU_EXPERIMENT_VALUE(variant >> 8); // timing experiments only, -DCHGPU_EXPERIMENTS builds (CHGPU_EXPERIMENT_JOIN_LDS: 1 = no slice build, 2 = no look-ups)
    variant &= 0xff;
    const bool miss_counts = variant == PV_ALL_LEFT || variant == PV_ANY_LEFT || variant == PV_ANTI_LEFT;
    const bool anti = variant == PV_ANTI_LEFT;
    if constexpr (!FROM_ROWS)
        if (threadIdx.x == 0)
        {
            const bool hz = t.ctrl->has_zero != 0; // the zero key lives out of line (cell `capacity`)
            cells[WIN] = jv2{hz ? 1ull : 0ull, hz ? payload[flat_of(t.kv[2 * t.capacity + 1])] : 0ull};
            cells[WIN + 1] = jv2{0, 0};
        }
    u64 cnt = 0, isum = 0;
    bool stray = false, dup = false;
    for (;;)
    {
        __syncthreads(); // the previous unit's cells have been read by everyone
        if (threadIdx.x == 0)
            sh_unit = atomicAdd(unit_ctr, 1u);
        __syncthreads();
        const u32 r2 = sh_unit;
        if (r2 >= R2)
            break; // (every workgroup reaches this exit)
        const u32 p1 = r2 >> lg_p2, p2 = r2 & (P2 - 1);
        const u64 rb = s_off[p1], re = s_off[p1 + 1];
        if (rb == re)
            continue; // no probe key lands in this partition
        // stage the slice, {key, row id} -> {key, payload}: all the cell loads first, then all the payload loads (two round trips, not 2 x 9)
        const u64 slice = (u64)r2 * JPL2_CELLS;
        if constexpr (FROM_ROWS)
        {
            // build the slice's cells from its build rows: LDS compare-and-swap, linear probing inside the window (it is private to this
            // unit, so a chain may run on into the cells behind the slice); the zero key sits in the extra cell
            constexpr u32 NWB = JPL2_THREADS / 64;
            u64 * cw = (u64 *)cells;
            for (u32 c = threadIdx.x; c <= WIN + 1; c += JPL2_THREADS)
                cells[c] = jv2{0, 0};
            __syncthreads();
            const u64 bb = s_boff[p1], be = s_boff[p1 + 1];
            if (bb != be && experiment != 1)
            {
                const u32 bt_lo = (u32)(bb / JBS_TILE), bt_hi = (u32)((be - 1) / JBS_TILE);
                const u32 my_bt = bt_lo + wave <= bt_hi ? (bt_hi - bt_lo - wave) / NWB + 1 : 0;
                auto insert_row = [&](u64 key, u64 pay) {
                    u32 c = key == 0 ? WIN : (u32)(slot_of(key) - slice);
                    const u64 want = key == 0 ? 1ull : key; // the zero key's cell holds {present, payload}
                    for (;;)
                    {
                        const u64 old = atomicCAS((unsigned long long *)&cw[2 * c], 0ull, (unsigned long long)want);
                        if (old == 0)
                        {
                            cw[2 * c + 1] = pay; // nobody reads it before the barrier below
                            break;
                        }
                        if (old == want || key == 0)
                        {
                            dup = true;
                            break;
                        }
                        if (++c >= WIN)
                        {
                            stray = true; // the window is too short for this chain: not this plan
                            break;
                        }
                    }
                };
                // as on the probe side: lane k fetches the run of this wave's k-th build tile, then the rows of TBB tiles are loaded
                // together (a run is ~60 rows: tile by tile the wave sat out an index and a row round trip per tile -- half the kernel)
                for (u32 kb = 0; kb < my_bt; kb += 64)
                {
                    u32 st_l = 0, ln_l = 0;
                    if (kb + lane < my_bt)
                    {
                        const u32 tile = bt_lo + wave + (kb + lane) * NWB;
                        const u64 row0 = (u64)tile * JBS_TILE;
                        u32 lo = 0, hi = P1 - 1;
                        while (lo < hi)
                        {
                            const u32 mid = (lo + hi + 1) >> 1;
                            if (s_boff[mid] <= row0)
                                lo = mid;
                            else
                                hi = mid - 1;
                        }
                        const u32 bucket = (p1 - lo) * P2 + p2;
                        if (bucket < PB) // (else: stray flag raised by the tile sort)
                        {
                            const u32 ia = btidx[(u64)tile * (PB + 1) + bucket], ib = btidx[(u64)tile * (PB + 1) + bucket + 1];
                            st_l = ia;
                            ln_l = ib - ia;
                        }
                    }
                    const u32 nkb = my_bt - kb < 64 ? my_bt - kb : 64;
                    constexpr u32 TBB = 4;
                    for (u32 k0 = 0; k0 < nkb; k0 += TBB)
                    {
                        u64 bkey[TBB], bpay[TBB], brow[TBB];
                        u32 blen[TBB];
#pragma unroll
                        for (u32 q = 0; q < TBB; ++q)
                        {
                            const u32 kk = k0 + q < nkb ? k0 + q : nkb - 1;
                            const u32 st = (u32)__builtin_amdgcn_readlane((int)st_l, (int)kk);
                            blen[q] = k0 + q < nkb ? (u32)__builtin_amdgcn_readlane((int)ln_l, (int)kk) : 0;
                            brow[q] = (u64)(bt_lo + wave + (kb + kk) * NWB) * JBS_TILE + st;
                            const u64 at = brow[q] + (lane < blen[q] ? lane : 0);
                            bkey[q] = bkeys2[at];
                            bpay[q] = bwords2[at];
                        }
#pragma unroll
                        for (u32 q = 0; q < TBB; ++q)
                        {
                            if (lane < blen[q])
                                insert_row(bkey[q], bpay[q]);
                            for (u32 o = 64 + lane; o < blen[q]; o += 64) // a run longer than a wave (skew)
                                insert_row(bkeys2[brow[q] + o], bwords2[brow[q] + o]);
                        }
                    }
                }
            }
        }
        else
        {
            constexpr u32 NC = (WIN + JPL2_THREADS - 1) / JPL2_THREADS;
            jv2 cl[NC];
#pragma unroll
            for (u32 i = 0; i < NC; ++i)
            {
                const u32 c = threadIdx.x + i * JPL2_THREADS;
                cl[i] = *(const jv2 *)(t.kv + 2 * ((slice + (c < WIN ? c : 0)) & mask));
            }
#pragma unroll
            for (u32 i = 0; i < NC; ++i)
                cl[i].y = payload[cl[i].x != 0 ? flat_of(cl[i].y) : 0];
#pragma unroll
            for (u32 i = 0; i < NC; ++i)
            {
                const u32 c = threadIdx.x + i * JPL2_THREADS;
                if (c < WIN)
                    cells[c] = cl[i];
            }
        }
        __syncthreads();
        const u32 t_lo = (u32)(rb / JPL2_TILE), t_hi = (u32)((re - 1) / JPL2_TILE);
        // This wave's tiles: t_lo + wave, + NW, ...  Lane k fetches the run of tile k (one round trip for all of them instead of one
        // per tile); the keys of tile k + 1 are in flight while tile k is answered from LDS.
        constexpr u32 NW = JPL2_THREADS / 64;
        const u32 my_tiles = (t_lo + wave <= t_hi && experiment != 2) ? (t_hi - t_lo - wave) / NW + 1 : 0;
        // One look-up, straight-line for all but the longest chains: the home cell and its successor are read unconditionally and the
        // answer is selected; the zero key is the same look-up aimed at the extra cell (which holds {1, payload} when present); lanes
        // without a key carry valid = false.  (With a branch per case the loop spent ~100 scalar instructions per 64 look-ups on
        // exec-mask bookkeeping -- PMC: 1.9e8 SALU against 1.4e8 VALU instructions for 1e8 keys -- but removing them did not move the
        // kernel's 0.69 ms: 59 % of its wave cycles are waits)
        auto probe_one = [&](u64 key, bool valid) {
            const bool zk = key == 0;
            const u64 want = zk ? 1ull : key;
            u32 c = zk ? WIN : (u32)(slot_of(key) - slice); // the home slot: inside the slice by construction
            const jv2 c0 = cells[c], c1 = cells[c + 1];
            bool found = c0.x == want;
            u64 v = c0.y;
            const bool second = !found && c0.x != 0 && !zk;
            found = found || (second && c1.x == want);
            v = (second && c1.x == want) ? c1.y : v;
            bool more = valid && second && c1.x != want && c1.x != 0; // a chain of three or more cells: rare at this load factor
            if (__any(more))
            {
                if (more)
                {
                    c += 2;
                    jv2 cell = c < WIN ? cells[c] : jv2{0, 0};
                    while (cell.x != want && cell.x != 0 && c + 1 < WIN)
                        cell = cells[++c];
                    stray = stray || (cell.x != want && cell.x != 0); // the chain runs on past the staged window
                    found = cell.x == want;
                    v = cell.y;
                }
            }
            cnt += !valid ? 0 : found ? (anti ? 0 : 1) : (miss_counts ? 1 : 0);
            isum += (valid && found && !anti) ? v : 0;
        };
        for (u32 kb = 0; kb < my_tiles; kb += 64)
        {
            u32 st_l = 0, ln_l = 0;
            if (kb + lane < my_tiles)
            {
                const u32 tile = t_lo + wave + (kb + lane) * NW;
                const u64 row0 = (u64)tile * JPL2_TILE;
                u32 lo = 0, hi = P1 - 1; // the first-level partition that owns the tile's first row: the largest p with s_off[p] <= row0
                while (lo < hi)          // (empty partitions share their start with the next one: the largest such p is the owner)
                {
                    const u32 mid = (lo + hi + 1) >> 1;
                    if (s_off[mid] <= row0)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                const u32 bucket = (p1 - lo) * P2 + p2;
                if (bucket < PB) // (else: a tile spanning three partitions -- k_rp_tilesort_keys raised the stray flag, the host discards this run)
                {
                    const u32 ia = tile_index[(u64)tile * (PB + 1) + bucket], ib = tile_index[(u64)tile * (PB + 1) + bucket + 1];
                    st_l = ia;
                    ln_l = ib - ia;
                }
            }
            const u32 nk = my_tiles - kb < 64 ? my_tiles - kb : 64;
            constexpr u32 U = 2;
            auto run_of = [&](u32 k, u64 & row, u32 & len) {
                const u32 kk = k < nk ? k : nk - 1;
                const u32 st = (u32)__builtin_amdgcn_readlane((int)st_l, (int)kk);
                len = k < nk ? (u32)__builtin_amdgcn_readlane((int)ln_l, (int)kk) : 0;
                row = (u64)(t_lo + wave + (kb + kk) * NW) * JPL2_TILE + st;
            };
            auto load_keys = [&](u64 row, u32 len, u32 o0, u64 (&key)[U]) {
#pragma unroll
                for (u32 u = 0; u < U; ++u)
                {
                    const u32 o = o0 + u * 64 + lane;
                    key[u] = __builtin_nontemporal_load(&keys2[row + (o < len ? o : 0)]);
                }
            };
            auto probe_keys = [&](u32 len, u32 o0, const u64 (&key)[U]) {
#pragma unroll
                for (u32 u = 0; u < U; ++u)
                    probe_one(key[u], o0 + u * 64 + lane < len);
            };
            // batches of TB tiles: all their key loads are issued before the first look-up (a run is only ~128 keys = two loads per lane:
            // tile by tile, even double-buffered, the wave had two tiles' worth of loads in flight and waited out the latency each time)
            constexpr u32 TB = 8;
            for (u32 k0 = 0; k0 < nk; k0 += TB)
            {
                u64 key[TB][U], row[TB];
                u32 len[TB];
#pragma unroll
                for (u32 q = 0; q < TB; ++q)
                {
                    run_of(k0 + q, row[q], len[q]);
                    load_keys(row[q], len[q], 0, key[q]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (u32 q = 0; q < TB; ++q)
                {
                    probe_keys(len[q], 0, key[q]);
                    for (u32 o0 = 64 * U; o0 < len[q]; o0 += 64 * U) // a run longer than 128 keys (skew): the rest, unpipelined
                    {
                        u64 kx[U];
                        load_keys(row[q], len[q], o0, kx);
                        probe_keys(len[q], o0, kx);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1)
    {
        cnt += __shfl_xor(cnt, dlt, 64);
        isum += __shfl_xor(isum, dlt, 64);
    }
    if (lane == 0 && (cnt || isum))
    {
        atomicAdd(&result2[0], (unsigned long long)cnt);
        atomicAdd(&result2[1], (unsigned long long)isum);
    }
    if (stray)
        *stray_flag = 1;
    if (dup)
        *dup_flag = 1;
}

// -> CHGPU_OK with res[] filled, or NOT_IMPLEMENTED (shape does not fit / a tile straddled three partitions): the caller goes on with the region probe
static int join_probe_agg_lds(chgpu_join * j, const chgpu_col * key_col, const chgpu_col * right_payload, int variant, u64 res[2])
{
    chgpu_ctx * ctx = j->ctx;
    const u64 n = key_col->rows, cap = j->t.capacity;
    const bool off = chgpu_opt(ctx, "tune_join_no_lds_probe", 0) != 0;
    const u64 min_rows = chgpu_opt(ctx, "tune_join_lds_min_rows", (8ull << 20));
    u32 lg_cap = 0;
    while ((1ull << lg_cap) < cap)
        ++lg_cap;
    if (off || !j->unique_keys || j->t.pf || !right_payload || chgpu_type_is_float(right_payload->type) || chgpu_type_size(right_payload->type) != 8
        || chgpu_type_size(j->key_type) != 8 || n < min_rows
        || n + JPL2_TILE + RP_SCATTER_SLACK >= (1ull << 32) || lg_cap < JPL2_LG_CELLS + JPL2_LG_P1 || lg_cap > JPL2_LG_CELLS + JPL2_LG_P1 + 7 || ((uintptr_t)key_col->data % 16) != 0)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    const u32 lg_r2 = lg_cap - JPL2_LG_CELLS, lg_p2 = lg_r2 - JPL2_LG_P1, P1 = 1u << JPL2_LG_P1, PB = 2u << lg_p2;
    const JoinRegionFn fn1{cap - 1, lg_cap - JPL2_LG_P1};
    const JoinBucket2Fn fn2{cap - 1, JPL2_LG_CELLS, lg_p2};
    const u32 G = (u32)ctx->num_cus;
    const u64 rows_per_wg = ((n + G - 1) / G + 63) / 64 * 64;
    const u64 rows_per_wg2 = ((n + G - 1) / G + JPL2_TILE - 1) / JPL2_TILE * JPL2_TILE;
    const u64 n_tiles = (n + JPL2_TILE - 1) / JPL2_TILE, n_pad = n_tiles * JPL2_TILE;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const u64 m = (u64)P1 * G;
    const size_t cnt_b = al(m * 4), off_b = al(m * 8 + 8), tmp_b = chgpu_scan_tmp_bytes(m), k1_b = al((n + RP_SCATTER_SLACK) * 8), k2_b = al(n_pad * 8 + 64), // (+ slack: an empty run at the very end of the last tile is addressed one row past it)
                 ix_b = al(n_tiles * (PB + 1) * 2 + 16);
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, cnt_b + off_b + 256 + tmp_b + k1_b + k2_b + ix_b + 256, &scratch));
    u32 * counts = (u32 *)scratch;
    u64 * offsets = (u64 *)((char *)scratch + cnt_b);
    u64 * total_dev = (u64 *)((char *)scratch + cnt_b + off_b); // [0] scan total, [2..3] the result, [4] unit counter + stray flag
    void * tmp = (char *)scratch + cnt_b + off_b + 256;
    u64 * keys1 = (u64 *)((char *)tmp + tmp_b);
    u64 * keys2 = (u64 *)((char *)keys1 + k1_b);
    unsigned short * tidx = (unsigned short *)((char *)keys2 + k2_b);
    unsigned long long * result2 = (unsigned long long *)(total_dev + 2);
    u32 * unit_ctr = (u32 *)(total_dev + 4), * stray = unit_ctr + 1;
    CHGPU_HIP(hipMemsetAsync(total_dev, 0, 64, ctx->stream));
    hipLaunchKernelGGL((k_rp_hist_wide<u64, JoinRegionFn>), dim3(G), dim3(RP_THREADS), 0, ctx->stream, (const u64 *)key_col->data, n, rows_per_wg, P1, counts, fn1);
    CHGPU_TRY(chgpu_scan_exclusive_u32_u64(ctx, counts, offsets, m, total_dev, tmp, tmp_b));
    {
        const size_t lds = rp_scatter_lds_bytes(12288, P1, 8, false);
        auto scat = k_rp_scatter<12288, u64, false, JoinRegionFn>;
        CHGPU_HIP(hipFuncSetAttribute((const void *)scat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(scat, dim3(G), dim3(RP_THREADS), lds, ctx->stream, (const u64 *)key_col->data, (const u64 *)nullptr, n, rows_per_wg, P1, (const u64 *)offsets, keys1,
                           (u64 *)nullptr, fn1);
    }
    {
        const size_t lds = (size_t)JPL2_TILE * 8 + (size_t)(PB + 1) * 8 + 64;
        auto sortk = k_rp_tilesort_keys<JPL2_TILE, JoinBucket2Fn>;
        CHGPU_HIP(hipFuncSetAttribute((const void *)sortk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(sortk, dim3(G), dim3(RP_THREADS), lds, ctx->stream, (const u64 *)keys1, n, rows_per_wg2, PB, keys2, tidx, fn2, stray, (const u64 *)nullptr, (u64 *)nullptr);
    }
    {
        const size_t lds = (size_t)(JPL2_CELLS + JPL2_TAIL + 2) * 16;
        CHGPU_HIP(hipFuncSetAttribute((const void *)k_join_probe_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_join_probe_lds<false>, dim3(2 * G), dim3(JPL2_THREADS), lds, ctx->stream, j->t, variant, (const u64 *)keys2, n, (const u64 *)offsets, G, lg_p2,
                           (const unsigned short *)tidx, (const u64 *)right_payload->data, (const u64 *)j->block_base_dev, (u64)j->blocks.size(), unit_ctr, stray, result2,
                           (const u64 *)nullptr, (const u64 *)nullptr, (const u64 *)nullptr, (const unsigned short *)nullptr, (u64)0, stray);
    }
    ctx->counters[6] += 5;
    CHGPU_HIP(hipGetLastError());
    u64 back[3];
    CHGPU_TRY(chgpu_read_back(ctx, result2, back, 24));
    if ((back[2] >> 32) != 0) // the stray flag: some tile spanned three first-level partitions (tiny partitions): not this plan
        return CHGPU_ERR_NOT_IMPLEMENTED;
    res[0] = back[0];
    res[1] = back[1];
    return CHGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// The fused probe WITHOUT a hash table in HBM (a radix join): while the table has not been built yet -- the build side is one block of
// staged keys -- both sides are partitioned down to 4096-cell slices of the table that WOULD be built (build rows {key, payload}: 0.16 ms
// for 1e7 rows; probe keys as in join_probe_agg_lds) and k_join_probe_lds<FROM_ROWS> fills every slice's LDS cells from its build
// rows and answers its probe keys.  Neither the global table (0.62 ms to build) nor the per-probe payload gathers exist.  Duplicate
// build keys, an overflowing window or tiny partitions raise a flag: NOT_IMPLEMENTED, and the caller builds the table after all.
// ---------------------------------------------------------------------------------------------
static int join_probe_agg_radix(chgpu_join * j, const chgpu_col * key_col, const chgpu_col * right_payload, int variant, u64 res[2])
{
    chgpu_ctx * ctx = j->ctx;
    const bool off = chgpu_opt(ctx, "tune_join_no_radix", 0) != 0;
    const u64 min_rows = chgpu_opt(ctx, "tune_join_lds_min_rows", (8ull << 20));
    const u64 n = key_col->rows, nb = j->total_rows;
    if (off || j->finished || j->blocks.size() != 1 || j->blocks[0].valid || nb < (1u << 20) || !right_payload || chgpu_type_is_float(right_payload->type)
        || chgpu_type_size(right_payload->type) != 8 || chgpu_type_size(j->key_type) != 8 || n < min_rows || n + JPL2_TILE + RP_SCATTER_SLACK >= (1ull << 32)
        || nb + JBS_TILE + RP_SCATTER_SLACK >= (1ull << 32) || ((uintptr_t)key_col->data % 16) != 0 || ((uintptr_t)right_payload->data % 16) != 0
        || ((uintptr_t)j->blocks[0].keys % 16) != 0)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    const u64 cap = join_capacity_for(j->ctx, nb);
    u32 lg_cap = 0;
    while ((1ull << lg_cap) < cap)
        ++lg_cap;
    if (lg_cap < JPL2_LG_CELLS + JPL2_LG_P1 || lg_cap > JPL2_LG_CELLS + JPL2_LG_P1 + 7)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    static_assert(JPL2_LG_CELLS == JBS_LG_CELLS && JPL2_LG_P1 == JBS_LG_P1, "one slice geometry for both sides");
    const u32 lg_p2 = lg_cap - JPL2_LG_CELLS - JPL2_LG_P1, P1 = 1u << JPL2_LG_P1, PB = 2u << lg_p2;
    const JoinRadixFn1 fn1{lg_cap, lg_cap - JPL2_LG_P1};
    const JoinRadixFn2 fn2{lg_cap, JPL2_LG_CELLS, lg_p2};
    const u32 G = (u32)ctx->num_cus;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const u64 m = (u64)P1 * G;
    const u64 p_rpw = ((n + G - 1) / G + 63) / 64 * 64, p_rpw2 = ((n + G - 1) / G + JPL2_TILE - 1) / JPL2_TILE * JPL2_TILE;
    const u64 b_rpw = ((nb + G - 1) / G + 63) / 64 * 64, b_rpw2 = ((nb + G - 1) / G + JBS_TILE - 1) / JBS_TILE * JBS_TILE;
    const u64 p_tiles = (n + JPL2_TILE - 1) / JPL2_TILE, b_tiles = (nb + JBS_TILE - 1) / JBS_TILE;
    const size_t cnt_b = al(m * 4), off_b = al(m * 8 + 8), tmp_b = chgpu_scan_tmp_bytes(m);
    const size_t pk1_b = al((n + RP_SCATTER_SLACK) * 8), pk2_b = al(p_tiles * JPL2_TILE * 8 + 64), pix_b = al(p_tiles * (PB + 1) * 2 + 16);
    const size_t bk1_b = al((nb + RP_SCATTER_SLACK) * 8), bk2_b = al(b_tiles * JBS_TILE * 8 + 64), bix_b = al(b_tiles * (PB + 1) * 2 + 16);
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, 2 * (cnt_b + off_b) + 256 + tmp_b + pk1_b + pk2_b + pix_b + 2 * bk1_b + 2 * bk2_b + bix_b + 256, &scratch));
    char * p = (char *)scratch;
    u32 * p_counts = (u32 *)p; p += cnt_b;
    u64 * p_offsets = (u64 *)p; p += off_b;
    u32 * b_counts = (u32 *)p; p += cnt_b;
    u64 * b_offsets = (u64 *)p; p += off_b;
    u64 * total_dev = (u64 *)p; p += 256; // [0] scan total, [2..3] the result, [4] unit counter | stray flag, [5] dup flag
    void * tmp = p; p += tmp_b;
    u64 * pk1 = (u64 *)p; p += pk1_b;
    u64 * pk2 = (u64 *)p; p += pk2_b;
    unsigned short * pix = (unsigned short *)p; p += pix_b;
    u64 * bk1 = (u64 *)p; p += bk1_b;
    u64 * bw1 = (u64 *)p; p += bk1_b;
    u64 * bk2 = (u64 *)p; p += bk2_b;
    u64 * bw2 = (u64 *)p; p += bk2_b;
    unsigned short * bix = (unsigned short *)p;
    unsigned long long * result2 = (unsigned long long *)(total_dev + 2);
    u32 * unit_ctr = (u32 *)(total_dev + 4), * stray = unit_ctr + 1, * dupf = (u32 *)(total_dev + 5);
    CHGPU_HIP(hipMemsetAsync(total_dev, 0, 64, ctx->stream));
    // the build side: rows {key, payload}
    const u64 * bkeys0 = j->blocks[0].keys;
    hipLaunchKernelGGL((k_rp_hist_wide<u64, JoinRadixFn1>), dim3(G), dim3(RP_THREADS), 0, ctx->stream, bkeys0, nb, b_rpw, P1, b_counts, fn1);
    CHGPU_TRY(chgpu_scan_exclusive_u32_u64(ctx, b_counts, b_offsets, m, total_dev, tmp, tmp_b));
    {
        const size_t lds = rp_scatter_lds_bytes(8192, P1, 8, true);
        auto scat = k_rp_scatter<8192, u64, true, JoinRadixFn1>;
        CHGPU_HIP(hipFuncSetAttribute((const void *)scat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(scat, dim3(G), dim3(RP_THREADS), lds, ctx->stream, bkeys0, (const u64 *)right_payload->data, nb, b_rpw, P1, (const u64 *)b_offsets, bk1, bw1, fn1);
        const size_t lds2 = (size_t)JBS_TILE * 16 + (size_t)(PB + 1) * 8 + 64;
        auto sortk = k_rp_tilesort_keys<JBS_TILE, JoinRadixFn2, RP_THREADS, true>;
        CHGPU_HIP(hipFuncSetAttribute((const void *)sortk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(sortk, dim3(G), dim3(RP_THREADS), lds2, ctx->stream, (const u64 *)bk1, nb, b_rpw2, PB, bk2, bix, fn2, stray, (const u64 *)bw1, bw2);
    }
    // the probe side: keys
    hipLaunchKernelGGL((k_rp_hist_wide<u64, JoinRadixFn1>), dim3(G), dim3(RP_THREADS), 0, ctx->stream, (const u64 *)key_col->data, n, p_rpw, P1, p_counts, fn1);
    CHGPU_TRY(chgpu_scan_exclusive_u32_u64(ctx, p_counts, p_offsets, m, total_dev, tmp, tmp_b));
    {
        const size_t lds = rp_scatter_lds_bytes(12288, P1, 8, false);
        auto scat = k_rp_scatter<12288, u64, false, JoinRadixFn1>;
        CHGPU_HIP(hipFuncSetAttribute((const void *)scat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(scat, dim3(G), dim3(RP_THREADS), lds, ctx->stream, (const u64 *)key_col->data, (const u64 *)nullptr, n, p_rpw, P1, (const u64 *)p_offsets, pk1,
                           (u64 *)nullptr, fn1);
        const size_t lds2 = (size_t)JPL2_TILE * 8 + (size_t)(PB + 1) * 8 + 64;
        auto sortk = k_rp_tilesort_keys<JPL2_TILE, JoinRadixFn2>;
        CHGPU_HIP(hipFuncSetAttribute((const void *)sortk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(sortk, dim3(G), dim3(RP_THREADS), lds2, ctx->stream, (const u64 *)pk1, n, p_rpw2, PB, pk2, pix, fn2, stray, (const u64 *)nullptr, (u64 *)nullptr);
    }
    {
        JoinTable vt{}; // only the capacity is read
        vt.capacity = cap;
        const size_t lds = (size_t)(JPL2_CELLS + JPL2_TAIL + 2) * 16;
        CHGPU_HIP(hipFuncSetAttribute((const void *)k_join_probe_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int jexp = CHGPU_EXPERIMENT(ctx, "experiment_join_lds");
        hipLaunchKernelGGL(k_join_probe_lds<true>, dim3(2 * G), dim3(JPL2_THREADS), lds, ctx->stream, vt, variant | (jexp << 8), (const u64 *)pk2, n, (const u64 *)p_offsets, G, lg_p2,
                           (const unsigned short *)pix, (const u64 *)nullptr, (const u64 *)nullptr, (u64)1, unit_ctr, stray, result2, (const u64 *)bk2, (const u64 *)bw2,
                           (const u64 *)b_offsets, (const unsigned short *)bix, nb, dupf);
    }
    ctx->counters[6] += 9;
    CHGPU_HIP(hipGetLastError());
    u64 back[4];
    CHGPU_TRY(chgpu_read_back(ctx, result2, back, 32));
    if ((back[2] >> 32) != 0 || (u32)back[3] != 0) // stray rows / a duplicate build key: not this plan
        return CHGPU_ERR_NOT_IMPLEMENTED;
    res[0] = back[0];
    res[1] = back[1];
    return CHGPU_OK;
}

extern "C" int chgpu_join_probe_agg(chgpu_join * j, const chgpu_col * key_col, const chgpu_col * null_map, const chgpu_col * right_payload,
                                    uint64_t * count_out, void * sum_out)
{
    ChgpuDeviceGuard _dev_guard(j ? j->ctx : nullptr);
    CHGPU_REQUIRE(j && key_col && count_out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(!right_payload || sum_out, CHGPU_ERR_BAD_ARGUMENTS, "sum_out must not be NULL when a payload column is given");
    CHGPU_REQUIRE(key_col->type == j->key_type, CHGPU_ERR_BAD_ARGUMENTS, "left key column has type %d, expected %d", key_col->type, j->key_type);
    if (null_map)
        CHGPU_REQUIRE(null_map->type == CHGPU_U8 && null_map->rows == key_col->rows, CHGPU_ERR_SIZES_MISMATCH, "null map size mismatch");
    CHGPU_REQUIRE(!(j->kind == CHGPU_JOIN_INNER && j->strictness == CHGPU_STRICT_ANY), CHGPU_ERR_NOT_IMPLEMENTED,
                  "INNER ANY consumes right rows across joinBlock calls (setUsedOnce): use chgpu_join_probe");
    CHGPU_REQUIRE(!jf_track_used(j), CHGPU_ERR_NOT_IMPLEMENTED, "RIGHT / FULL joins emit non-joined rows afterwards: use chgpu_join_probe");
    if (right_payload)
    {
        CHGPU_REQUIRE(chgpu_type_size(right_payload->type) != 0, CHGPU_ERR_BAD_ARGUMENTS, "payload type %d", right_payload->type);
        CHGPU_REQUIRE(right_payload->rows == j->total_rows, CHGPU_ERR_SIZES_MISMATCH,
                      "payload column has %llu rows, the right side has %llu (it must be the right blocks' column glued in insertion order)",
                      (unsigned long long)right_payload->rows, (unsigned long long)j->total_rows);
    }
    chgpu_ctx * ctx = j->ctx;
    const u64 n = key_col->rows;
    int variant;
    if (j->strictness == CHGPU_STRICT_ALL) variant = jf_left_kind(j) == CHGPU_JOIN_LEFT ? PV_ALL_LEFT : PV_ALL_INNER;
    else if (j->strictness == CHGPU_STRICT_SEMI) variant = PV_SEMI_LEFT;
    else if (j->strictness == CHGPU_STRICT_ANTI) variant = PV_ANTI_LEFT;
    else variant = PV_ANY_LEFT;
    const bool is_float = right_payload && chgpu_type_is_float(right_payload->type);
    u64 res[2] = {0, 0};
    int plan = CHGPU_ERR_NOT_IMPLEMENTED;
    j->build_closed = true; // (a probe ends the build phase, as a joinBlock does)
    if (n && !null_map && !j->finished)
    {
        plan = join_probe_agg_radix(j, key_col, right_payload, variant, res); // no table at all: both sides partitioned, slices built and probed in LDS
        if (plan != CHGPU_OK && plan != CHGPU_ERR_NOT_IMPLEMENTED)
            return plan;
    }
    if (plan != CHGPU_OK && !j->finished)
        CHGPU_TRY(join_build_table(j));
    if (plan != CHGPU_OK && n && !null_map)
    {
        plan = join_probe_agg_lds(j, key_col, right_payload, variant, res); // table slices staged in LDS (unique keys, integer payload)
        if (plan == CHGPU_ERR_NOT_IMPLEMENTED)
            plan = join_probe_agg_regions(j, key_col, right_payload, variant, res); // table regions resident in L2
    }
    if (plan != CHGPU_OK && plan != CHGPU_ERR_NOT_IMPLEMENTED)
        return plan;
    if (n && plan != CHGPU_OK)
    {
        const u32 grid = chgpu_grid_for(ctx, (n + 3) / 4, JT, 8);
        void * scratch = nullptr;
        CHGPU_TRY(chgpu_scratch(ctx, ((size_t)grid + 1) * 16, &scratch));
        u64 * partials = (u64 *)scratch;
        u64 * out2 = partials + 2 * (size_t)grid;
        const void * pp = right_payload ? right_payload->data : nullptr;
        const int pt = right_payload ? right_payload->type : CHGPU_U64;
        const u8 * nm = null_map ? (const u8 *)null_map->data : nullptr;
#define CHGPU_JPA(PFV, FL)                                                                                                                              \
    hipLaunchKernelGGL((k_join_probe_agg<PFV, FL>), dim3(grid), dim3(JT), 0, ctx->stream, j->t, variant, (const void *)key_col->data, j->key_type, nm, n, pp, pt, \
                       (const u64 *)j->block_base_dev, (u64)j->blocks.size(), partials)
        if (j->t.pf)
        {
            if (is_float) CHGPU_JPA(true, true); else CHGPU_JPA(true, false);
        }
        else
        {
            if (is_float) CHGPU_JPA(false, true); else CHGPU_JPA(false, false);
        }
#undef CHGPU_JPA
        if (is_float)
            hipLaunchKernelGGL(k_join_probe_agg_finish<true>, dim3(1), dim3(64), 0, ctx->stream, (const u64 *)partials, grid, out2);
        else
            hipLaunchKernelGGL(k_join_probe_agg_finish<false>, dim3(1), dim3(64), 0, ctx->stream, (const u64 *)partials, grid, out2);
        ctx->counters[6] += 2;
        CHGPU_HIP(hipGetLastError());
        CHGPU_TRY(chgpu_read_back(ctx, out2, res, sizeof(res)));
    }
    *count_out = res[0];
    if (sum_out)
        memcpy(sum_out, &res[1], 8);
    ctx->counters[3] += n;
    ctx->counters[4] += res[0];
    return CHGPU_OK;
}
#include "join_chain.h"
