// agg_kernels.hip — GROUP BY on the device: Aggregator::executeOnBlock / merge / convertToBlocks for one numeric key
// and POD-state aggregate functions (count, sum, avg).
//
// Reference loops replaced (file:line in the reference checkout):
//   k_agg_rows      Aggregator::executeImplBatch loops A+B (emplaceKey + addBatch)   src/Interpreters/Aggregator.cpp:1010-1206,
//                   HashTable::emplace / findCell                                     src/Common/HashTable/HashTable.h:448-459,901-1027,
//                   IAggregateFunctionHelper::addBatch                                src/AggregateFunctions/IAggregateFunction.h:428-452
//   k_agg_tuples    mergeDataImpl / HashMap::mergeToViaEmplace, resize+reinsert       Aggregator.cpp:2468-2521, HashMap.h:203-233,
//                                                                                     HashTable.h:504-593
//   finalize        convertToBlockImplFinal / insertResultsIntoColumns                Aggregator.cpp:1948-2117
//
// Table geometry mirrors the reference's (HashTable.h:217-330,358-391): power-of-two capacity, linear probing,
// max fill 1/2, growth x4 until 2^23 cells then x2, empty <=> key == 0, the zero key kept out of line (slot index
// == capacity).  Layout is SoA in HBM: keys[capacity+1], then one u64/f64 array per state word, so probes touch only
// 8-byte keys and state updates are single 64-bit atomics.  Placement hash = intHash64 (Hash.h:27-36); CRC32-C is
// only needed where the hash is externally visible (partition_kernels.hip).
//
// Strategies (chgpu_agg_add_block picks by promised / observed cardinality):
//   LDS-STAGED  a per-workgroup open-addressing table in LDS absorbs repeated keys -- the device analogue of the
//               consecutive-key cache, ColumnsHashingImpl.h:313-366 -- and is flushed once per workgroup; k_agg_part_lds in
//               RANGE mode straight over the source columns (k_agg_rows_lds is the older generic form of it);
//   PARTITIONED k_gb_hist -> scan -> k_gb_scatter -> k_gb_units -> k_agg_part_lds, one or two partitioning levels;
//   DIRECT      k_agg_rows_direct, one HBM atomic per row and state word.
// Rows that would push the table over max fill are marked in a pending bitmap; the host grows the table (rehash) and
// re-runs only those rows, which is the reference's resize-on-overflow (HashTable.h:921-944) restructured for a device
// that cannot realloc inside a kernel.
#include "chgpu_internal.h"
#include "radix_partition.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

static constexpr u32 AGG_MAX_AGGS = 8;
static constexpr u32 AGG_MAX_WORDS = 16;
static constexpr u64 AGG_MIN_CAPACITY = 1ull << 22; // 4 Mi cells: >= 2 Mi cells of slack above max fill for the LDS flushes of ~1000 workgroups
static constexpr u32 AGG_LDS_BYTES = 52 * 1024;    // LDS table budget per workgroup (3 workgroups per CU; 2048 cells x 24 B fits)
static constexpr u32 AGG_THREADS = 256;

// The group counter is striped: one same-address atomic per wave and claim serialises at ~10 ns each -- 16 M groups cost
// 160 ms in the counter alone, and a rehash of 16 M cells 2.9 ms instead of 0.2.  A wave adds to (and checks the soft
// limit against) the stripe picked by its workgroup/wave index, each stripe on its own 128-byte line.
static constexpr u32 AGG_STRIPES = 64;
struct AggCtrl
{
    unsigned long long n_groups; // host side only: sum of the stripes, filled in by agg_read_ctrl
    u32 overflow;                // some row hit the max-fill limit and was left pending
    u32 has_zero;
    u32 fatal;                   // table completely full during a flush (cannot happen by construction)
    u32 pad;
    unsigned long long pad2[13];
    unsigned long long stripe[AGG_STRIPES][16]; // occupied cells incl. the zero key, [s][0] is the counter
};
static constexpr size_t AGG_HDR_BYTES = (sizeof(AggCtrl) + 255) / 256 * 256;

struct AggArg
{
    const void * ptr; // argument column (NULL for count)
    int kind;
    int arg_type;
    u32 word;         // first state word
    u32 pre;          // partitioned path: index of this argument's word column in the partition buffers
};

struct AggDesc
{
    u32 n_aggs;
    u32 n_words;
    AggArg a[AGG_MAX_AGGS];
    u32 word_is_f64; // bit w set: state word w is Float64
    // deterministic Float64 sums (see Fx128): bit w set = word w is the LOW half of a 128-bit fixed-point sum whose high half is word
    // fx_hi[w] (one of the words appended behind the regular ones); values are multiples of 2^fx_base
    u32 word_fx;
    u32 word_fx_hi; // the high halves (never updated on their own)
    unsigned char fx_hi[AGG_MAX_WORDS];
    int fx_base;
    u64 row_seq; // any(): row i of the argument columns is the (row_seq + i)-th row this aggregation has seen (modulo 2^64)
    // A pass over a SUBSET of the functions (one argument word at a time through the tile-sorted plan) numbers its state words locally
    // (0 .. n_words - 1: the LDS cells hold only those) and finds the table's words through this map; the identity otherwise.
    unsigned char word_map[AGG_MAX_WORDS];
};

struct AggTable
{
    u64 * keys;      // [capacity + 1]
    u64 * words;     // [n_words][capacity + 1]
    u64 capacity;    // power of two
    u64 max_fill;    // capacity / 2
    AggCtrl * ctrl;
};

struct chgpu_agg
{
    chgpu_ctx * ctx = nullptr;
    int key_type = -1;
    u32 n_aggs = 0, n_words = 0;
    int kinds[AGG_MAX_AGGS];
    int arg_types[AGG_MAX_AGGS];
    u32 word_off[AGG_MAX_AGGS];
    u32 word_is_f64 = 0;
    u64 size_hint = 0;
    AggTable t{nullptr, nullptr, 0, 0, nullptr};
    void * table_mem = nullptr; // from the context's column pool (stream-ordered reuse: no hipMalloc/hipFree per query)
    size_t table_class = 0;
    u64 n_groups = 0; // host copy, refreshed after every call
    bool hint_probed = false; // the cardinality of a hint-less aggregation was sampled on its first large block
    bool has_extremum = false; // some function is min / max / any: rows take the DIRECT kernel (the LDS-staged plans only know how to add)
    // any(): {claim, value} words.  claim = ~(ordinal of the row that set the value) under an unsigned max: the EARLIEST row of the group
    // wins whatever order the hardware serves the rows in; the value is stored by a second pass from the winner's row (k_agg_any_resolve)
    u32 word_any = 0; // bit w: word w is a claim, word w + 1 its value
    u64 any_seq = 0;  // rows seen so far
    u64 nokey_kept = 0; // without key: rows that reached the states (0 = min / max / any have no value: insertResultInto gives the default)
    // deterministic Float64 sums (option deterministic_float_sums, the default): sum / avg over a float argument keep a 128-bit fixed-point
    // state {word, fx_hi[word]} in units of 2^fx_base instead of a double.  n_words counts the appended high halves too; the first
    // n_pub_words are the words the C ABI shows (state columns, wire format): exports fold a pair back into its Float64 column.
    u32 n_pub_words = 0;
    u32 word_fx = 0, word_fx_hi = 0;
    unsigned char fx_hi[AGG_MAX_WORDS] = {0};
    // window invariant: every state is a sum of at most fx_rows values, each below 2^(127 - fx_log_cap) units of 2^fx_base
    int fx_base = 0;
    bool fx_base_set = false;
    u64 fx_rows = 0;
    int fx_log_cap = 30;
    int fx_emin = 4096; // smallest (unbiased) exponent among the non-zero values added so far
    u64 host_words[AGG_MAX_WORDS]; // without_key states live on the host (8 B each)
};

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 load_key_zext(const void * keys, int type, u64 i)
{
    // HashMethodOneNumber::getKeyHolder (ColumnsHashing/HashMethod.h:91): raw bits of the key, zero-extended to the
    // UInt64 table key (AggregatedDataVariants.h:63-64)
    switch (type)
    {
        case CHGPU_U32: case CHGPU_I32: return ((const u32 *)keys)[i];
        case CHGPU_U16: case CHGPU_I16: return ((const u16 *)keys)[i];
        case CHGPU_U8: case CHGPU_I8: return ((const u8 *)keys)[i];
        default: return ((const u64 *)keys)[i];
    }
}

__device__ __forceinline__ u64 load_arg_bits(const void * p, int type, u64 i)
{
    switch (type)
    {
        case CHGPU_I64: case CHGPU_U64: case CHGPU_F64: return ((const u64 *)p)[i];
        case CHGPU_U32: return ((const u32 *)p)[i];
        case CHGPU_I32: return (u64)(i64)((const i32 *)p)[i]; // sign-extend: wrap-around two's complement sum
        case CHGPU_U8: return ((const u8 *)p)[i];
        case CHGPU_U16: return ((const u16 *)p)[i];
        case CHGPU_I16: return (u64)(i64)((const i16 *)p)[i];
        case CHGPU_I8: return (u64)(i64)((const i8 *)p)[i];
        case CHGPU_F32: return (u64)__double_as_longlong((double)((const float *)p)[i]); // Float32 is accumulated as Float64
        default: return 0;
    }
}

// op: 0 = wrap-around integer add, 1 = Float64 add, 2 = unsigned max (min / max states: order keys, see agg_order_key)
__device__ __forceinline__ void global_add_word(u64 * p, u64 bits, int op)
{
    if (op == 2)
    {
        if (bits) // (0 is the identity: nothing to do)
            __hip_atomic_fetch_max((unsigned long long *)p, (unsigned long long)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const bool is_f64 = op == 1;
    if (is_f64)
        __hip_atomic_fetch_add((double *)p, __longlong_as_double((long long)bits), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        __hip_atomic_fetch_add((unsigned long long *)p, (unsigned long long)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- deterministic Float64 sums -------------------------------------------------------------------------------------------------
// A double atomicAdd makes a group's sum depend on the order the hardware happened to serve the rows in (the reference is deterministic
// for a fixed block split: AggregateFunctionSum.h:72-101 adds in row order).  Integer addition is associative, so the state of sum /
// avg over a float argument is a two's complement 128-bit integer in units of 2^base instead: every row contributes trunc(x * 2^-base),
// whatever the order, the plan or the number of workgroups.  base = (largest exponent seen) - 96: the window holds 2^30 rows of the
// largest magnitude and keeps every bit of values down to 2^-44 of it (2^-20 relative precision down to 2^-76 of it) -- the host widens
// the window (an arithmetic shift of every state) when a block brings a larger exponent or the row count outgrows the head room.
// The two halves are separate words: low += x.lo returns the old value, and the adder that sees the wrap carries into the high word --
// each wrap is seen by exactly one adder, so the pair ends at the exact sum.
struct Fx128
{
    u64 lo, hi;
};
__device__ __host__ __forceinline__ Fx128 fx_from_double(u64 bits, int base)
{
    const u32 e = (u32)(bits >> 52) & 0x7ffu;
    u64 m = bits & 0xFFFFFFFFFFFFFull;
    if (e)
        m |= 1ull << 52;
    const int sh = (int)(e ? e : 1u) - 1075 - base; // x = m * 2^(e - 1075)
    Fx128 r;
    if (sh >= 64)
        r.lo = 0, r.hi = sh < 128 ? m << (sh - 64) : 0; // (sh + 53 <= 97 by the choice of base)
    else if (sh > 0)
        r.lo = m << sh, r.hi = m >> (64 - sh);
    else if (sh > -64)
        r.lo = m >> -sh, r.hi = 0;
    else
        r.lo = 0, r.hi = 0;
    if (bits >> 63)
    {
        r.lo = ~r.lo + 1;
        r.hi = ~r.hi + (r.lo == 0 ? 1 : 0);
    }
    return r;
}
// the pair as the nearest double (ties to even): the only rounding of the whole sum
__device__ __host__ __forceinline__ double fx_to_double(u64 lo, u64 hi, int base)
{
    const bool neg = (hi >> 63) != 0;
    if (neg)
    {
        lo = ~lo + 1;
        hi = ~hi + (lo == 0 ? 1 : 0);
    }
    if ((lo | hi) == 0)
        return 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
    const int top = hi ? 127 - __clzll((long long)hi) : 63 - __clzll((long long)lo);
#else
    const int top = hi ? 127 - __builtin_clzll(hi) : 63 - __builtin_clzll(lo);
#endif
    u64 m;
    int sh = top - 52; // bits dropped
    if (sh <= 0)
        m = lo, sh = 0; // (top <= 52: the magnitude is exact in a double)
    else
    {
        // m = magnitude >> sh, rem = the dropped bits against one half
        u64 rem_hi, rem_lo; // dropped bits, left-aligned in 128 bits
        if (sh < 64)
        {
            m = (lo >> sh) | (hi << (64 - sh));
            rem_hi = lo << (64 - sh);
            rem_lo = 0;
        }
        else if (sh == 64)
        {
            m = hi;
            rem_hi = lo;
            rem_lo = 0;
        }
        else
        {
            m = hi >> (sh - 64);
            rem_hi = (hi << (128 - sh)) | (lo >> (sh - 64));
            rem_lo = lo << (128 - sh);
        }
        const u64 half = 1ull << 63;
        if (rem_hi > half || (rem_hi == half && (rem_lo != 0 || (m & 1))))
            ++m; // (2^53 is a double too)
    }
    const double v = ldexp((double)m, sh + base);
    return neg ? -v : v;
}
__device__ __forceinline__ void global_add_fx(u64 * lo, u64 * hi, Fx128 x)
{
    u64 h = x.hi;
    if (x.lo)
    {
        const u64 old = __hip_atomic_fetch_add((unsigned long long *)lo, (unsigned long long)x.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h += (old + x.lo < old) ? 1 : 0;
    }
    if (h)
        __hip_atomic_fetch_add((unsigned long long *)hi, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lds_add_fx(u64 * lo, u64 * hi, Fx128 x)
{
    u64 h = x.hi;
    if (x.lo)
    {
        const u64 old = atomicAdd((unsigned long long *)lo, (unsigned long long)x.lo);
        h += (old + x.lo < old) ? 1 : 0;
    }
    if (h)
        atomicAdd((unsigned long long *)hi, (unsigned long long)h);
}

// Find-or-claim the cell of `key` (emplace).  Returns the slot, or ~0 when the row must wait for a bigger table.
// soft_limit: refuse to claim new cells once n_groups >= max_fill (rows); flushes/rehash pass false and may use
// the slack above max fill.  Every loop is bounded by the capacity, so the wave always exits.
__device__ __forceinline__ u32 agg_stripe() { return (blockIdx.x * 5 + (threadIdx.x >> 6)) & (AGG_STRIPES - 1); }

__device__ __forceinline__ void count_claim(const AggTable & t, bool claimed)
{
    // one atomic per wave, on the wave's stripe of the group counter
    const u64 cb = __ballot(claimed);
    if (claimed && mbcnt(cb) == 0)
        atomicAdd(&t.ctrl->stripe[agg_stripe()][0], (unsigned long long)__popcll(cb));
}
__device__ __forceinline__ u64 table_emplace_impl(const AggTable & t, u64 key, bool soft_limit, bool & claimed)
{
    claimed = false;
    if (key == 0)
    {
        // zero key lives out of line (HashTable.h:874-898)
        if (__hip_atomic_load(&t.ctrl->has_zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            if (atomicExch(&t.ctrl->has_zero, 1u) == 0)
                claimed = true;
        return t.capacity;
    }
    const u64 mask = t.capacity - 1;
    u64 slot = dev_intHash64(key) & mask;
    for (u64 step = 0; step < t.capacity; ++step)
    {
        u64 k = t.keys[slot];
        if (k == key)
            return slot;
        if (k == 0)
        {
            if (soft_limit && __hip_atomic_load(&t.ctrl->stripe[agg_stripe()][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= t.max_fill / AGG_STRIPES)
                return ~0ull;
            const u64 prev = atomicCAS((unsigned long long *)&t.keys[slot], 0ull, (unsigned long long)key);
            if (prev == 0)
            {
                claimed = true;
                return slot;
            }
            if (prev == key)
                return slot;
        }
        slot = (slot + 1) & mask;
    }
    return ~0ull;
}

__device__ __forceinline__ u64 table_emplace(const AggTable & t, u64 key, bool soft_limit)
{
    bool claimed;
    const u64 slot = table_emplace_impl(t, key, soft_limit, claimed);
    count_claim(t, claimed);
    return slot;
}

// min / max states (AggregateFunctionsMinMax.cpp, SingleValueDataFixed<T>::setIfSmaller / setIfGreater, SingleValueData.cpp:219-262): the
// state word holds an ORDER KEY -- the value mapped to an unsigned 64-bit integer that sorts like the value (unsigned: itself; signed: sign
// bit flipped; floats: the IEEE total-order fold, Float32 after its exact widening) -- for max, and its complement for min, combined with
// an unsigned atomic max.  A freshly zeroed cell is then the identity of both, exactly like the sums' zero: the rehash, the merges and the
// exports move min / max words with no special case but the combining operation.  (A group only exists because a row created it, so
// `has()` is always true for it.)  NaNs take their total-order place -- above +inf / below -inf by sign -- where the reference's answer
// depends on which row came first (`NaN < x` is false either way).
__device__ __host__ __forceinline__ u64 agg_order_key(u64 bits, int type)
{
    switch (type)
    {
        case CHGPU_I64: case CHGPU_I32: case CHGPU_I16: case CHGPU_I8: return bits ^ 0x8000000000000000ull; // (already sign-extended)
        case CHGPU_F64: case CHGPU_F32: return (bits >> 63) ? ~bits : bits ^ 0x8000000000000000ull;
        default: return bits;
    }
}
__device__ __host__ __forceinline__ u64 agg_order_key_inverse(u64 key, int type)
{
    switch (type)
    {
        case CHGPU_I64: case CHGPU_I32: case CHGPU_I16: case CHGPU_I8: return key ^ 0x8000000000000000ull;
        case CHGPU_F64: case CHGPU_F32: return (key >> 63) ? key ^ 0x8000000000000000ull : ~key;
        default: return key;
    }
}

// add row i's contribution of every aggregate to the cell `slot` (IAggregateFunction::add per function)
__device__ __forceinline__ void add_row_global(const AggTable & t, const AggDesc & d, u64 slot, u64 i)
{
    const u64 stride = t.capacity + 1;
    for (u32 j = 0; j < d.n_aggs; ++j)
    {
        const AggArg & a = d.a[j];
        u64 * w = t.words + (u64)d.word_map[a.word] * stride + slot;
        if (a.kind == CHGPU_AGG_COUNT)
            global_add_word(w, 1, false);
        else if (a.kind == CHGPU_AGG_MIN || a.kind == CHGPU_AGG_MAX)
        {
            const u64 k = agg_order_key(load_arg_bits(a.ptr, a.arg_type, i), a.arg_type);
            global_add_word(w, a.kind == CHGPU_AGG_MAX ? k : ~k, 2);
        }
        else if (a.kind == CHGPU_AGG_ANY)
            global_add_word(w, ~(d.row_seq + i), 2); // the claim of the earliest row; its value follows in k_agg_any_resolve
        else
        {
            if ((d.word_fx >> a.word) & 1)
                global_add_fx(w, t.words + (u64)d.word_map[d.fx_hi[a.word]] * stride + slot, fx_from_double(load_arg_bits(a.ptr, a.arg_type, i), d.fx_base));
            else
                global_add_word(w, load_arg_bits(a.ptr, a.arg_type, i), a.arg_type == CHGPU_F64 || a.arg_type == CHGPU_F32);
            if (a.kind == CHGPU_AGG_AVG)
                global_add_word(w + stride, 1, false); // denominator
        }
    }
}

// The same update from values already in registers: `bits0/bits1` are the 8-byte argument words selected by AggArg::pre,
// `cnt` the number of rows they stand for.
__device__ __forceinline__ void add_vals_global(const AggTable & t, const AggDesc & d, u64 slot, u64 bits0, u64 bits1, u64 cnt)
{
    const u64 stride = t.capacity + 1;
    for (u32 j = 0; j < d.n_aggs; ++j)
    {
        const AggArg & a = d.a[j];
        u64 * w = t.words + (u64)d.word_map[a.word] * stride + slot;
        if (a.kind == CHGPU_AGG_COUNT)
            global_add_word(w, cnt, false);
        else
        {
            if ((d.word_fx >> a.word) & 1)
                global_add_fx(w, t.words + (u64)d.word_map[d.fx_hi[a.word]] * stride + slot, fx_from_double(a.pre == 0 ? bits0 : bits1, d.fx_base));
            else
                global_add_word(w, a.pre == 0 ? bits0 : bits1, a.arg_type == CHGPU_F64 || a.arg_type == CHGPU_F32);
            if (a.kind == CHGPU_AGG_AVG)
                global_add_word(w + stride, cnt, false); // denominator
        }
    }
}

enum { AGG_MODE_ALL = 0, AGG_MODE_PENDING = 1 };

// DIRECT kernel: one global emplace + one atomic per state word per row.
template <int MODE>
__global__ __launch_bounds__(AGG_THREADS) void k_agg_rows_direct(AggTable t, AggDesc d, const void * __restrict__ keys, int key_type,
                                                                 u64 row_begin, u64 n, u64 * __restrict__ pending)
{
    const u32 lane = threadIdx.x & 63;
    const u64 wave0 = ((u64)blockIdx.x * AGG_THREADS + threadIdx.x) >> 6;
    const u64 n_waves = ((u64)gridDim.x * AGG_THREADS) >> 6;
    const u64 n_groups64 = (n + 63) / 64;
    for (u64 g = wave0; g < n_groups64; g += n_waves)
    {
        const u64 r = g * 64 + lane;
        bool active = r < n;
        if (MODE == AGG_MODE_PENDING)
        {
            const u64 word = pending[g];
            if (word == 0)
                continue;
            active = active && ((word >> lane) & 1);
        }
        bool failed = false;
        if (active)
        {
            const u64 i = row_begin + r;
            const u64 slot = table_emplace(t, load_key_zext(keys, key_type, i), true);
            if (slot == ~0ull)
                failed = true;
            else
                add_row_global(t, d, slot, i);
        }
        const u64 b = __ballot(failed);
        if (lane == 0)
            pending[g] = b;
        if (b != 0 && lane == 0)
            t.ctrl->overflow = 1; // benign race: every writer stores 1
    }
}

// LDS-STAGED kernel.  Dynamic LDS: lkeys[S+1] then lwords[n_words][S+1]; cell S is the zero key's.
__global__ __launch_bounds__(1024) void k_agg_rows_lds(AggTable t, AggDesc d, const void * __restrict__ keys, int key_type,
                                                              u64 row_begin, u64 n, u64 * __restrict__ pending, u32 S)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u64 * lkeys = (u64 *)lds_raw;
    u64 * lwords = lkeys + (S + 1);
    __shared__ u32 lzero;
    const u32 lstride = S + 1;
    for (u32 s = threadIdx.x; s < (d.n_words + 1) * lstride; s += blockDim.x)
        lkeys[s] = 0;
    if (threadIdx.x == 0)
        lzero = 0;
    __syncthreads();

    const u32 lane = threadIdx.x & 63;
    // aggregate descriptors decoded once into wave-uniform registers (loops over them are fully unrolled): an s_load of d.a[j]
    // per function and row group also drains the wave's LDS queue through lgkmcnt(0) -- see k_agg_part_lds
    const void * a_ptr[AGG_MAX_AGGS];
    int a_kind[AGG_MAX_AGGS], a_type[AGG_MAX_AGGS];
    u32 a_word[AGG_MAX_AGGS], a_hi[AGG_MAX_AGGS]; // a_hi: the high word of a fixed-point sum (0 = an ordinary state word)
    const int fx_base = d.fx_base;
#pragma unroll
    for (u32 j = 0; j < AGG_MAX_AGGS; ++j)
    {
        const bool on = j < d.n_aggs;
        a_ptr[j] = on ? d.a[j].ptr : nullptr;
        a_kind[j] = on ? d.a[j].kind : -1;
        a_type[j] = on ? d.a[j].arg_type : 0;
        a_word[j] = on ? d.a[j].word : 0;
        a_hi[j] = (on && d.a[j].kind != CHGPU_AGG_COUNT && ((d.word_fx >> d.a[j].word) & 1)) ? d.fx_hi[d.a[j].word] : 0;
    }
    const u64 wave0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u64 n_waves = ((u64)gridDim.x * blockDim.x) >> 6;
    const u64 n_groups64 = (n + 63) / 64;
    // Each wave takes LDS_R consecutive 64-row groups per iteration and issues ALL their key/argument loads before
    // touching LDS: with one row per lane per iteration only ~24 KiB of 4-8-byte loads were in flight per CU and the
    // kernel was latency-bound (measured 18.6 -> 10.2 ms from more waves alone).
    constexpr int LDS_R = 4;
    constexpr u32 PRE = 3; // argument columns preloaded per row (further ones are loaded on use)
    for (u64 g0 = wave0 * LDS_R; g0 < n_groups64; g0 += n_waves * LDS_R)
    {
        u64 keyv[LDS_R];
        u64 argv[LDS_R][PRE];
        bool act[LDS_R];
#pragma unroll
        for (int q = 0; q < LDS_R; ++q)
        {
            const u64 r = (g0 + q) * 64 + lane;
            act[q] = r < n;
            const u64 i = row_begin + (act[q] ? r : 0);
            keyv[q] = act[q] ? load_key_zext(keys, key_type, i) : 0;
#pragma unroll
            for (u32 j = 0; j < PRE; ++j)
                argv[q][j] = (act[q] && a_kind[j] >= 0 && a_kind[j] != CHGPU_AGG_COUNT) ? load_arg_bits(a_ptr[j], a_type[j], i) : 0;
        }
#pragma unroll
        for (int q = 0; q < LDS_R; ++q)
        {
            const u64 g = g0 + q;
            if (g >= n_groups64)
                break;
            bool failed = false;
            if (act[q])
            {
                const u64 i = row_begin + g * 64 + lane;
                const u64 key = keyv[q];
                // ---- LDS emplace: linear probing, give up after 32 cells (a nearly full LDS table) -> HBM path ----
                u32 ls = ~0u;
                if (key == 0)
                {
                    ls = S;
                    lzero = 1;
                }
                else
                {
                    u32 s = (u32)(dev_intHash64(key) >> 40) & (S - 1);
#pragma unroll 1
                    for (int probe = 0; probe < 32; ++probe)
                    {
                        u64 k = lkeys[s];
                        if (k == 0)
                            k = atomicCAS((unsigned long long *)&lkeys[s], 0ull, (unsigned long long)key), k = (k == 0) ? key : k;
                        if (k == key)
                        {
                            ls = s;
                            break;
                        }
                        s = (s + 1) & (S - 1);
                    }
                }
                if (ls != ~0u)
                {
#pragma unroll
                    for (u32 j = 0; j < AGG_MAX_AGGS; ++j)
                    {
                        if (a_kind[j] < 0)
                            break;
                        u64 * w = lwords + a_word[j] * lstride + ls;
                        if (a_kind[j] == CHGPU_AGG_COUNT)
                            atomicAdd((unsigned long long *)w, 1ull);
                        else
                        {
                            u64 bits;
                            if (j < PRE) bits = argv[q][j < PRE ? j : 0];
                            else bits = load_arg_bits(a_ptr[j], a_type[j], i);
                            if (a_hi[j])
                                lds_add_fx(w, lwords + a_hi[j] * lstride + ls, fx_from_double(bits, fx_base));
                            else if (a_type[j] == CHGPU_F64 || a_type[j] == CHGPU_F32)
                                atomicAdd((double *)w, __longlong_as_double((long long)bits));
                            else
                                atomicAdd((unsigned long long *)w, (unsigned long long)bits);
                            if (a_kind[j] == CHGPU_AGG_AVG)
                                atomicAdd((unsigned long long *)(w + lstride), 1ull);
                        }
                    }
                }
                else
                {
                    const u64 slot = table_emplace(t, key, true);
                    if (slot == ~0ull)
                        failed = true;
                    else
                        add_row_global(t, d, slot, i);
                }
            }
            const u64 b = __ballot(failed);
            if (lane == 0)
                pending[g] = b;
            if (b != 0 && lane == 0)
                t.ctrl->overflow = 1;
        }
    }
    __syncthreads();

    // ---- flush the workgroup's partial states: one emplace + n_words atomics per distinct key ----
    const u64 gstride = t.capacity + 1;
    for (u32 s = threadIdx.x; s <= S; s += blockDim.x)
    {
        const u64 key = lkeys[s];
        const bool occupied = (s == S) ? (lzero != 0) : (key != 0);
        if (!occupied)
            continue;
        const u64 slot = table_emplace(t, s == S ? 0 : key, false); // may use the slack above max fill
        if (slot == ~0ull)
        {
            t.ctrl->fatal = 1;
            continue;
        }
        for (u32 w = 0; w < d.n_words; ++w)
        {
            if ((d.word_fx_hi >> w) & 1)
                continue; // flushed with its low half
            const u64 bits = lwords[w * lstride + s];
            if ((d.word_fx >> w) & 1)
            {
                const u32 wh = d.fx_hi[w];
                const u64 hb = lwords[wh * lstride + s];
                if (bits | hb)
                    global_add_fx(t.words + (u64)d.word_map[w] * gstride + slot, t.words + (u64)d.word_map[wh] * gstride + slot, Fx128{bits, hb});
                continue;
            }
            const bool f = (d.word_is_f64 >> w) & 1;
            if (f ? (__longlong_as_double((long long)bits) != 0.0 || bits != 0) : (bits != 0))
                global_add_word(t.words + (u64)d.word_map[w] * gstride + slot, bits, f);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// PARTITIONED path for large cardinalities (config C3: 1 M groups over 1 B rows).
// Scattered device-scope atomics top out near 2e10 per second chip-wide, so one HBM atomic per row and state word caps
// the DIRECT kernel at ~1e10 rows/s whatever the bandwidth.  Instead the rows are first split by key hash into P
// partitions small enough that a partition's groups fit one workgroup's LDS table, then every partition is aggregated
// entirely in LDS by one workgroup and flushed once:
//   k_gb_hist[_wide]  per-workgroup histogram of partition ids over its contiguous row range      (key bytes read)
//   scan              exclusive scan of counts[P][G] -> exact, atomics-free write offsets
//   k_gb_scatter      per 12288-row tile (8192 / 4096 when LDS is short): LDS counting sort by partition, then coalesced run
//                     writes                                                                     (rows read, key + 8K B/row written)
//   k_gb_units        cuts partitions into work units (a hot key's partition is shared by many workgroups)
//   k_agg_part_lds    a workgroup aggregates a unit in an LDS open-addressing table and flushes it to the HBM table
// Partition buffers hold keys in 4 bytes (key types of <= 4 bytes) or 8, argument values widened to 8-byte words.
// More groups than 1024 partitions x half an LDS table: a first level with an independent hash cuts the rows into big
// partitions, each of which goes through the same passes again.
// ---------------------------------------------------------------------------------------------
#ifndef GBP_THREADS_V
#define GBP_THREADS_V 1024
#endif
#ifndef GBP_WG_PER_CU
#define GBP_WG_PER_CU 1
#endif
static constexpr u32 GBP_THREADS = GBP_THREADS_V;
static constexpr u32 GBP_MAX_P = 1024;
static constexpr u32 GBP_MAX_K = 2;

// Fibonacci hashing: one 64-bit multiply; the top bits pick the partition, bits 20.. pick the LDS cell (the full murmur
// finalizer cost ~20 VALU ops per row in three passes that turned out to be issue-bound, not HBM-bound)
static constexpr u64 GBP_MULT = 0x9E3779B97F4A7C15ull;  // partitions of the (second-level) pass that feeds the LDS aggregate
static constexpr u64 GBP_MULT1 = 0xC2B2AE3D27D4EB4Full; // first level of a two-level partitioning: an independent multiplier
__device__ __forceinline__ u64 gbp_mix(u64 key) { return key * GBP_MULT; }
__device__ __forceinline__ u32 gbp_part_of(u64 key, u32 pmask, u64 mult) { return (u32)((key * mult) >> 52) & pmask; }
// Keys of <= 4 bytes (the partition buffers hold them as u32) hash in 32 bits: one v_mul_lo_u32 instead of a 64-bit multiply
// (~5 VALU ops) in each of the three passes, which are issue-bound.  The TOP bits of key * odd pick the partition
// (mul_hi(h, P) = top log2 P bits) and the bits right below them the LDS cell: bit b of the product depends on key bits 0..b only,
// so low product bits must not be used (keys that differ in high bits only would share them).
template <typename KT>
__device__ __forceinline__ u32 gbp_part(KT key, u32 P, u64 mult)
{
    if constexpr (sizeof(KT) == 4)
        return __umulhi((u32)key * ((u32)(mult >> 32) | 1u), P);
    else
        return gbp_part_of((u64)key, P - 1, mult);
}
template <typename KT>
__device__ __forceinline__ u32 gbp_cell(KT key, u32 P, u32 S)
{
    if constexpr (sizeof(KT) == 4)
        return __umulhi((u32)key * ((u32)(GBP_MULT >> 32) | 1u) * P, S); // the partition's bits shifted out, the next log2 S bits
    else
        return (u32)(gbp_mix((u64)key) >> 20) & (S - 1); // bits disjoint from the partition id (>> 52)
}

template <typename KT>
struct GbpPartFn
{
    u32 P;
    u64 mult;
    __device__ __forceinline__ u32 operator()(KT key) const { return gbp_part<KT>(key, P, mult); }
};

struct GbpCols
{
    u32 k;                       // number of 8-byte argument words per row
    const void * src[GBP_MAX_K];
    int type[GBP_MAX_K];
    u64 * dst[GBP_MAX_K];
};

__global__ __launch_bounds__(GBP_THREADS) void k_gb_hist(const void * __restrict__ keys, int key_type, u64 row_begin, u64 n, u64 rows_per_wg,
                                                         u32 P, u32 * __restrict__ counts, u64 mult, int key32)
{
    auto part_of = [&](u64 k) -> u32 { return key32 ? gbp_part<u32>((u32)k, P, mult) : gbp_part<u64>(k, P, mult); };
    __shared__ u32 cnt[GBP_MAX_P];
    for (u32 p = threadIdx.x; p < P; p += GBP_THREADS)
        cnt[p] = 0;
    __syncthreads();
    const u64 r0 = (u64)blockIdx.x * rows_per_wg;
    const u64 r1 = r0 + rows_per_wg < n ? r0 + rows_per_wg : n;
    constexpr int HU = 8; // independent key loads in flight per lane
    u64 i = r0 + threadIdx.x;
    for (; i + (u64)(HU - 1) * GBP_THREADS < r1; i += (u64)HU * GBP_THREADS)
    {
        u64 k[HU];
#pragma unroll
        for (int q = 0; q < HU; ++q)
            k[q] = load_key_zext(keys, key_type, row_begin + i + (u64)q * GBP_THREADS);
#pragma unroll
        for (int q = 0; q < HU; ++q)
            atomicAdd(&cnt[part_of(k[q])], 1u);
    }
    for (; i < r1; i += GBP_THREADS)
        atomicAdd(&cnt[part_of(load_key_zext(keys, key_type, row_begin + i))], 1u);
    __syncthreads();
    for (u32 p = threadIdx.x; p < P; p += GBP_THREADS)
        counts[(u64)p * gridDim.x + blockIdx.x] = cnt[p];
}

// dynamic LDS: stage_word u64[K][TILE] | cursor u64[P] | delta u64[P] | stage_key KT[TILE] | tile_cnt u32[P] | tile_off u32[P]
// (the partition of a staged row is recomputed from its key in the write-out phase: one multiply instead of 2 B/row of LDS,
//  which buys a 12288-row tile -> 1.5x longer partition runs for the 4-byte-key, one-word shape)
// KT = u32 for key types of <= 4 bytes (the partition buffers then hold 4-byte keys: 12 instead of 16 B/row for C3), else u64
// WIDE: the key column is KT-wide, every argument is 8 bytes wide and the first row is 16-byte aligned in all of them;
// a thread then owns row PAIRS (tile row q*2*threads + 2*tid + {0,1}) and fetches each pair with one 8/16-byte
// nontemporal load; otherwise rows are strided by the workgroup size and loaded one by one through the type switches.
template <u32 GBP_TILE, typename KT, bool WIDE>
__global__ __launch_bounds__(GBP_THREADS) void k_gb_scatter(const void * __restrict__ keys, int key_type, u64 row_begin, u64 n, u64 rows_per_wg,
                                                            u32 P, const u64 * __restrict__ offsets, GbpCols cols, KT * __restrict__ out_keys, u64 mult, int gmajor = 0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
    u64 * stage_word = (u64 *)gb_lds;
    u64 * cursor = stage_word + (size_t)cols.k * GBP_TILE;
    u64 * delta = cursor + P; // cursor[p] - tile_off[p] of the current tile: destination row = delta[p] + position in the sorted tile
    KT * stage_key = (KT *)(delta + P);
    u32 * tile_cnt = (u32 *)(stage_key + GBP_TILE);
    u32 * tile_off = tile_cnt + P;
    __shared__ u32 wave_tot[GBP_THREADS / 64];

    for (u32 p = threadIdx.x; p < P; p += GBP_THREADS)
    {
        cursor[p] = offsets[gmajor ? (u64)blockIdx.x * P + p : (u64)p * gridDim.x + blockIdx.x];
        tile_cnt[p] = 0;
    }
    __syncthreads();
    const u64 r0 = (u64)blockIdx.x * rows_per_wg;
    const u64 r1 = r0 + rows_per_wg < n ? r0 + rows_per_wg : n;
    constexpr u32 RPT = GBP_TILE / GBP_THREADS; // rows per thread per tile
    // Register budget at 1024 threads is 128 VGPRs: keys are held in the buffer's key width, and the 12288-row tile -- which
    // only fits LDS with one argument word -- does not carry registers for a second one (it used to spill 9 dwords per lane)
    constexpr u32 KW = GBP_TILE >= 12288 ? 1 : GBP_MAX_K;
    KT key[RPT];
    u64 argw[RPT][KW];
    // rows of a tile are held in registers; the NEXT tile's loads are issued right after the current tile has been
    // staged to LDS, so their latency hides behind the write-out phase (one workgroup per CU: nothing else would)
    auto row_of = [&](u64 tb, u32 j) -> u64 {
        if constexpr (WIDE)
            return tb + (u64)(j >> 1) * (2 * GBP_THREADS) + 2 * threadIdx.x + (j & 1);
        else
            return tb + (u64)j * GBP_THREADS + threadIdx.x;
    };
    auto load_tile = [&](u64 tb) {
        if constexpr (WIDE)
        {
            typedef u64 v2q __attribute__((ext_vector_type(2)));
            typedef u32 v2d __attribute__((ext_vector_type(2)));
            static_assert(RPT % 2 == 0, "row pairs");
#pragma unroll
            for (u32 j = 0; j < RPT; j += 2)
            {
                const u64 i = row_of(tb, j);
                if (i + 1 < r1)
                {
                    if constexpr (sizeof(KT) == 4)
                    {
                        const v2d kk = __builtin_nontemporal_load((const v2d *)((const u32 *)keys + row_begin + i));
                        key[j] = kk.x, key[j + 1] = kk.y;
                    }
                    else
                    {
                        const v2q kk = __builtin_nontemporal_load((const v2q *)((const u64 *)keys + row_begin + i));
                        key[j] = kk.x, key[j + 1] = kk.y;
                    }
                    if (cols.k > 0)
                    {
                        const v2q a = __builtin_nontemporal_load((const v2q *)((const u64 *)cols.src[0] + row_begin + i));
                        argw[j][0] = a.x, argw[j + 1][0] = a.y;
                    }
                    if constexpr (KW > 1)
                        if (cols.k > 1)
                        {
                            const v2q a = __builtin_nontemporal_load((const v2q *)((const u64 *)cols.src[1] + row_begin + i));
                            argw[j][KW - 1] = a.x, argw[j + 1][KW - 1] = a.y;
                        }
                }
                else
                {
                    const bool in = i < r1; // at most the first row of the pair is left
                    key[j] = in ? ((const KT *)keys)[row_begin + i] : (KT)0;
                    argw[j][0] = (in && cols.k > 0) ? ((const u64 *)cols.src[0])[row_begin + i] : 0;
                    key[j + 1] = 0, argw[j + 1][0] = 0;
                    if constexpr (KW > 1)
                    {
                        argw[j][KW - 1] = (in && cols.k > 1) ? ((const u64 *)cols.src[1])[row_begin + i] : 0;
                        argw[j + 1][KW - 1] = 0;
                    }
                }
            }
        }
        else
        {
#pragma unroll
            for (u32 j = 0; j < RPT; ++j)
            {
                const u64 i = row_of(tb, j);
                const bool in = i < r1;
                key[j] = in ? (KT)load_key_zext(keys, key_type, row_begin + i) : (KT)0;
                argw[j][0] = (in && cols.k > 0) ? load_arg_bits(cols.src[0], cols.type[0], row_begin + i) : 0;
                if constexpr (KW > 1)
                    argw[j][KW - 1] = (in && cols.k > 1) ? load_arg_bits(cols.src[1], cols.type[1], row_begin + i) : 0;
            }
        }
    };
    if (r0 < r1)
        load_tile(r0);
    for (u64 tbase = r0; tbase < r1; tbase += GBP_TILE)
    {
        u32 part[RPT], rank[RPT];
        // 1. take a rank inside the tile's partition bucket
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
        {
            const u64 i = row_of(tbase, j);
            part[j] = ~0u;
            if (i < r1)
            {
                part[j] = gbp_part<KT>(key[j], P, mult);
                rank[j] = atomicAdd(&tile_cnt[part[j]], 1u);
            }
        }
        __syncthreads();
        // 2. exclusive scan of tile_cnt[P] -> tile_off[P]   (P <= 2 * threads)
        {
            const u32 e0 = threadIdx.x * 2, e1 = e0 + 1;
            const u32 c0 = e0 < P ? tile_cnt[e0] : 0, c1 = e1 < P ? tile_cnt[e1] : 0;
            u32 v = c0 + c1;
            const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            u32 inc = v;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1)
            {
                const u32 o = __shfl_up(inc, dlt, WAVE);
                if (lane >= (u32)dlt)
                    inc += o;
            }
            if (lane == 63)
                wave_tot[wave] = inc;
            __syncthreads();
            u32 base = inc - v;
            for (u32 w = 0; w < wave; ++w)
                base += wave_tot[w];
            // the same thread that scanned partition p also advances its cursor and clears its counter for the next tile:
            // step 1 of the next tile only starts after the barrier below, and nobody reads cursor[] again in this tile
            if (e0 < P)
            {
                tile_off[e0] = base;
                const u64 c = cursor[e0];
                delta[e0] = c - base;
                cursor[e0] = c + c0;
                tile_cnt[e0] = 0;
            }
            if (e1 < P)
            {
                tile_off[e1] = base + c0;
                const u64 c = cursor[e1];
                delta[e1] = c - (base + c0);
                cursor[e1] = c + c1;
                tile_cnt[e1] = 0;
            }
        }
        __syncthreads();
        // 3. counting sort into the LDS staging arrays
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
        {
            if (part[j] == ~0u)
                continue;
            const u32 pos = tile_off[part[j]] + rank[j];
            stage_key[pos] = (KT)key[j];
            if (cols.k > 0)
                stage_word[pos] = argw[j][0];
            if constexpr (KW > 1)
                if (cols.k > 1)
                    stage_word[(size_t)GBP_TILE + pos] = argw[j][KW - 1];
        }
        if (tbase + GBP_TILE < r1)
            load_tile(tbase + GBP_TILE); // prefetch: lands while this tile is written out
        __syncthreads();
        // 4. write the partition runs: consecutive lanes -> consecutive addresses inside a run
        const u32 tile_rows = (u32)(r1 - tbase < GBP_TILE ? r1 - tbase : GBP_TILE);
        for (u32 pos = threadIdx.x; pos < tile_rows; pos += GBP_THREADS)
        {
            const u32 p = gbp_part<KT>(stage_key[pos], P, mult);
            const u64 dst = delta[p] + pos;
            // plain stores: runs are 64-128 B, L2 write-combining completes the lines (nontemporal stores: 4.7 -> 8.1 ms)
            out_keys[dst] = stage_key[pos];
            for (u32 c = 0; c < cols.k; ++c)
                cols.dst[c][dst] = stage_word[(size_t)c * GBP_TILE + pos];
        }
        // no barrier here: the next tile's step 1 touches only tile_cnt[] (cleared in step 2 above), and its step 2 -- the
        // first writer of tile_off[]/delta[] -- sits behind the barrier that ends step 1, which every wave reaches only
        // after it has finished writing this tile out
    }
}

// Work units of the aggregate pass: partition p is cut into ceil(rows_p / chunk_rows) chunks so that a partition swollen
// by a hot key (Zipf) is shared by many workgroups instead of serialising the pass on one.  unit_start[p] = first unit of
// partition p, unit_start[P] = number of units; ctr is the dynamic work counter the workgroups draw units from.
__global__ __launch_bounds__(1024) void k_gb_units(const u64 * __restrict__ offsets, u32 G, u32 P, u64 n, u64 chunk_rows, u32 * __restrict__ unit_start, u32 * __restrict__ ctr)
{
    __shared__ u32 sc[1024];
    const u32 p = threadIdx.x;
    u32 c = 0;
    if (p < P)
    {
        const u64 begin = offsets[(u64)p * G];
        const u64 end = p + 1 < P ? offsets[(u64)(p + 1) * G] : n;
        c = (u32)((end - begin + chunk_rows - 1) / chunk_rows);
    }
    sc[p] = c;
    __syncthreads();
    for (u32 dlt = 1; dlt < 1024; dlt <<= 1)
    {
        const u32 o = p >= dlt ? sc[p - dlt] : 0;
        __syncthreads();
        sc[p] += o;
        __syncthreads();
    }
    if (p < P)
        unit_start[p] = sc[p] - c;
    if (p == P - 1)
        unit_start[P] = sc[p];
    if (p == 0)
        *ctr = 0;
}

// One workgroup aggregates whole partitions in LDS.  Partition p occupies rows [offsets[p*G], offsets[(p+1)*G]) of the
// partition buffers (n for the last).  Rows whose key cannot be placed in LDS go to the HBM table directly; rows that hit
// the max-fill limit there are marked pending (atomicOr: 64-row groups straddle partition boundaries).
//
// LDS cells are compact: the key array has the width of the partition buffer's keys (KT) and, when `cnt32` has bit w set,
// state word w is a row COUNT kept as 32 bits (a call never sees 2^32 rows; the host checks).  For the C3 shape
// (UInt32 key, sum, count) a cell is 4+8+4 = 16 B, so 8192 cells fit and 256 partitions suffice for 1 M groups --
// half as many partitions means partition runs twice as long in the scatter, whose cost is dominated by short runs.
// Layout: keys KT[S+1] (padded to 8 B) | every 8-byte word u64[S+1] in word order | every 4-byte word u32[S+1].
// widening of a zero-extended narrow argument load (see ex0/ex1 in k_agg_part_lds)
__device__ __forceinline__ u64 part_extend(u64 raw, int ex)
{
    switch (ex)
    {
        case 1: return (u64)(i64)(i8)(u8)raw;
        case 2: return (u64)(i64)(i16)(u16)raw;
        case 3: return (u64)(i64)(i32)(u32)raw;
        default: return (u64)__double_as_longlong((double)__uint_as_float((u32)raw));
    }
}

// does any state word of the compile-time update code use operation a or b?  (7: the low half of a fixed-point sum of argument word 0 --
// it counts as a user of that word with 1 and 3; 9: the high half, updated together with its low half)
__host__ __device__ constexpr bool gbp_ops_use(u32 ops, u32 a, u32 b)
{
    for (u32 w = 0; w < 8; ++w)
    {
        const u32 op = (ops >> (4 * w)) & 15u;
        if (op == a || op == b || (op == 7 && (a == 1 || b == 1 || a == 3 || b == 3)))
            return true;
    }
    return false;
}
// index of the (first) state word with operation `code`
__host__ __device__ constexpr u32 gbp_ops_find(u32 ops, u32 code)
{
    for (u32 w = 0; w < 8; ++w)
        if (((ops >> (4 * w)) & 15u) == code)
            return w;
    return 0;
}

struct PartLds
{
    u32 S1, cnt32, n8, keys_bytes;
    __device__ __forceinline__ u32 off(u32 w) const
    {
        const u32 low = (1u << w) - 1;
        if ((cnt32 >> w) & 1)
            return keys_bytes + 8 * S1 * n8 + 4 * S1 * (u32)__popc(cnt32 & low);
        return keys_bytes + 8 * S1 * (u32)__popc(~cnt32 & low);
    }
};

// AW: bytes per element of the argument columns (8 in PARTITION mode -- the buffers hold widened words; 8, 4 or 1 in RANGE mode,
// where the source columns are read as they are and 4-byte signed arguments are sign-extended after the load)
// KS: element type of the key column as stored (UInt8 keys are read as they are and held as KT = UInt32 in LDS)
// EXT: some argument word of this launch needs more than the zero extension its typed load gives (Int8/16/32 sign extension,
// Float32 -> Float64); compiled out otherwise -- the pass is issue-bound and the extension logic cost it 4-14 % when present
// OPS != 0: the state update is fixed at compile time -- 4 bits per state word, word 0 in the low nibble: 1 / 2 = integer sum of
// argument word 0 / 1, 3 / 4 = Float64 sum of argument word 0 / 1, 5 = row count kept in 32 bits, 6 = row count in 64 bits.  The
// run-time descriptor walk costs ~25 scalar + ~10 vector instructions per 64 rows of a pass that is bound by instruction issue.
template <typename KT, int AW, typename KS = KT, bool EXT = false, u32 OPS = 0>
__global__ __launch_bounds__(1024) void k_agg_part_lds(AggTable t, AggDesc d, const KS * __restrict__ keys, const void * __restrict__ words0, const void * __restrict__ words1,
                                                       const u64 * __restrict__ offsets, u32 G, u32 P, u64 n, u64 * __restrict__ pending, u32 S, u32 K, u32 cnt32,
                                                       u64 rows_per_chunk, const u32 * __restrict__ unit_start, u32 * __restrict__ unit_ctr,
                                                       const u8 * __restrict__ cond)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef typename std::conditional<sizeof(KT) == 4, unsigned int, unsigned long long>::type CasT;
    KT * lkeys = (KT *)lds_raw;
    PartLds L;
    L.S1 = S + 1;
    L.cnt32 = cnt32;
    L.n8 = d.n_words - (u32)__popc(cnt32);
    L.keys_bytes = ((u32)sizeof(KT) * L.S1 + 7) & ~7u;
    const u32 lds_bytes = L.keys_bytes + 8 * L.S1 * L.n8 + 4 * L.S1 * (u32)__popc(cnt32);
    __shared__ u32 lzero, sh_unit;
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const u64 gstride = t.capacity + 1;
    // The aggregate descriptors are decoded ONCE into wave-uniform registers (all loops over them are fully unrolled, so the
    // indices are constants): fetching d.a[j] inside the row loop meant an s_load per function and row group, and its
    // s_waitcnt lgkmcnt(0) also drains every LDS operation the wave has in flight -- the pass was issue-bound at ~170
    // clocks per 64 rows while the LDS itself can take this mix at 8.5 lanes/clock (tools/lds_bench.hip).
    //   op: 0 none, 1 add u64 (integer sum), 2 add f64, 3 count as u32, 4 count as u64
    u32 a_off[AGG_MAX_AGGS], a_off2[AGG_MAX_AGGS], a_off3[AGG_MAX_AGGS];
    int a_op[AGG_MAX_AGGS], a_op2[AGG_MAX_AGGS], a_src[AGG_MAX_AGGS];
    const int fx_base = d.fx_base;
    // what to do with the zero-extended load of argument word 0 / 1: 0 nothing, 1/2/3 sign-extend from 8/16/32 bits (Int8/16/32
    // columns), 4 Float32 -> Float64 bits
    int ex0 = 0, ex1 = 0;
#pragma unroll
    for (u32 j = 0; j < AGG_MAX_AGGS; ++j)
    {
        a_off[j] = a_off2[j] = a_off3[j] = 0;
        a_op[j] = a_op2[j] = a_src[j] = 0;
        if (OPS == 0 && j < d.n_aggs)
        {
            const u32 w = d.a[j].word;
            a_off[j] = L.off(w);
            if (d.a[j].kind == CHGPU_AGG_COUNT)
                a_op[j] = ((cnt32 >> w) & 1) ? 3 : 4;
            else
            {
                a_op[j] = (d.a[j].arg_type == CHGPU_F64 || d.a[j].arg_type == CHGPU_F32) ? 2 : 1;
                if ((d.word_fx >> w) & 1)
                {
                    a_op[j] = 5; // 128-bit fixed-point sum: the low half at a_off, the high half at a_off3
                    a_off3[j] = L.off(d.fx_hi[w]);
                }
                a_src[j] = (int)d.a[j].pre;
                const int ex = d.a[j].arg_type == CHGPU_I8 ? 1 : d.a[j].arg_type == CHGPU_I16 ? 2 : d.a[j].arg_type == CHGPU_I32 ? 3 : d.a[j].arg_type == CHGPU_F32 ? 4 : 0;
                (d.a[j].pre == 0 ? ex0 : ex1) = ex;
                if (d.a[j].kind == CHGPU_AGG_AVG)
                {
                    a_off2[j] = L.off(w + 1);
                    a_op2[j] = ((cnt32 >> (w + 1)) & 1) ? 3 : 4;
                }
            }
        }
    }
    u32 w_off[4];
#pragma unroll
    for (u32 w = 0; w < 4; ++w)
        w_off[w] = L.off(w);
    // PARTITION mode: work units (partition, chunk) are drawn from a device-wide counter until it passes the unit count
    // (every workgroup reaches that exit).  RANGE mode (offsets == nullptr): chunk blockIdx.x, +gridDim.x, ... of the
    // source columns themselves (keys/words0/words1 point at the block's first row) -- the low-cardinality GROUP BY runs this way.
    const u32 n_units = offsets ? unit_start[P] : P;
    for (u32 iter = 0;; ++iter)
    {
        u32 unit;
        if (offsets)
        {
            if (threadIdx.x == 0)
                sh_unit = atomicAdd(unit_ctr, 1u);
            __syncthreads();
            unit = sh_unit;
        }
        else
            unit = blockIdx.x + iter * gridDim.x;
        if (unit >= n_units)
            break;
        u64 begin, end;
        if (offsets)
        {
            u32 lo = 0, hi = P - 1; // largest p with unit_start[p] <= unit
            while (lo < hi)
            {
                const u32 mid = (lo + hi + 1) >> 1;
                if (unit_start[mid] <= unit)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            const u32 p = lo;
            const u64 pbegin = offsets[(u64)p * G];
            const u64 pend = p + 1 < P ? offsets[(u64)(p + 1) * G] : n;
            begin = pbegin + (u64)(unit - unit_start[p]) * rows_per_chunk;
            end = begin + rows_per_chunk < pend ? begin + rows_per_chunk : pend;
        }
        else
        {
            begin = (u64)unit * rows_per_chunk;
            end = begin + rows_per_chunk < n ? begin + rows_per_chunk : n;
        }
        for (u32 s = threadIdx.x; s < lds_bytes / 8 + 1; s += blockDim.x)
            ((u64 *)lds_raw)[s] = 0; // the host rounds the allocation up to 8 bytes past lds_bytes
        if (threadIdx.x == 0)
            lzero = 0;
        __syncthreads();
        const u64 g0 = begin / 64, g1 = (end + 63) / 64;
        constexpr int PR = 4;  // 64-row groups per wave iteration: all their loads are issued before LDS is touched
        constexpr u32 PPRE = 2; // argument words preloaded per row (GBP_MAX_K)
        // Loads are unconditional (row indices clamped to the buffer) and double-buffered in registers: the loads of the
        // wave's next PR groups are in flight while the current ones go through the LDS table.  No branch surrounds a
        // load, so the compiler can keep exact vmcnt waits instead of draining the queue at a control-flow join.
        // cond (RANGE mode only): the WHERE mask of a fused filter + GROUP BY; rows whose byte is 0 are skipped entirely, as if a
        // FilterTransform had removed them before the AggregatingTransform.
        auto load_set = [&](u64 gb, u64 (&kv)[PR], u64 (&av)[PR][PPRE], u32 (&cv)[PR]) {
#pragma unroll
            for (int q = 0; q < PR; ++q)
            {
                u64 i = (gb + q) * 64 + lane;
                i = i < n ? i : n - 1;
                cv[q] = cond ? (u32)__builtin_nontemporal_load(&cond[i]) : 1u;
                kv[q] = (u64)__builtin_nontemporal_load(&keys[i]);
                typedef typename std::conditional<AW == 8, u64, typename std::conditional<AW == 4, u32, typename std::conditional<AW == 2, u16, u8>::type>::type>::type AT;
                // (with a compile-time OPS the loads are unconditional or absent: a run-time `K > 0` puts a branch around each load)
                if constexpr (OPS != 0)
                {
                    av[q][0] = gbp_ops_use(OPS, 1, 3) ? (u64)__builtin_nontemporal_load((const AT *)words0 + i) : 0;
                    av[q][1] = gbp_ops_use(OPS, 2, 4) ? (u64)__builtin_nontemporal_load((const AT *)words1 + i) : 0;
                }
                else
                {
                    av[q][0] = K > 0 ? (u64)__builtin_nontemporal_load((const AT *)words0 + i) : 0;
                    av[q][1] = K > 1 ? (u64)__builtin_nontemporal_load((const AT *)words1 + i) : 0;
                }
            }
        };
        // (Combining the rows of a hot key in registers before the LDS atomic was tried for Zipf inputs: once partitions are
        //  cut into work units it gains nothing -- 15.2 ms without vs 14.9-15.7 ms with -- and costs the uniform case 3-10 %.)
        auto process_set = [&](u64 gb, const u64 (&keyv)[PR], const u64 (&argv)[PR][PPRE], const u32 (&cv)[PR]) {
#pragma unroll
            for (int q = 0; q < PR; ++q)
            {
                const u64 g = gb + q;
                if (g >= g1)
                    break;
                const u64 i = g * 64 + lane;
                bool failed = false;
                if (i >= begin && i < end && cv[q] != 0)
                {
                    const u64 key = keyv[q];
                    u64 b0 = argv[q][0], b1 = argv[q][1];
                    if constexpr (EXT)
                    {
                        if (ex0) // wave-uniform
                            b0 = part_extend(b0, ex0);
                        if (ex1)
                            b1 = part_extend(b1, ex1);
                    }
                    u32 ls = ~0u;
                    if (key == 0)
                    {
                        ls = S;
                        lzero = 1;
                    }
                    else
                    {
                        u32 s = gbp_cell<KT>((KT)key, offsets ? P : 1u, S); // bits disjoint from the partition id
#pragma unroll 1
                        for (int probe = 0; probe < 64; ++probe)
                        {
                            KT k = lkeys[s];
                            if (k == 0)
                                k = (KT)atomicCAS((CasT *)&lkeys[s], (CasT)0, (CasT)key), k = (k == 0) ? (KT)key : k;
                            if (k == (KT)key)
                            {
                                ls = s;
                                break;
                            }
                            s = (s + 1) & (S - 1);
                        }
                    }
                    if (ls != ~0u)
                    {
                        if constexpr (OPS != 0)
                        {
#pragma unroll
                            for (u32 w = 0; w < 4; ++w)
                            {
                                const u32 op = (OPS >> (4 * w)) & 15u;
                                if (op == 0)
                                    break;
                                unsigned char * wp = lds_raw + w_off[w];
                                if (op == 1 || op == 2)
                                    atomicAdd((unsigned long long *)wp + ls, (unsigned long long)(op == 1 ? b0 : b1));
                                else if (op == 7)
                                    lds_add_fx((u64 *)wp + ls, (u64 *)(lds_raw + w_off[gbp_ops_find(OPS, 9)]) + ls, fx_from_double(b0, fx_base));
                                else if (op == 9)
                                    continue;
                                else if (op == 3 || op == 4)
                                    atomicAdd((double *)wp + ls, __longlong_as_double((long long)(op == 3 ? b0 : b1)));
                                else if (op == 5)
                                    atomicAdd((unsigned int *)wp + ls, 1u);
                                else
                                    atomicAdd((unsigned long long *)wp + ls, 1ull);
                            }
                        }
                        else
#pragma unroll
                        for (u32 j = 0; j < AGG_MAX_AGGS; ++j)
                        {
                            if (a_op[j] == 0)
                                break;
                            unsigned char * w = lds_raw + a_off[j];
                            const u64 bits = a_src[j] == 0 ? b0 : b1;
                            if (a_op[j] == 1)
                                atomicAdd((unsigned long long *)w + ls, (unsigned long long)bits);
                            else if (a_op[j] == 2)
                                atomicAdd((double *)w + ls, __longlong_as_double((long long)bits));
                            else if (a_op[j] == 3)
                                atomicAdd((unsigned int *)w + ls, 1u);
                            else if (a_op[j] == 5)
                                lds_add_fx((u64 *)w + ls, (u64 *)(lds_raw + a_off3[j]) + ls, fx_from_double(bits, fx_base));
                            else
                                atomicAdd((unsigned long long *)w + ls, 1ull);
                            if (a_op2[j] == 3)
                                atomicAdd((unsigned int *)(lds_raw + a_off2[j]) + ls, 1u); // avg's denominator
                            else if (a_op2[j] == 4)
                                atomicAdd((unsigned long long *)(lds_raw + a_off2[j]) + ls, 1ull);
                        }
                    }
                    else
                    {
                        const u64 slot = table_emplace(t, key, true);
                        if (slot == ~0ull)
                            failed = true;
                        else
                            add_vals_global(t, d, slot, b0, b1, 1);
                    }
                }
                const u64 b = __ballot(failed);
                if (b != 0 && lane == 0)
                {
                    atomicOr((unsigned long long *)&pending[g], (unsigned long long)b);
                    t.ctrl->overflow = 1;
                }
            }
        };
        {
            const u64 step = (u64)n_waves * PR;
            u64 kA[PR], aA[PR][PPRE], kB[PR], aB[PR][PPRE];
            u32 cA[PR], cB[PR];
            u64 gb = g0 + (u64)wave * PR;
            load_set(gb, kA, aA, cA);
            for (; gb < g1; gb += 2 * step)
            {
                load_set(gb + step, kB, aB, cB);
                __builtin_amdgcn_sched_barrier(0);
                process_set(gb, kA, aA, cA);
                load_set(gb + 2 * step, kA, aA, cA);
                __builtin_amdgcn_sched_barrier(0);
                process_set(gb + step, kB, aB, cB);
            }
        }
        __syncthreads();
        for (u32 s = threadIdx.x; s <= S; s += blockDim.x)
        {
            const u64 key = (u64)lkeys[s];
            const bool occupied = (s == S) ? (lzero != 0) : (key != 0);
            if (!occupied)
                continue;
            const u64 slot = table_emplace(t, s == S ? 0 : key, false);
            if (slot == ~0ull)
            {
                t.ctrl->fatal = 1;
                continue;
            }
            for (u32 w = 0; w < d.n_words; ++w)
            {
                if ((d.word_fx_hi >> w) & 1)
                    continue; // flushed with its low half
                const unsigned char * wp = lds_raw + L.off(w);
                const u64 bits = ((cnt32 >> w) & 1) ? (u64)((const u32 *)wp)[s] : ((const u64 *)wp)[s];
                if ((d.word_fx >> w) & 1)
                {
                    const u32 wh = d.fx_hi[w];
                    const u64 hb = ((const u64 *)(lds_raw + L.off(wh)))[s];
                    if (bits | hb)
                        global_add_fx(t.words + (u64)d.word_map[w] * gstride + slot, t.words + (u64)d.word_map[wh] * gstride + slot, Fx128{bits, hb});
                    continue;
                }
                if (bits != 0)
                    global_add_word(t.words + (u64)d.word_map[w] * gstride + slot, bits, (d.word_is_f64 >> w) & 1);
            }
        }
        __syncthreads();
    }
}

// ---- the TILE-SORTED plan (radix_partition.h, k_rp_tilesort): units and the aggregate pass that gathers one run per tile ----
// A partition of r rows gets c = ceil(r / chunk_rows) units, unit (p, j) = the j-th of c equal shares of the TILES (a partition swollen
// by a hot key is long in every tile, so cutting by tiles cuts its rows evenly).  unit_list is ordered by (j, p): the workgroups draw
// units in that order, so at any time they work on the SAME stretch of tiles for different partitions -- the lines at the two ends of
// a run also hold the neighbouring partitions' rows and are then found in L2 / Infinity Cache by the neighbours instead of being
// fetched from HBM once per partition.  unit_list[u] = p | j << 16 | c << 40; unit_count[0] = number of units.
static constexpr u32 TILE_QUEUES = 8; // XCDs of an MI355X
__global__ __launch_bounds__(1024) void k_tile_units(const unsigned long long * __restrict__ part_total, u32 P, u64 chunk_rows, u32 n_tiles, u64 * __restrict__ unit_list,
                                                      u32 max_units, u32 * __restrict__ qstart, u32 * __restrict__ ctr)
{
    __shared__ u32 sc[1024], cc[1024];
    const u32 p = threadIdx.x;
    u32 c = 0;
    if (p < P)
    {
        const u64 c64 = (part_total[p] + chunk_rows - 1) / chunk_rows;
        c = (u32)(c64 < n_tiles ? c64 : n_tiles); // a unit is at least one tile
    }
    sc[p] = c;
    cc[p] = c;
    __syncthreads();
    for (u32 dlt = 1; dlt < 1024; dlt <<= 1)
    {
        const u32 o = p >= dlt ? sc[p - dlt] : 0;
        __syncthreads();
        sc[p] += o;
        __syncthreads();
    }
    // Eight queues, one per XCD: queue x holds the units of partitions [x * PX, (x + 1) * PX) in (j, p) order, so that the 32
    // workgroups of an XCD sweep the same tiles for 32 NEIGHBOURING partitions -- the shared boundary lines are then L2 hits, not
    // just Infinity Cache hits.  (the host sized the list for the bound sum ceil(r_p / chunk) <= n / chunk + P)
    const u32 PX = (P + TILE_QUEUES - 1) / TILE_QUEUES;
    const u32 U = sc[1023] < max_units ? sc[1023] : max_units;
    for (u32 u = threadIdx.x; u < U; u += 1024)
    {
        u32 lo = 0, hi = P - 1; // the partition whose units [sc[q] - cc[q], sc[q]) hold u
        while (lo < hi)
        {
            const u32 mid = (lo + hi) >> 1;
            if (sc[mid] > u)
                hi = mid;
            else
                lo = mid + 1;
        }
        const u32 q = lo, j = u - (sc[q] - cc[q]);
        const u32 x = q / PX, r0 = x * PX, r1 = r0 + PX < P ? r0 + PX : P;
        u32 pos = r0 ? sc[r0 - 1] : 0; // the queue's first unit: all units of the partitions before it
        for (u32 r = r0; r < r1; ++r) // units of the queue ordered before (j, q): every (j', .) with j' < j, and (j, q') with q' < q
            pos += (cc[r] < j ? cc[r] : j) + ((r < q && cc[r] > j) ? 1u : 0u);
        if (pos < max_units)
            unit_list[pos] = (u64)q | ((u64)j << 16) | ((u64)cc[q] << 40);
    }
    if (p <= TILE_QUEUES)
    {
        const u32 r0 = p * PX < P ? p * PX : P;
        const u32 first = r0 ? sc[r0 - 1] : 0;
        qstart[p] = first < U ? first : U;
    }
    if (p < TILE_QUEUES)
        ctr[p] = 0;
}

// The tile index transposed for the aggregate pass: run_index[p][t] = start | length << 16 of partition p's run in tile t, so that a
// wave reads the entries of 64 consecutive tiles as ONE 256-byte load (read straight from tile_index[t][p] every entry costs a
// 128-byte line of its own: one line in seven of the whole pass).  84 MB for 1e9 rows: ~30 us.
__global__ __launch_bounds__(256) void k_tile_index_transpose(const unsigned short * __restrict__ tile_index, u32 n_tiles, u32 P, u32 * __restrict__ run_index)
{
    __shared__ u32 sm[64][65];
    const u32 tb = blockIdx.x * 64, pb = blockIdx.y * 64;
    const u32 x = threadIdx.x & 63, y0 = threadIdx.x >> 6;
    for (u32 y = y0; y < 64; y += 4) // tile tb + y, partition pb + x: consecutive lanes -> consecutive u16 entries
    {
        const u32 t = tb + y, pp = pb + x;
        u32 v = 0;
        if (t < n_tiles && pp < P)
        {
            const u32 a = tile_index[(u64)t * (P + 1) + pp], b = tile_index[(u64)t * (P + 1) + pp + 1];
            v = a | ((b - a) << 16);
        }
        sm[y][x] = v;
    }
    __syncthreads();
    for (u32 y = y0; y < 64; y += 4) // partition pb + y, tile tb + x: consecutive lanes -> consecutive tiles
    {
        const u32 t = tb + x, pp = pb + y;
        if (t < n_tiles && pp < P)
            run_index[(u64)pp * n_tiles + t] = sm[x][y];
    }
}

// A block of 64 entries of run_index, lane l <- the wave's tile ordinal kb + l (kb a multiple of 64: 64 consecutive tiles starting at
// `first`), 0 beyond the unit's last tile t1.
__device__ __forceinline__ u32 tiles_load_block(const u32 * __restrict__ run_index_p, u32 first, u32 t1, u32 lane)
{
    const u32 tl = first + lane;
    const bool valid = tl < t1;
    const u32 v = run_index_p[valid ? tl : t1 - 1];
    return valid ? v : 0u;
}

// One workgroup aggregates a unit = (partition p, a range of tiles) in the same compact LDS table as k_agg_part_lds (PartLds), then
// flushes it into the HBM table.  Wave w of the workgroup takes the unit's tiles w, w + 16, w + 32, ...: the index entries (start,
// length of partition p's run) of its next 64 tiles are fetched by ONE vector load (lane l = the l-th of those tiles) into a register
// block, two blocks alternate; a step takes PR tiles: their entries come out of the block by v_readlane (wave-uniform, so the row
// addresses are scalar base + lane), their rows are loaded (lanes beyond the run's length idle: a uniform input gives runs of
// TILE / P = 48 rows) and the previous step's rows go through the LDS table meanwhile.  Runs longer than 64 rows (skewed keys) are
// finished by a plain loop after the pipelined one.
// tile_index: u16 [n_tiles][P + 1]; the two entries of (tile, p) are read as one unaligned 32-bit load.
// OPS: the compile-time state update code of k_agg_part_lds (one argument word: operations 1, 3, 5, 6).
// AOS: the sorted copy is one array of 12-byte {word, key} records (words0 = its base; k_rp_tilesort AOS), 4-byte keys only.
template <typename KT, u32 OPS, u32 TILE, bool AOS = false>
__global__ __launch_bounds__(1024) void k_agg_tiles_lds(AggTable t, AggDesc d, const KT * __restrict__ keys, const u64 * __restrict__ words0,
                                                        const u32 * __restrict__ run_index, u32 n_tiles, u32 P, u64 * __restrict__ pending, u32 S, u32 cnt32,
                                                        const u64 * __restrict__ unit_list, const u32 * __restrict__ qstart, u32 * __restrict__ qctr, int experiment)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef typename std::conditional<sizeof(KT) == 4, unsigned int, unsigned long long>::type CasT;
    KT * lkeys = (KT *)lds_raw;
    PartLds L;
    L.S1 = S + 1;
    L.cnt32 = cnt32;
    L.n8 = d.n_words - (u32)__popc(cnt32);
    L.keys_bytes = ((u32)sizeof(KT) * L.S1 + 7) & ~7u;
    const u32 lds_bytes = L.keys_bytes + 8 * L.S1 * L.n8 + 4 * L.S1 * (u32)__popc(cnt32);
    __shared__ u32 lzero, sh_unit;
    const u32 lane = threadIdx.x & 63;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), n_waves = blockDim.x >> 6;
    const u64 gstride = t.capacity + 1;
    const u64 last_row = (u64)n_tiles * TILE - 1;
    u32 w_off[4];
#pragma unroll
    for (u32 w = 0; w < 4; ++w)
        w_off[w] = L.off(w);
    // units are drawn from the queue of this workgroup's XCD first (HW_REG_XCC_ID only steers the choice: when a queue runs dry the
    // workgroup goes on with the next one, so every unit is taken whatever the placement)
    const u32 xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & (TILE_QUEUES - 1);
    u32 dry = 0; // (thread 0) queues found empty
    for (;;)
    {
        if (threadIdx.x == 0)
        {
            u32 got = ~0u;
            for (u32 a = 0; a < TILE_QUEUES && got == ~0u; ++a)
            {
                const u32 y = (xcc + a) & (TILE_QUEUES - 1);
                if ((dry >> y) & 1u)
                    continue;
                const u32 len = qstart[y + 1] - qstart[y];
                const u32 k = len ? atomicAdd(&qctr[y], 1u) : len;
                if (k < len)
                    got = qstart[y] + k;
                else
                    dry |= 1u << y;
            }
            sh_unit = got;
        }
        __syncthreads();
        const u32 unit = sh_unit;
        if (unit == ~0u)
            break; // (every workgroup reaches this exit: all queues are dry)
        const u64 ud = unit_list[unit];
        const u32 p = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(ud & 0xffffu)), j = (u32)__builtin_amdgcn_readfirstlane((int)(u32)((ud >> 16) & 0xffffffu)),
                  c_p = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(ud >> 40));
        const u32 t0 = (u32)((u64)n_tiles * j / c_p), t1 = (u32)((u64)n_tiles * (j + 1) / c_p);
        for (u32 s = threadIdx.x; s < lds_bytes / 8 + 1; s += blockDim.x)
            ((u64 *)lds_raw)[s] = 0; // the host rounds the allocation up to 8 bytes past lds_bytes
        if (threadIdx.x == 0)
            lzero = 0;
        __syncthreads();
        constexpr u32 PR = 8; // tiles (= runs of partition p) per wave and step
        constexpr u32 NS = 7; // 64-row slots the PR runs of a step are packed into: PR x 48 rows on average, 7 x 64 = 448 slots
        // This wave's tiles: blocks of 64 consecutive tiles, block w, w + n_waves, ... of the unit; ordinal k -> tile
        // t0 + ((k >> 6) * n_waves + wave) * 64 + (k & 63), beyond t1 = no tile.
        const u32 n_blocks = (t1 - t0 + 63) / 64;
        const u32 my_blocks = wave < n_blocks ? (n_blocks - wave + n_waves - 1) / n_waves : 0;
        const u32 ns = my_blocks * (64 / PR);
        const u32 * __restrict__ run_index_p = run_index + (u64)p * n_tiles;
        auto tile_of = [&](u32 k) -> u32 { return t0 + ((k >> 6) * n_waves + wave) * 64 + (k & 63u); };
        u32 blk0 = tiles_load_block(run_index_p, tile_of(0), t1, lane), blk1 = tiles_load_block(run_index_p, tile_of(64), t1, lane);
        // A step's PR runs are PACKED: virtual row v = slot * 64 + lane belongs to the run r with cs[r] <= v < cs[r + 1] (cs = running
        // sum of the run lengths, wave-uniform) and sits at row v + dl[r] of the sorted copy (dl[r] = tile * TILE + start - cs[r], modulo
        // 2^32: the host keeps this plan below 2^32 rows).  Runs of 48 +- 7 rows fill 86 % of the lanes instead of 75 %, and a single
        // long run borrows the slack of its neighbours; only a step with more than NS * 64 rows (skewed keys) takes the plain loop.
        auto row_of = [&](u32 v, const u32 (&cs)[PR + 1], const u32 (&dl)[PR]) -> u32 {
            // (a sum of masked steps, not a chain of selects: the compiler turns the chain into a look-up in a stack copy of dl[])
            u32 i = v + dl[0];
#pragma unroll
            for (u32 r = 1; r < PR; ++r)
                i += v >= cs[r] ? dl[r] - dl[r - 1] : 0u;
            return i;
        };
        // the LDS update of one row; the row is virtual row v of the step (PACKED_ = true_type) or row v of the sorted copy itself; its index is only needed when the LDS table is full
        auto update_row = [&](KT key, u64 b0, u32 v, auto packed, const u32 (&cs)[PR + 1], const u32 (&dl)[PR]) {
            u32 ls = ~0u;
            if (key == 0)
            {
                ls = S;
                lzero = 1;
            }
            else
            {
                u32 s = gbp_cell<KT>(key, P, S); // bits disjoint from the partition id
#pragma unroll 1
                for (int probe = 0; probe < 64; ++probe)
                {
                    KT k = lkeys[s];
                    if (k == 0)
                        k = (KT)atomicCAS((CasT *)&lkeys[s], (CasT)0, (CasT)key), k = (k == 0) ? key : k;
                    if (k == key)
                    {
                        ls = s;
                        break;
                    }
                    s = (s + 1) & (S - 1);
                }
            }
            if (ls != ~0u)
            {
#pragma unroll
                for (u32 w = 0; w < 4; ++w)
                {
                    const u32 op = (OPS >> (4 * w)) & 15u;
                    if (op == 0)
                        break;
                    unsigned char * wp = lds_raw + w_off[w];
                    if (op == 1)
                        atomicAdd((unsigned long long *)wp + ls, (unsigned long long)b0);
                    else if (op == 7)
                        lds_add_fx((u64 *)wp + ls, (u64 *)(lds_raw + w_off[gbp_ops_find(OPS, 9)]) + ls, fx_from_double(b0, d.fx_base));
                    else if (op == 9)
                        continue;
                    else if (op == 3)
                        atomicAdd((double *)wp + ls, __longlong_as_double((long long)b0));
                    else if (op == 5)
                        atomicAdd((unsigned int *)wp + ls, 1u);
                    else
                        atomicAdd((unsigned long long *)wp + ls, 1ull);
                }
            }
            else
            {
                // the LDS table is full around this key's cell (more groups than promised): the row is left to the finish rounds, which
                // send pending rows through the HBM table (kept out of this loop: the copies of the row update must stay small)
                const u64 i = decltype(packed)::value ? row_of(v, cs, dl) : v;
                atomicOr((unsigned long long *)&pending[i >> 6], 1ull << (i & 63));
                t.ctrl->overflow = 1;
            }
        };
        // step s: its PR index entries out of the blocks (crossing into a new block refills the other one), then its row loads
        auto fetch = [&](u32 s, u32 (&cs)[PR + 1], u32 (&dl)[PR], KT (&kv)[NS], u64 (&av)[NS]) {
            const u32 k0 = s * PR;
            // (the blocks are handled as values: a `cond ? blk1 : blk0` on the captured variables becomes a select of their ADDRESSES and
            //  pins the whole closure to scratch memory)
            u32 b0v = blk0, b1v = blk1;
            if ((k0 & 63u) == 0 && k0 != 0)
            {
                const u32 nb = tiles_load_block(run_index_p, tile_of(k0 + 64), t1, lane);
                const bool odd = ((k0 >> 6) & 1u) != 0;
                b0v = odd ? nb : b0v;
                b1v = odd ? b1v : nb;
                blk0 = b0v;
                blk1 = b1v;
            }
            const u32 cur = ((k0 >> 6) & 1u) ? b1v : b0v; // (PR divides 64: a step never straddles two blocks)
            u32 c = 0;
#pragma unroll
            for (u32 r = 0; r < PR; ++r)
            {
                const u32 k = k0 + r;
                const u32 ix = (u32)__builtin_amdgcn_readlane((int)cur, (int)(k & 63u));
                const u32 tl = tile_of(k);
                cs[r] = c;
                dl[r] = (tl < n_tiles ? tl : n_tiles - 1) * TILE + (ix & 0xffffu) - c;
                c += ix >> 16;
            }
            cs[PR] = c;
#pragma unroll
            for (u32 m = 0; m < NS; ++m)
            {
                u32 i = row_of(m * 64 + lane, cs, dl);
                i = i < (u32)last_row ? i : (u32)last_row; // (the lanes beyond the step's rows computed anything)
                if constexpr (AOS && sizeof(KT) == 4)
                {
                    // (the word as ONE 8-byte load from its 4-byte aligned place: combining two loaded halves is an operation on the
                    //  loaded registers, which the scheduler puts right behind the loads -- and the wave then waits for them there)
                    typedef u64 u64_a4 __attribute__((aligned(4)));
                    const u32 * r = (const u32 *)words0 + (u64)i * 3;
                    av[m] = gbp_ops_use(OPS, 1, 3) ? (u64)__builtin_nontemporal_load((const u64_a4 *)r) : 0;
                    kv[m] = (KT)__builtin_nontemporal_load(r + 2);
                }
                else if constexpr (AOS)
                {
                    typedef u64 v2q __attribute__((ext_vector_type(2)));
                    const v2q rec = __builtin_nontemporal_load((const v2q *)words0 + i); // {word, key}
                    av[m] = rec.x;
                    kv[m] = (KT)rec.y;
                }
                else
                {
                    kv[m] = __builtin_nontemporal_load(&keys[i]);
                    av[m] = gbp_ops_use(OPS, 1, 3) ? __builtin_nontemporal_load(&words0[i]) : 0;
                }
            }
        };
        auto process = [&](const u32 (&cs)[PR + 1], const u32 (&dl)[PR], const KT (&kv)[NS], const u64 (&av)[NS]) {
            const u32 total = cs[PR];
            if (CHGPU_EXPERIMENT_VALUE(experiment) == 1) // timing experiment: the gather alone
            {
                u64 acc = 0;
#pragma unroll
                for (u32 m = 0; m < NS; ++m)
                    acc += (u64)kv[m] ^ av[m];
                if (acc == 0x123456789abcdefull)
                    lzero = 1;
                return;
            }
#pragma unroll
            for (u32 m = 0; m < NS; ++m)
            {
                const u32 v = m * 64 + lane;
                if (v < total)
                    update_row(kv[m], av[m], v, std::true_type{}, cs, dl);
            }
            if (total > NS * 64) // (wave-uniform) skewed keys: the rest of the step's rows, unpipelined
#pragma unroll 1
                for (u32 v = NS * 64 + lane; v < total; v += 64)
                {
                    const u32 i = row_of(v, cs, dl);
                    if constexpr (AOS && sizeof(KT) == 4)
                    {
                        const u32 * r = (const u32 *)words0 + (u64)i * 3;
                        update_row((KT)r[2], gbp_ops_use(OPS, 1, 3) ? (u64)r[0] | ((u64)r[1] << 32) : 0, i, std::false_type{}, cs, dl);
                    }
                    else if constexpr (AOS)
                        update_row((KT)words0[2 * (u64)i + 1], words0[2 * (u64)i], i, std::false_type{}, cs, dl);
                    else
                        update_row(keys[i], gbp_ops_use(OPS, 1, 3) ? words0[i] : 0, i, std::false_type{}, cs, dl);
                }
        };
        {
            u32 csA[PR + 1], dlA[PR], csB[PR + 1], dlB[PR];
            KT kA[NS], kB[NS];
            u64 aA[NS], aB[NS];
            fetch(0, csA, dlA, kA, aA);
            for (u32 s = 0; s < ns; s += 2)
            {
                fetch(s + 1, csB, dlB, kB, aB);
                __builtin_amdgcn_sched_barrier(0);
                process(csA, dlA, kA, aA);
                fetch(s + 2, csA, dlA, kA, aA);
                __builtin_amdgcn_sched_barrier(0);
                process(csB, dlB, kB, aB);
            }
        }
        __syncthreads();
        for (u32 s = threadIdx.x; s <= S; s += blockDim.x)
        {
            const u64 key = (u64)lkeys[s];
            const bool occupied = (s == S) ? (lzero != 0) : (key != 0);
            if (!occupied)
                continue;
            const u64 slot = table_emplace(t, s == S ? 0 : key, false);
            if (slot == ~0ull)
            {
                t.ctrl->fatal = 1;
                continue;
            }
            for (u32 w = 0; w < d.n_words; ++w)
            {
                if ((d.word_fx_hi >> w) & 1)
                    continue; // flushed with its low half
                const unsigned char * wp = lds_raw + L.off(w);
                const u64 bits = ((cnt32 >> w) & 1) ? (u64)((const u32 *)wp)[s] : ((const u64 *)wp)[s];
                if ((d.word_fx >> w) & 1)
                {
                    const u32 wh = d.fx_hi[w];
                    const u64 hb = ((const u64 *)(lds_raw + L.off(wh)))[s];
                    if (bits | hb)
                        global_add_fx(t.words + (u64)d.word_map[w] * gstride + slot, t.words + (u64)d.word_map[wh] * gstride + slot, Fx128{bits, hb});
                    continue;
                }
                if (bits != 0)
                    global_add_word(t.words + (u64)d.word_map[w] * gstride + slot, bits, (d.word_is_f64 >> w) & 1);
            }
        }
        __syncthreads();
    }
}

// Merge (key, state words) tuples into the table: mergeToViaEmplace, also the rehash of a grown table.
// src_words[w] + i*1 ; src keys are u64; key==0 entries are skipped when skip_zero_keys (table arrays: empty cells),
// zero_slot_index: index in the source arrays of the out-of-line zero key (or ~0).
struct AggFxWords
{
    u32 word_fx, word_fx_hi, word_any;
    unsigned char fx_hi[AGG_MAX_WORDS];
};
template <int MODE>
__global__ __launch_bounds__(AGG_THREADS) void k_agg_tuples(AggTable t, u32 n_words, u32 word_is_f64, AggFxWords fx, const u64 * __restrict__ src_keys,
                                                            const u64 * __restrict__ src_words, u64 src_stride, u64 n, int skip_zero_keys,
                                                            u64 zero_slot_index, int soft_limit, u64 * __restrict__ pending)
{
    const u32 lane = threadIdx.x & 63;
    const u64 wave0 = ((u64)blockIdx.x * AGG_THREADS + threadIdx.x) >> 6;
    const u64 n_waves = ((u64)gridDim.x * AGG_THREADS) >> 6;
    const u64 n_groups64 = (n + 63) / 64;
    const u64 gstride = t.capacity + 1;
    u32 my_claims = 0; // without a soft limit nobody reads the counter mid-kernel: count in registers, add once per wave
    for (u64 g = wave0; g < n_groups64; g += n_waves)
    {
        const u64 i = g * 64 + lane;
        bool active = i < n;
        if (MODE == AGG_MODE_PENDING)
        {
            const u64 word = pending[g];
            if (word == 0)
                continue;
            active = active && ((word >> lane) & 1);
        }
        bool failed = false;
        if (active)
        {
            u64 key = src_keys[i];
            const bool is_zero_cell = (i == zero_slot_index);
            if (is_zero_cell)
                key = 0;
            if (!(skip_zero_keys && key == 0 && !is_zero_cell))
            {
                bool claimed;
                const u64 slot = table_emplace_impl(t, key, soft_limit != 0, claimed);
                if (soft_limit)
                    count_claim(t, claimed); // the only divergent caller: count_claim's ballot sees the lanes in this branch
                else
                    my_claims += claimed;
                if (slot == ~0ull)
                    failed = true;
                else
                    for (u32 w = 0; w < n_words; ++w)
                    {
                        if ((fx.word_fx_hi >> w) & 1)
                            continue; // merged with its low half
                        if ((fx.word_any >> w) & 1)
                        {
                            // any(): changeFirstTime (SingleValueData.cpp) -- a state that has a value keeps it; {claim, value} move together
                            const u64 claim = src_words[(u64)w * src_stride + i];
                            if (claim && atomicCAS((unsigned long long *)(t.words + (u64)w * gstride + slot), 0ull, (unsigned long long)claim) == 0ull)
                                t.words[(u64)(w + 1) * gstride + slot] = src_words[(u64)(w + 1) * src_stride + i];
                            ++w;
                            continue;
                        }
                        if ((fx.word_fx >> w) & 1)
                        {
                            const u32 wh = fx.fx_hi[w];
                            global_add_fx(t.words + (u64)w * gstride + slot, t.words + (u64)wh * gstride + slot,
                                          Fx128{src_words[(u64)w * src_stride + i], src_words[(u64)wh * src_stride + i]});
                            continue;
                        }
                        global_add_word(t.words + (u64)w * gstride + slot, src_words[(u64)w * src_stride + i],
                                        (int)((word_is_f64 >> w) & 1) | (int)(((word_is_f64 >> (16 + w)) & 1) << 1)); // upper half of the mask: max words
                    }
            }
        }
        const u64 b = __ballot(failed);
        if (pending && lane == 0)
            pending[g] = b;
        if (b != 0 && lane == 0)
            t.ctrl->overflow = 1;
    }
    if (!soft_limit)
    {
        u32 tot = my_claims;
#pragma unroll
        for (int dlt = 32; dlt >= 1; dlt >>= 1)
            tot += __shfl_xor(tot, dlt, WAVE);
        if (lane == 0 && tot)
            atomicAdd(&t.ctrl->stripe[agg_stripe()][0], (unsigned long long)tot);
    }
}

__global__ __launch_bounds__(256) void k_occupied_mask(const u64 * __restrict__ keys, u64 capacity, const AggCtrl * __restrict__ ctrl, u8 * __restrict__ mask)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i <= capacity; i += (u64)gridDim.x * 256)
        mask[i] = (i == capacity) ? (ctrl->has_zero != 0) : (keys[i] != 0);
}

__global__ __launch_bounds__(256) void k_fix_zero_key(u64 * __restrict__ keys, u64 capacity)
{
    // the zero-key cell's key word is never written by emplace; make it read as 0 for the exported key column
    if (blockIdx.x == 0 && threadIdx.x == 0)
        keys[capacity] = 0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_narrow_keys(const u64 * __restrict__ in, u64 n, T * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
        out[i] = (T)in[i];
}

// AvgFraction::divide (AggregateFunctionAvg.h:61-67): Float64(numerator) / denominator
__global__ __launch_bounds__(256) void k_avg_divide(const u64 * __restrict__ num, const u64 * __restrict__ den, u64 n, int num_type, double * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        double x;
        if (num_type == CHGPU_F64)
            x = __longlong_as_double((long long)num[i]);
        else if (num_type == CHGPU_I64)
            x = (double)(i64)num[i];
        else
            x = (double)num[i];
        out[i] = x / (double)den[i];
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static u64 pow2_ceil(u64 x)
{
    u64 p = 1;
    while (p < x)
        p <<= 1;
    return p;
}

static AggFxWords agg_fx_words(const chgpu_agg * a)
{
    AggFxWords f;
    f.word_fx = a->word_fx;
    f.word_fx_hi = a->word_fx_hi;
    f.word_any = a->word_any;
    memcpy(f.fx_hi, a->fx_hi, sizeof(f.fx_hi));
    return f;
}

static int agg_alloc_table(chgpu_agg * a, u64 capacity, AggTable * t, void ** mem, size_t * mem_class)
{
    const size_t cells = capacity + 1;
    const size_t bytes = cells * 8 * (1 + a->n_words) + AGG_HDR_BYTES;
    void * m = nullptr;
    CHGPU_TRY(chgpu_pool_alloc(a->ctx, bytes, &m, mem_class));
    hipError_t e = hipMemsetAsync(m, 0, bytes, a->ctx->stream); // HashTableAllocator zero-fills (HashTableAllocator.h:11)
    if (e != hipSuccess)
    {
        chgpu_pool_free(a->ctx, m, *mem_class);
        return chgpu_set_error(CHGPU_ERR_DEVICE, "memset: %s", hipGetErrorString(e));
    }
    t->ctrl = (AggCtrl *)m;
    t->keys = (u64 *)((char *)m + AGG_HDR_BYTES);
    t->words = t->keys + cells;
    t->capacity = capacity;
    t->max_fill = capacity / 2;
    *mem = m;
    return CHGPU_OK;
}

static int agg_read_ctrl(chgpu_agg * a, AggCtrl * out)
{
    CHGPU_TRY(chgpu_read_back(a->ctx, a->t.ctrl, out, sizeof(AggCtrl)));
    out->n_groups = 0;
    for (u32 s = 0; s < AGG_STRIPES; ++s)
        out->n_groups += out->stripe[s][0];
    a->n_groups = out->n_groups;
    CHGPU_REQUIRE(!out->fatal, CHGPU_ERR_LOGICAL, "aggregation table filled completely during a flush");
    return CHGPU_OK;
}

// resize (HashTable.h:504-560): new capacity per the reference's grower, rehash every occupied cell
static int agg_grow(chgpu_agg * a, u64 min_groups, bool has_zero)
{
    u64 cap = a->t.capacity;
    do
    {
        // HashTableGrowerWithPrecalculation::increaseSize (HashTable.h:303): degree += degree >= 23 ? 1 : 2
        cap = cap >= (1ull << 23) ? cap * 2 : cap * 4;
    } while (cap / 2 <= min_groups);
    AggTable nt;
    void * nmem = nullptr;
    size_t nclass = 0;
    CHGPU_TRY(agg_alloc_table(a, cap, &nt, &nmem, &nclass));
    // old cells [0, capacity) plus the out-of-line zero cell when it is set; no soft limit: the new table fits them all
    const u64 n = a->t.capacity + (has_zero ? 1 : 0);
    const u32 grid = chgpu_grid_for(a->ctx, n, AGG_THREADS, 8);
    hipLaunchKernelGGL(k_agg_tuples<AGG_MODE_ALL>, dim3(grid), dim3(AGG_THREADS), 0, a->ctx->stream, nt, a->n_words, a->word_is_f64, agg_fx_words(a),
                       a->t.keys, a->t.words, a->t.capacity + 1, n, 1, has_zero ? a->t.capacity : ~0ull, 0, (u64 *)nullptr);
    a->ctx->counters[6] += 1;
    a->ctx->counters[7] += 1;
    CHGPU_HIP(hipGetLastError());
    chgpu_pool_free(a->ctx, a->table_mem, a->table_class); // stream-ordered: later users queue behind the rehash
    a->table_mem = nmem;
    a->table_class = nclass;
    a->t = nt;
    return CHGPU_OK;
}

// min_cells: what the first strategy to touch the table wants it to hold (the partitioned path's flush slack): a table that
// does not exist yet is created that large at once instead of being created small and rehashed empty a moment later
static int agg_ensure_table(chgpu_agg * a, u64 min_cells = 0)
{
    if (a->table_mem)
        return CHGPU_OK;
    u64 cap = pow2_ceil(a->size_hint * 2);
    if (cap < AGG_MIN_CAPACITY)
        cap = AGG_MIN_CAPACITY;
    if (cap < min_cells)
        cap = pow2_ceil(min_cells);
    return agg_alloc_table(a, cap, &a->t, &a->table_mem, &a->table_class);
}

extern "C" int chgpu_agg_create(chgpu_ctx * ctx, int key_type, uint32_t n_aggs, const int * agg_kinds, const int * arg_types,
                                uint64_t size_hint, chgpu_agg ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && out && (agg_kinds || n_aggs == 0), CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(n_aggs <= AGG_MAX_AGGS, CHGPU_ERR_NOT_IMPLEMENTED, "more than %u aggregate functions: CPU path", AGG_MAX_AGGS);
    CHGPU_REQUIRE(key_type < 0 || (chgpu_type_is_int(key_type)),
                  CHGPU_ERR_NOT_IMPLEMENTED, "GROUP BY key type %d: CPU path", key_type);
    chgpu_agg * a = new chgpu_agg();
    a->ctx = ctx;
    a->key_type = key_type;
    a->n_aggs = n_aggs;
    a->size_hint = size_hint;
    u32 w = 0;
    for (u32 j = 0; j < n_aggs; ++j)
    {
        const int kind = agg_kinds[j];
        const int at = (kind == CHGPU_AGG_COUNT || !arg_types) ? CHGPU_U64 : arg_types[j];
        const bool any_value = kind == CHGPU_AGG_ANY;
        const bool extremum = kind == CHGPU_AGG_MIN || kind == CHGPU_AGG_MAX || any_value;
        if (kind != CHGPU_AGG_COUNT && kind != CHGPU_AGG_SUM && kind != CHGPU_AGG_AVG && !extremum)
        {
            delete a;
            return chgpu_set_error(CHGPU_ERR_NOT_IMPLEMENTED, "aggregate function kind %d has no device state: CPU path", kind);
        }
        if (kind != CHGPU_AGG_COUNT && !chgpu_type_size(at))
        {
            delete a;
            return chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "bad argument type %d", at);
        }
        a->kinds[j] = kind;
        a->arg_types[j] = at;
        a->word_off[j] = w;
        if (extremum)
        {
            a->word_is_f64 |= 1u << (16 + w); // upper half of the mask: the word combines by unsigned max (order keys), never by an add
            a->has_extremum = true;
            if (any_value)
                a->word_any |= 1u << w;
        }
        else if (kind != CHGPU_AGG_COUNT && chgpu_type_is_float(at))
            a->word_is_f64 |= 1u << w;
        w += (kind == CHGPU_AGG_AVG || any_value) ? 2 : 1;
    }
    if (w > AGG_MAX_WORDS)
    {
        delete a;
        return chgpu_set_error(CHGPU_ERR_NOT_IMPLEMENTED, "more than %u state words: CPU path", AGG_MAX_WORDS);
    }
    a->n_pub_words = w;
    if (key_type >= 0 && chgpu_opt(ctx, "deterministic_float_sums", 1))
    {
        u32 n_fx = 0;
        for (u32 j = 0; j < n_aggs; ++j)
            n_fx += ((a->word_is_f64 >> a->word_off[j]) & 1) ? 1 : 0;
        if (w + n_fx <= AGG_MAX_WORDS) // (more words than the masks hold: such an aggregation keeps its double states)
            for (u32 j = 0; j < n_aggs; ++j)
            {
                const u32 lo = a->word_off[j];
                if (!((a->word_is_f64 >> lo) & 1))
                    continue;
                a->word_is_f64 &= ~(1u << lo); // the low half combines by an integer add
                a->word_fx |= 1u << lo;
                a->word_fx_hi |= 1u << w;
                a->fx_hi[lo] = (unsigned char)w;
                ++w;
            }
    }
    a->n_words = w;
    memset(a->host_words, 0, sizeof(a->host_words));
    chgpu_ctx_retain(ctx);
    *out = a;
    return CHGPU_OK;
}

extern "C" int chgpu_agg_free(chgpu_agg * a)
{
    ChgpuDeviceGuard _dev_guard(a ? a->ctx : nullptr);
    if (!a)
        return CHGPU_OK;
    if (a->table_mem)
        chgpu_pool_free(a->ctx, a->table_mem, a->table_class);
    chgpu_ctx * ctx = a->ctx;
    delete a;
    chgpu_ctx_release(ctx);
    return CHGPU_OK;
}

static void agg_fill_desc(const chgpu_agg * a, const chgpu_col * const * arg_cols, AggDesc * d)
{
    d->n_aggs = a->n_aggs;
    d->n_words = a->n_words;
    d->word_is_f64 = a->word_is_f64;
    d->word_fx = a->word_fx;
    d->word_fx_hi = a->word_fx_hi;
    memcpy(d->fx_hi, a->fx_hi, sizeof(d->fx_hi));
    d->fx_base = a->fx_base;
    d->row_seq = 0;
    for (u32 w = 0; w < AGG_MAX_WORDS; ++w)
        d->word_map[w] = (unsigned char)w;
    for (u32 j = 0; j < a->n_aggs; ++j)
    {
        d->a[j].ptr = (arg_cols && arg_cols[j]) ? arg_cols[j]->data : nullptr;
        d->a[j].kind = a->kinds[j];
        d->a[j].arg_type = a->arg_types[j];
        d->a[j].word = a->word_off[j];
        d->a[j].pre = 0;
    }
}

static int agg_finish_rounds(chgpu_agg * a, const AggDesc & d, const void * keys, int key_type, u64 row_begin, u64 n, u64 * pending);

// min / max / any WITHOUT key (executeWithoutKeyImpl, Aggregator.cpp:1276-1321: addBatchSinglePlace): one order-key maximum over the rows of
// the block that pass `cond`; the first such row for any()
__global__ __launch_bounds__(256) void k_nokey_extremum(const void * __restrict__ p, int type, u64 row_begin, u64 n, const u8 * __restrict__ cond, int is_min,
                                                         unsigned long long * __restrict__ out)
{
    u64 best = 0;
    for (u64 r = (u64)blockIdx.x * 256 + threadIdx.x; r < n; r += (u64)gridDim.x * 256)
    {
        const u64 i = row_begin + r;
        if (cond && !cond[i])
            continue;
        const u64 k = agg_order_key(load_arg_bits(p, type, i), type);
        const u64 v = is_min ? ~k : k;
        best = v > best ? v : best;
    }
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1)
    {
        const u64 o = ((u64)(u32)__shfl_xor((int)(u32)(best >> 32), dlt, WAVE) << 32) | (u32)__shfl_xor((int)(u32)best, dlt, WAVE);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0 && best)
        atomicMax(out, (unsigned long long)best);
}
__global__ __launch_bounds__(256) void k_nokey_first_row(u64 row_begin, u64 n, const u8 * __restrict__ cond, unsigned long long * __restrict__ out)
{
    u64 first = ~0ull;
    for (u64 r = (u64)blockIdx.x * 256 + threadIdx.x; r < n && first == ~0ull; r += (u64)gridDim.x * 256)
        if (!cond || cond[row_begin + r])
            first = r;
    if (first != ~0ull)
        atomicMin(out, (unsigned long long)first);
}
__global__ void k_nokey_load(const void * __restrict__ p, int type, u64 i, u64 * __restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        out[0] = load_arg_bits(p, type, i);
}

// any(): the second pass of a block.  Every group's claim word now names the earliest of its rows (all blocks so far); the row a claim
// names stores its value.  Claims set by earlier blocks name rows of those blocks: no row of this block matches them, the value stays.
__global__ __launch_bounds__(AGG_THREADS) void k_agg_any_resolve(AggTable t, AggDesc d, const void * __restrict__ keys, int key_type, u64 row_begin, u64 n)
{
    const u64 stride = t.capacity + 1, mask = t.capacity - 1;
    for (u64 r = (u64)blockIdx.x * AGG_THREADS + threadIdx.x; r < n; r += (u64)gridDim.x * AGG_THREADS)
    {
        const u64 i = row_begin + r;
        const u64 key = load_key_zext(keys, key_type, i);
        u64 slot = t.capacity; // the zero key's cell
        if (key != 0)
        {
            slot = dev_intHash64(key) & mask;
            for (u64 step = 0; step < t.capacity; ++step)
            {
                const u64 k = t.keys[slot];
                if (k == key || k == 0)
                    break;
                slot = (slot + 1) & mask;
            }
            if (t.keys[slot] != key)
                continue; // (every row of the block was placed before this pass: not reached)
        }
        const u64 claim = ~(d.row_seq + i);
        for (u32 j = 0; j < d.n_aggs; ++j)
            if (d.a[j].kind == CHGPU_AGG_ANY && t.words[(u64)d.a[j].word * stride + slot] == claim)
                t.words[(u64)(d.a[j].word + 1) * stride + slot] = load_arg_bits(d.a[j].ptr, d.a[j].arg_type, i);
    }
}

// ---- the fixed-point window of the deterministic Float64 sums (see Fx128) ----
// Over the non-zero finite values (as doubles; a subnormal counts as exponent 1): out[0] = largest biased exponent (0 = no such value),
// out[2] = 2047 - smallest biased exponent; out[1] = 1 when some value is NaN / +-inf
__global__ __launch_bounds__(256) void k_fx_exp_stats(const void * __restrict__ p, int type, u64 row_begin, u64 n, u32 * __restrict__ out)
{
    u32 emax = 0, emin_c = 0, bad = 0;
    auto take = [&](u64 bits) {
        u32 e = (u32)(bits >> 52) & 0x7ffu;
        if (e == 0x7ffu)
            bad = 1;
        else if (bits << 1)
        {
            e = e ? e : 1u;
            emax = e > emax ? e : emax;
            emin_c = 2047u - e > emin_c ? 2047u - e : emin_c;
        }
    };
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, nthreads = (u64)gridDim.x * 256;
    if (type == CHGPU_F64 && (((uintptr_t)p + row_begin * 8) & 15) == 0)
    {
        // a streaming read: two 16-byte nontemporal loads per lane in flight
        typedef u64 v2q __attribute__((ext_vector_type(2)));
        const v2q * q = (const v2q *)((const u64 *)p + row_begin);
        const u64 pairs = n / 2;
        u64 i = tid;
        for (; i + nthreads < pairs; i += 2 * nthreads)
        {
            const v2q a = __builtin_nontemporal_load(q + i), b = __builtin_nontemporal_load(q + i + nthreads);
            take(a.x), take(a.y), take(b.x), take(b.y);
        }
        for (; i < pairs; i += nthreads)
        {
            const v2q a = __builtin_nontemporal_load(q + i);
            take(a.x), take(a.y);
        }
        if ((n & 1) && tid == 0)
            take(((const u64 *)p)[row_begin + n - 1]);
    }
    else
        for (u64 i = tid; i < n; i += nthreads)
            take(load_arg_bits(p, type, row_begin + i));
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1)
    {
        const u32 o = (u32)__shfl_xor((int)emax, dlt, WAVE), q2 = (u32)__shfl_xor((int)emin_c, dlt, WAVE);
        emax = o > emax ? o : emax;
        emin_c = q2 > emin_c ? q2 : emin_c;
    }
    const u64 anybad = __ballot(bad != 0);
    if ((threadIdx.x & 63) == 0)
    {
        if (emax)
        {
            atomicMax(&out[0], emax);
            atomicMax(&out[2], emin_c);
        }
        if (anybad)
            out[1] = 1;
    }
}
// every state of one pair shifted right by sh bits (arithmetic: floor), the window's unit growing from 2^base to 2^(base + sh)
__global__ __launch_bounds__(256) void k_fx_shift(u64 * __restrict__ lo, u64 * __restrict__ hi, u64 cells, int sh)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < cells; i += (u64)gridDim.x * 256)
    {
        const u64 l = lo[i], h = hi[i];
        if ((l | h) == 0)
            continue;
        u64 nl, nh;
        if (sh >= 128)
            nl = nh = (u64)((i64)h >> 63);
        else if (sh >= 64)
            nl = (u64)((i64)h >> (sh - 64 < 63 ? sh - 64 : 63)), nh = (u64)((i64)h >> 63);
        else
            nl = (l >> sh) | (h << (64 - sh)), nh = (u64)((i64)h >> sh);
        lo[i] = nl;
        hi[i] = nh;
    }
}
// out[i] = the pair as a double (out may be lo itself); hi is cleared when clear_hi
__global__ __launch_bounds__(256) void k_fx_to_double(u64 * __restrict__ lo, u64 * __restrict__ hi, u64 n, int base, int clear_hi)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        const double v = fx_to_double(lo[i], hi[i], base);
        lo[i] = (u64)__double_as_longlong(v);
        if (clear_hi)
            hi[i] = 0;
    }
}
__global__ __launch_bounds__(256) void k_fx_from_double(const u64 * __restrict__ src, u64 n, int base, u64 * __restrict__ lo, u64 * __restrict__ hi)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        const Fx128 x = fx_from_double(src[i], base);
        lo[i] = x.lo;
        hi[i] = x.hi;
    }
}

static int agg_fx_shift(chgpu_agg * a, int sh)
{
    if (sh <= 0 || !a->table_mem)
        return CHGPU_OK;
    const u64 cells = a->t.capacity + 1;
    for (u32 w = 0; w < a->n_pub_words; ++w)
        if ((a->word_fx >> w) & 1)
        {
            hipLaunchKernelGGL(k_fx_shift, dim3(chgpu_grid_for(a->ctx, cells, 256, 8)), dim3(256), 0, a->ctx->stream, a->t.words + (u64)w * cells,
                               a->t.words + (u64)a->fx_hi[w] * cells, cells, sh);
            a->ctx->counters[6] += 1;
        }
    CHGPU_HIP(hipGetLastError());
    return CHGPU_OK;
}
// Moves the window to (base, log_cap); never narrows it.
static int agg_fx_set_window(chgpu_agg * a, int base, int log_cap)
{
    if (!a->fx_base_set)
    {
        a->fx_base = base;
        a->fx_log_cap = log_cap;
        a->fx_base_set = true;
        return CHGPU_OK;
    }
    if (base > a->fx_base)
        CHGPU_TRY(agg_fx_shift(a, base - a->fx_base));
    a->fx_base = base > a->fx_base ? base : a->fx_base;
    a->fx_log_cap = log_cap > a->fx_log_cap ? log_cap : a->fx_log_cap;
    return CHGPU_OK;
}
// The aggregation goes back to double states (a NaN or an infinity cannot be a fixed-point value: from here on its sums behave like the
// reference's, poisoned groups included); every pair becomes the double it stands for.
static int agg_fx_to_plain(chgpu_agg * a)
{
    if (!a->word_fx)
        return CHGPU_OK;
    if (a->table_mem)
    {
        const u64 cells = a->t.capacity + 1;
        for (u32 w = 0; w < a->n_pub_words; ++w)
            if ((a->word_fx >> w) & 1)
            {
                hipLaunchKernelGGL(k_fx_to_double, dim3(chgpu_grid_for(a->ctx, cells, 256, 8)), dim3(256), 0, a->ctx->stream, a->t.words + (u64)w * cells,
                                   a->t.words + (u64)a->fx_hi[w] * cells, cells, a->fx_base, 1);
                a->ctx->counters[6] += 1;
            }
        CHGPU_HIP(hipGetLastError());
    }
    a->word_is_f64 |= a->word_fx;
    a->word_fx = 0; // (word_fx_hi stays: the spare words keep being skipped; they hold zeros)
    return CHGPU_OK;
}
// exponent statistics of `n` values of one column: *emax_biased = 0 when every value is zero (then *emin_biased is 2047)
static int agg_fx_stats(chgpu_ctx * ctx, const void * data, int type, u64 row_begin, u64 n, u32 * emax_biased, u32 * emin_biased, bool * nonfinite)
{
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, 256, &scratch));
    CHGPU_HIP(hipMemsetAsync(scratch, 0, 16, ctx->stream));
    hipLaunchKernelGGL(k_fx_exp_stats, dim3(chgpu_grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, data, type, row_begin, n, (u32 *)scratch);
    ctx->counters[6] += 1;
    CHGPU_HIP(hipGetLastError());
    u32 r[4];
    CHGPU_TRY(chgpu_read_back(ctx, scratch, r, 16));
    *emax_biased = r[0];
    *emin_biased = 2047u - r[2];
    *nonfinite = r[1] != 0;
    return CHGPU_OK;
}
// Every value must keep at least this many significant bits in the window; an input whose magnitudes spread further (2^(97 - 24) = 1e22
// between the largest and the smallest non-zero value, less 8 bits per widening for more than 2^30 rows) goes back to double states.
static constexpr int FX_MIN_BITS = 24;
// Makes room for `n` more values whose biased exponents span [emin, emax] (emax 0 = all of them zero).
static int agg_fx_admit(chgpu_agg * a, u32 emax_biased, u32 emin_biased, u64 n)
{
    if (emax_biased == 0)
    {
        a->fx_rows += n; // zeros fit any window
        return CHGPU_OK;
    }
    int log_cap = a->fx_base_set ? a->fx_log_cap : 30;
    const u64 rows = a->fx_rows + n;
    while (log_cap < 62 && rows > (1ull << log_cap))
        log_cap += 8;
    const int e_unb = (int)emax_biased - 1023;                        // every |x| < 2^(e_unb + 1)
    int base = e_unb + 1 - 127 + log_cap;                             // ... = 2^(127 - log_cap) units
    if (a->fx_base_set && a->fx_base + (log_cap - a->fx_log_cap) > base)
        base = a->fx_base + (log_cap - a->fx_log_cap);                // the states already there keep the invariant
    const int emin = (int)emin_biased - 1023 < a->fx_emin ? (int)emin_biased - 1023 : a->fx_emin;
    if (emin + 1 - FX_MIN_BITS < base) // the smallest value ever added would keep fewer than FX_MIN_BITS bits
        return agg_fx_to_plain(a);
    CHGPU_TRY(agg_fx_set_window(a, base, log_cap));
    a->fx_emin = emin;
    a->fx_rows = rows;
    return CHGPU_OK;
}
// before a block's rows are added: look at the float arguments of the fixed-point sums
static int agg_fx_prepare_block(chgpu_agg * a, const chgpu_col * const * arg_cols, u64 row_begin, u64 n)
{
    if (!a->word_fx || n == 0)
        return CHGPU_OK;
    u32 emax = 0, emin = 2047;
    for (u32 j = 0; j < a->n_aggs; ++j)
    {
        if (a->kinds[j] == CHGPU_AGG_COUNT || !((a->word_fx >> a->word_off[j]) & 1))
            continue;
        u32 e = 0, em = 2047;
        bool bad = false;
        CHGPU_TRY(agg_fx_stats(a->ctx, arg_cols[j]->data, a->arg_types[j], row_begin, n, &e, &em, &bad));
        if (bad)
            return agg_fx_to_plain(a);
        emax = e > emax ? e : emax;
        emin = em < emin ? em : emin;
    }
    return agg_fx_admit(a, emax, emin, n);
}

// Compact LDS cell of the partition-aggregate kernel for this aggregator's shape (see PartLds): bytes per cell and which
// state words are 32-bit counts.
static size_t agg_part_cell_bytes(const chgpu_agg * a, u64 n, u32 * cnt32_out)
{
    const bool key32 = chgpu_type_size(a->key_type) <= 4;
    u32 cnt32 = 0;
    const bool no_cnt32 = chgpu_opt(a->ctx, "tune_gb_nocnt32", 0) != 0;
    if (n < (1ull << 32) && !no_cnt32)
        for (u32 j = 0; j < a->n_aggs; ++j)
        {
            if (a->kinds[j] == CHGPU_AGG_COUNT)
                cnt32 |= 1u << a->word_off[j];
            else if (a->kinds[j] == CHGPU_AGG_AVG)
                cnt32 |= 1u << (a->word_off[j] + 1);
        }
    const u32 n4 = (u32)__builtin_popcount(cnt32), n8 = a->n_words - n4;
    if (cnt32_out)
        *cnt32_out = cnt32;
    return (key32 ? 4 : 8) + 8 * n8 + 4 * n4;
}

// Largest power-of-two cell count whose table fits ~150 KiB of LDS (at most 8192).
static u32 agg_part_max_cells(const chgpu_ctx * ctx, size_t cell_b)
{
    const u32 s_max = (u32)chgpu_opt(ctx, "tune_gb_s", 8192);
    const u32 s_kib = (u32)chgpu_opt(ctx, "tune_gb_kib", 150);
    u32 S = s_max;
    while ((size_t)(S + 1) * cell_b + 32 > (size_t)s_kib * 1024 && S > 256)
        S >>= 1;
    return S;
}

// Number of distinct keys D that makes a uniform sample of m rows show d distinct ones: d = D (1 - exp(-m / D)).
// (the reference adapts its strategy from observed statistics too: Aggregator.cpp:944-958, :83-89)
static u64 agg_estimate_groups(u64 d, u64 m)
{
    if (d == 0)
        return 0;
    if ((double)d >= 0.97 * (double)m)
        return ~0ull >> 8; // (nearly) every sampled row opened a group: no upper bound can be inferred
    double lo = (double)d, hi = 1e15;
    for (int it = 0; it < 200 && hi / lo > 1.0001; ++it)
    {
        const double mid = std::sqrt(lo * hi);
        const double seen = mid * (1.0 - std::exp(-(double)m / mid));
        if (seen < (double)d)
            lo = mid;
        else
            hi = mid;
    }
    return (u64)hi;
}

// The finish rounds of the tile-sorted plan over 12-byte records: the rows k_agg_tiles_lds left pending (LDS table full) go through the
// HBM table; a row that meets the max-fill limit stays pending for the next round (after the table has grown).
__global__ __launch_bounds__(AGG_THREADS) void k_agg_tiles_pending_aos(AggTable t, AggDesc d, const u32 * __restrict__ rec, int key64, u64 n, u64 * __restrict__ pending)
{
    const u32 lane = threadIdx.x & 63;
    const u64 wave0 = ((u64)blockIdx.x * AGG_THREADS + threadIdx.x) >> 6;
    const u64 n_waves = ((u64)gridDim.x * AGG_THREADS) >> 6;
    const u64 n_groups64 = (n + 63) / 64;
    for (u64 g = wave0; g < n_groups64; g += n_waves)
    {
        const u64 word = pending[g];
        if (word == 0)
            continue;
        const u64 i = g * 64 + lane;
        bool failed = false;
        if (i < n && ((word >> lane) & 1))
        {
            const u32 * r = rec + i * (key64 ? 4 : 3); // records are {word, key}: 12 bytes with a 4-byte key, 16 with an 8-byte key
            const u64 slot = table_emplace(t, key64 ? (u64)r[2] | ((u64)r[3] << 32) : (u64)r[2], true);
            if (slot == ~0ull)
                failed = true;
            else
                add_vals_global(t, d, slot, (u64)r[0] | ((u64)r[1] << 32), 0, 1);
        }
        const u64 b = __ballot(failed);
        if (lane == 0)
            pending[g] = b;
        if (b != 0 && lane == 0)
            t.ctrl->overflow = 1;
    }
}

static int agg_finish_rounds_aos(chgpu_agg * a, const AggDesc & d, const u32 * rec, int key64, u64 n, u64 * pending)
{
    chgpu_ctx * ctx = a->ctx;
    for (int round = 0; round < 64; ++round)
    {
        AggCtrl c;
        CHGPU_TRY(agg_read_ctrl(a, &c));
        if (!c.overflow && c.n_groups <= a->t.max_fill)
            return CHGPU_OK;
        CHGPU_TRY(agg_grow(a, c.n_groups, c.has_zero != 0));
        if (!c.overflow)
            return CHGPU_OK;
        const u32 grid = chgpu_grid_for(ctx, n, AGG_THREADS, 8);
        hipLaunchKernelGGL(k_agg_tiles_pending_aos, dim3(grid), dim3(AGG_THREADS), 0, ctx->stream, a->t, d, rec, key64, n, pending);
        ctx->counters[6] += 1;
        CHGPU_HIP(hipGetLastError());
    }
    return chgpu_set_error(CHGPU_ERR_LOGICAL, "aggregation table did not converge after 64 growth rounds");
}

// The TILE-SORTED plan of a partitioned executeOnBlock (k_rp_tilesort + k_agg_tiles_lds): one level, one 8-byte argument column
// (or none besides counts), 4- or 8-byte keys, a compile-time state update.  Two passes over the rows instead of three (no histogram),
// and the partition pass writes whole lines in row order.  NOT_IMPLEMENTED = the shape does not fit (the caller runs the scatter plan).
// The descriptor of a pass over a subset of the functions (d->a[0 .. n_aggs) already compacted to them): its state words renumbered
// 0 .. n-1 in the order of the functions (a fixed-point sum's high half behind the regular words, as in the aggregator), word_map[] back
// to the table's words, every per-word mask re-expressed in the local numbering; *cnt32 (which words are 32-bit counts in LDS) likewise.
static void agg_localise_desc(const chgpu_agg * a, AggDesc * d, u32 * cnt32)
{
    const u32 g_cnt32 = *cnt32, g_fx = d->word_fx, g_f64 = d->word_is_f64;
    unsigned char g_hi[AGG_MAX_WORDS];
    memcpy(g_hi, d->fx_hi, sizeof(g_hi));
    u32 wl = 0, l_cnt32 = 0, l_fx = 0, l_fx_hi = 0, l_f64 = 0;
    memset(d->fx_hi, 0, sizeof(d->fx_hi));
    u32 fx_lo_local[AGG_MAX_AGGS], n_fx = 0;
    unsigned char fx_hi_global[AGG_MAX_AGGS];
    for (u32 m = 0; m < d->n_aggs; ++m)
    {
        const u32 gw = d->a[m].word, words = (d->a[m].kind == CHGPU_AGG_AVG || d->a[m].kind == CHGPU_AGG_ANY) ? 2 : 1;
        d->a[m].word = wl;
        for (u32 x = 0; x < words; ++x)
        {
            d->word_map[wl + x] = (unsigned char)(gw + x);
            l_cnt32 |= ((g_cnt32 >> (gw + x)) & 1u) << (wl + x);
            l_f64 |= ((g_f64 >> (gw + x)) & 1u) << (wl + x);
            l_f64 |= ((g_f64 >> (16 + gw + x)) & 1u) << (16 + wl + x);
        }
        if ((g_fx >> gw) & 1)
        {
            l_fx |= 1u << wl;
            fx_lo_local[n_fx] = wl;
            fx_hi_global[n_fx++] = g_hi[gw]; // placed behind the regular words below
        }
        wl += words;
    }
    for (u32 k = 0; k < n_fx; ++k)
    {
        d->word_map[wl] = fx_hi_global[k];
        d->fx_hi[fx_lo_local[k]] = (unsigned char)wl;
        l_fx_hi |= 1u << wl;
        ++wl;
    }
    d->n_words = wl;
    d->word_fx = l_fx;
    d->word_fx_hi = l_fx_hi;
    d->word_is_f64 = l_f64;
    *cnt32 = l_cnt32;
    (void)a;
}
// bytes of the compact LDS cell that holds only the state words of the functions in `agg_mask` (see agg_part_cell_bytes)
static size_t agg_part_cell_bytes_masked(const chgpu_agg * a, u64 n, u32 agg_mask)
{
    const bool key32 = chgpu_type_size(a->key_type) <= 4;
    const bool c32 = n < (1ull << 32) && !chgpu_opt(a->ctx, "tune_gb_nocnt32", 0);
    size_t b = key32 ? 4 : 8;
    for (u32 j = 0; j < a->n_aggs; ++j)
    {
        if (!((agg_mask >> j) & 1))
            continue;
        if (a->kinds[j] == CHGPU_AGG_COUNT)
            b += c32 ? 4 : 8;
        else
        {
            b += 8;
            if ((a->word_fx >> a->word_off[j]) & 1)
                b += 8;
            if (a->kinds[j] == CHGPU_AGG_AVG)
                b += c32 ? 4 : 8;
        }
    }
    return b;
}

static int agg_add_block_tiled(chgpu_agg * a, const chgpu_col * key_col, const chgpu_col * const * arg_cols, u64 row_begin, u64 n, u32 P, u32 S, u32 cnt32,
                               u32 agg_mask, u64 chunk_rows, bool probe_only = false)
{
    chgpu_ctx * ctx = a->ctx;
    const size_t key_w = chgpu_type_size(a->key_type);
    if ((key_w != 4 && key_w != 8) || P > 512 || ((uintptr_t)key_col->data + row_begin * key_w) % 16 != 0)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    const bool key32 = key_w == 4;
    const u32 TILE = key32 ? 12288u : 8192u;
    if (n < (u64)TILE * ctx->num_cus || n + TILE >= (1ull << 32)) // (k_agg_tiles_lds indexes the sorted copy with 32 bits)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    // the one argument column
    int arg_j = -1;
    for (u32 j = 0; j < a->n_aggs; ++j)
        if (a->kinds[j] != CHGPU_AGG_COUNT && ((agg_mask >> j) & 1))
        {
            if (arg_j >= 0 && arg_cols[j]->data != arg_cols[arg_j]->data)
                return CHGPU_ERR_NOT_IMPLEMENTED;
            if (arg_j < 0)
                arg_j = (int)j;
        }
    // the argument column as it is: 8-byte integers / Float64, or UInt32 / Int32 / Float32 widened inside the partition pass
    const int arg_t = arg_j >= 0 ? a->arg_types[arg_j] : -1;
    const size_t arg_w = arg_j >= 0 ? chgpu_type_size(arg_t) : 0;
    const int arg_ex = arg_t == CHGPU_I32 ? 3 : arg_t == CHGPU_F32 ? 4 : 0;
    if (arg_j < 0 || (arg_w != 8 && arg_w != 4) || ((uintptr_t)arg_cols[arg_j]->data + row_begin * arg_w) % (2 * arg_w) != 0)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    AggDesc d;
    agg_fill_desc(a, arg_cols, &d);
    const u32 n_tiles = (u32)((n + TILE - 1) / TILE);
    const u64 n_pad = (u64)n_tiles * TILE;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    // (6 % of slack on the unit size: the partitions of a uniform input all get the same number of units, so that units of equal rank
    //  cover equal stretches of tiles -- see k_tile_units)
    chunk_rows += chunk_rows / 16;
    const u32 max_units = (u32)(n / chunk_rows + P);
    const size_t tot_b = al((size_t)P * 8), unit_b = al((size_t)(max_units + 2) * 8 + 128), pend_b = al((n_pad / 64 + 1) * 8),
                 idx_b = al((size_t)n_tiles * (P + 1) * 2 + 16), ridx_b = al((size_t)n_tiles * P * 4), keys_b = al((size_t)n_pad * key_w), words_b = al((size_t)n_pad * 8);
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, tot_b + unit_b + pend_b + idx_b + ridx_b + keys_b + words_b, &scratch));
    unsigned long long * part_total = (unsigned long long *)scratch;
    u64 * unit_list = (u64 *)((char *)scratch + tot_b);
    u32 * unit_qstart = (u32 *)(unit_list + max_units + 1); // [TILE_QUEUES + 1], then the queues' work counters [TILE_QUEUES]
    u32 * unit_ctr = unit_qstart + TILE_QUEUES + 1;
    u64 * pending = (u64 *)((char *)scratch + tot_b + unit_b);
    unsigned short * tidx = (unsigned short *)((char *)pending + pend_b);
    u32 * run_index = (u32 *)((char *)tidx + idx_b);
    void * pkeys = (char *)run_index + ridx_b;
    // the sorted copy as {word, key} records (one piece per run and tile for the gather instead of two); the records take the key region
    // and the word region together, and `pwords` is then the base of the record array
    const bool no_aos = chgpu_opt(ctx, "tune_gb_no_aos", 0) != 0;
    const bool aos = !no_aos;
    u64 * pwords = aos ? (u64 *)pkeys : (u64 *)((char *)pkeys + keys_b);
    // the aggregate pass reads the widened words of the sorted copy
    for (u32 j = 0; j < a->n_aggs; ++j)
        if (a->kinds[j] != CHGPU_AGG_COUNT && ((agg_mask >> j) & 1))
        {
            d.a[j].ptr = pwords;
            d.a[j].arg_type = chgpu_type_is_float(a->arg_types[j]) ? CHGPU_F64 : CHGPU_U64;
            d.a[j].pre = 0;
        }
    if (agg_mask != ~0u)
    {
        u32 m = 0;
        for (u32 j = 0; j < a->n_aggs; ++j)
            if ((agg_mask >> j) & 1)
                d.a[m++] = d.a[j];
        d.n_aggs = m;
        agg_localise_desc(a, &d, &cnt32); // the pass's own state words 0 .. n-1 (the caller sized S and P for exactly those)
    }
    // the compile-time update code (see k_agg_part_lds, OPS)
    u32 ops = 0;
    {
        u32 word_op[AGG_MAX_WORDS] = {0};
        bool ok = d.n_words <= 4;
        for (u32 j = 0; j < d.n_aggs && ok; ++j)
        {
            const u32 w = d.a[j].word;
            if (d.a[j].kind == CHGPU_AGG_COUNT)
                word_op[w] = ((cnt32 >> w) & 1) ? 5 : 6;
            else
            {
                word_op[w] = d.a[j].arg_type == CHGPU_F64 ? 3 : 1;
                if ((d.word_fx >> w) & 1)
                    word_op[w] = 7, word_op[d.fx_hi[w]] = 9;
                if (d.a[j].kind == CHGPU_AGG_AVG)
                    word_op[w + 1] = ((cnt32 >> (w + 1)) & 1) ? 5 : 6;
            }
        }
        for (u32 w = 0; w < d.n_words && ok; ++w)
        {
            ok = ok && word_op[w] != 0;
            ops |= word_op[w] << (4 * w);
        }
        if (!ok)
            ops = 0;
    }
    if (ops != 0x51 && ops != 0x15 && ops != 0x1 && ops != 0x53 && ops != 0x3 && ops != 0x61 && ops != 0x16 && ops != 0x97 && ops != 0x957 && ops != 0x967)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    if (probe_only)
        return CHGPU_OK; // the plan takes this shape (nothing was launched)
    const u32 G = (u32)ctx->num_cus;
    const u64 rows_per_wg = ((n + G - 1) / G + TILE - 1) / TILE * TILE;
    const int tiles_experiment = CHGPU_EXPERIMENT(ctx, "experiment_tiles"); // timing experiments only (wrong results): -DCHGPU_EXPERIMENTS builds
    const bool debug = chgpu_opt(ctx, "debug", 0) != 0;
    if (debug)
        fprintf(stderr, "chgpu: tile-sorted GROUP BY n=%llu hint=%llu S=%u P=%u tile=%u ops=0x%x\n", (unsigned long long)n, (unsigned long long)a->size_hint, S, P, TILE, ops);
    CHGPU_HIP(hipMemsetAsync(scratch, 0, tot_b + unit_b + pend_b, ctx->stream));
    int rc = CHGPU_OK;
    const size_t lds_sort = rp_tilesort_lds_bytes(TILE, P, key_w);
#define GB_TILESORT(TILE_, KT_, AT_, EX_, AOS_)                                                                                                  \
    do                                                                                                                                          \
    {                                                                                                                                           \
        auto kern = k_rp_tilesort<TILE_, KT_, GbpPartFn<KT_>, RP_THREADS, AT_, EX_, AOS_>;                                                       \
        rc = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort) == hipSuccess ? CHGPU_OK : CHGPU_ERR_DEVICE; \
        if (rc == CHGPU_OK)                                                                                                                     \
            hipLaunchKernelGGL(kern, dim3(G), dim3(RP_THREADS), lds_sort, ctx->stream, (const KT_ *)key_col->data + row_begin, (const AT_ *)arg_cols[arg_j]->data + row_begin, n, \
                               rows_per_wg, P, (KT_ *)pkeys, pwords, tidx, part_total, GbpPartFn<KT_>{P, GBP_MULT});                             \
    } while (0)
#define GB_TILESORT_ARG(TILE_, KT_, AOS_)                             \
    do                                                                \
    {                                                                 \
        if (arg_w == 8) GB_TILESORT(TILE_, KT_, u64, 0, AOS_);        \
        else if (arg_ex == 3) GB_TILESORT(TILE_, KT_, u32, 3, AOS_);  \
        else if (arg_ex == 4) GB_TILESORT(TILE_, KT_, u32, 4, AOS_);  \
        else GB_TILESORT(TILE_, KT_, u32, 0, AOS_);                   \
    } while (0)
    if (aos && key32)
        GB_TILESORT_ARG(12288, u32, true);
    else if (aos)
        GB_TILESORT_ARG(8192, u64, true);
    else if (key32)
        GB_TILESORT_ARG(12288, u32, false);
    else
        GB_TILESORT_ARG(8192, u64, false);
#undef GB_TILESORT_ARG
#undef GB_TILESORT
    if (rc == CHGPU_OK)
    {
        hipLaunchKernelGGL(k_tile_units, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned long long *)part_total, P, chunk_rows, n_tiles, unit_list, max_units, unit_qstart, unit_ctr);
        hipLaunchKernelGGL(k_tile_index_transpose, dim3((n_tiles + 63) / 64, (P + 63) / 64), dim3(256), 0, ctx->stream, (const unsigned short *)tidx, n_tiles, P, run_index);
        const u32 n4 = (u32)__builtin_popcount(cnt32), n8 = d.n_words - n4;
        const size_t keys_lds = ((size_t)key_w * (S + 1) + 7) & ~(size_t)7;
        const size_t lds_ag = keys_lds + (size_t)(S + 1) * (8 * n8 + 4 * n4) + 16;
#define GB_TILES(KT_, OPS_, TILE_, AOS_)                                                                                                              \
    do                                                                                                                                                \
    {                                                                                                                                                 \
        auto kern = k_agg_tiles_lds<KT_, OPS_, TILE_, AOS_>;                                                                                           \
        rc = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ag) == hipSuccess ? CHGPU_OK : CHGPU_ERR_DEVICE; \
        if (rc == CHGPU_OK)                                                                                                                           \
            hipLaunchKernelGGL(kern, dim3(G), dim3(1024), lds_ag, ctx->stream, a->t, d, (const KT_ *)pkeys, (const u64 *)pwords, (const u32 *)run_index, n_tiles, P, \
                               pending, S, cnt32, (const u64 *)unit_list, (const u32 *)unit_qstart, unit_ctr, tiles_experiment);                      \
    } while (0)
#define GB_TILES_OPS(KT_, TILE_, AOS_)                        \
    switch (ops)                                              \
    {                                                         \
        case 0x51: GB_TILES(KT_, 0x51, TILE_, AOS_); break;   \
        case 0x15: GB_TILES(KT_, 0x15, TILE_, AOS_); break;   \
        case 0x1: GB_TILES(KT_, 0x1, TILE_, AOS_); break;     \
        case 0x53: GB_TILES(KT_, 0x53, TILE_, AOS_); break;   \
        case 0x3: GB_TILES(KT_, 0x3, TILE_, AOS_); break;     \
        case 0x61: GB_TILES(KT_, 0x61, TILE_, AOS_); break;   \
        case 0x97: GB_TILES(KT_, 0x97, TILE_, AOS_); break;   \
        case 0x957: GB_TILES(KT_, 0x957, TILE_, AOS_); break; \
        case 0x967: GB_TILES(KT_, 0x967, TILE_, AOS_); break; \
        default: GB_TILES(KT_, 0x16, TILE_, AOS_); break;     \
    }
        if (aos && key32)
        {
            GB_TILES_OPS(u32, 12288, true)
        }
        else if (aos)
        {
            GB_TILES_OPS(u64, 8192, true)
        }
        else if (key32)
        {
            GB_TILES_OPS(u32, 12288, false)
        }
        else
        {
            GB_TILES_OPS(u64, 8192, false)
        }
#undef GB_TILES_OPS
#undef GB_TILES
    }
    ctx->counters[6] += 3;
    ctx->counters[5] += n;
    if (rc == CHGPU_OK && hipGetLastError() != hipSuccess)
        rc = CHGPU_ERR_DEVICE;
    if (rc == CHGPU_OK)
        rc = aos ? agg_finish_rounds_aos(a, d, (const u32 *)pwords, key32 ? 0 : 1, n_pad, pending) : agg_finish_rounds(a, d, pkeys, key32 ? CHGPU_U32 : CHGPU_U64, 0, n_pad, pending);
    else
    {
        (void)hipGetLastError();
        chgpu_set_error(rc, "tile-sorted aggregation launch failed");
    }
    return rc;
}

// PARTITIONED executeOnBlock (see the kernel block comment).  Returns NOT_IMPLEMENTED when the shape does not fit
// (the caller then uses the DIRECT kernel).
// level 0: called by add_block; may turn itself into level 1 (the first of two partitioning levels: partitions the rows
// into P1 big partitions and runs a level-2 call over each partition buffer slice); level 2 never recurses.
// agg_mask: the aggregate functions this call applies (bit j = function j); the caller splits more than GBP_MAX_K argument
// columns into several calls over the same rows, each partitioning the key column with its own two argument columns.
static int agg_add_block_partitioned(chgpu_agg * a, const chgpu_col * key_col, const chgpu_col * const * arg_cols, u64 row_begin, u64 n, u32 K,
                                     int level = 0, size_t scratch_off = 0, u32 agg_mask = ~0u, int word_pass = 0 /* 1: a per-word local pass, 2: its probe */)
{
    const bool probe_only = word_pass == 2;
    chgpu_ctx * ctx = a->ctx;
    // LDS table of the aggregate pass (one 1024-thread workgroup per CU): compact cells -- key as wide as the partition
    // buffer's keys, COUNT words as 32 bits while the call has fewer than 2^32 rows -- and as many cells as fit ~150 KiB
    const bool key32 = chgpu_type_size(a->key_type) <= 4; // 4-byte (or narrower) keys are stored as 4 bytes in the partition buffers
    u32 cnt32 = 0;
    size_t cell_b = agg_part_cell_bytes(a, n, &cnt32);
    const u32 n4 = (u32)__builtin_popcount(cnt32), n8 = a->n_words - n4;
    // a pass over ONE argument word of a subset of the functions goes through the tile-sorted plan with cells that hold only its words
    const bool local_pass = word_pass != 0 && level == 0 && K == 1 && agg_mask != ~0u;
    if (local_pass)
        cell_b = agg_part_cell_bytes_masked(a, n, agg_mask);
    const u32 S = agg_part_max_cells(a->ctx, cell_b);
    // partitions so that a partition's expected groups fill at most 70 % of the LDS table (fewer partitions = longer runs in
    // the scatter: an estimate of 1.25 M groups still gets 256 partitions)
    const u64 part_cap = (u64)S * 7 / 10;
    u64 want_p = (a->size_hint + part_cap - 1) / part_cap;
    u32 P = 64;
    while (P < (u32)ctx->num_cus && P < GBP_MAX_P) // the aggregate pass runs one workgroup per partition: give every CU one
        P <<= 1;
    while (P < want_p && P < GBP_MAX_P)
        P <<= 1;
    // More groups than P_max partitions x half an LDS table: TWO LEVELS.  Level 1 cuts the rows into P1 big partitions with an
    // independent hash (long runs: close to a copy), then every big partition -- already in the 4/8-byte key + 8-byte word
    // layout -- goes through this function again (level 2) with its share of the promised groups.
    u64 mult = GBP_MULT;
    u32 P1 = 0;
    // (want_p already allows LDS tables 70 % full: up to ~5.9 M groups one level is the faster plan, 16 vs 24 ms at 5 M)
    if (want_p > GBP_MAX_P && level == 0 && !chgpu_opt(ctx, "tune_gb_no_two_level", 0))
    {
        const u64 sub_groups = (u64)(GBP_MAX_P / 2) * (S / 2); // leaves the second level at half its partition budget
        for (P1 = 2; (u64)P1 * sub_groups < a->size_hint && P1 < 256; P1 <<= 1)
            ;
        if ((u64)P1 * sub_groups * 2 < a->size_hint || n / P1 < (1u << 20))
            return CHGPU_ERR_NOT_IMPLEMENTED; // beyond two levels, or partitions too small to be worth three passes each
        P = P1;
        mult = GBP_MULT1;
        level = 1;
    }
    else if (level == 0 && (u64)P * (S / 2) < a->size_hint / 4) // hopelessly more groups than P * S: partitioning would not localise them
        return CHGPU_ERR_NOT_IMPLEMENTED;
    // work units of the aggregate pass: half an average partition each, so a uniform input gives every workgroup two
    // units and a partition swollen by a hot key is spread over many workgroups; each unit flushes its LDS table once
    const u32 unit_div = (u32)chgpu_opt(ctx, "tune_gb_unitdiv", 2);
    u64 chunk_rows = (n / ((u64)P * unit_div) + 63) / 64 * 64;
    if (chunk_rows < 65536)
        chunk_rows = 65536;
    const u64 max_units = n / chunk_rows + P; // sum over partitions of ceil(rows_p / chunk_rows)
    // every unit's flush may claim up to S+1 cells without the max-fill check: keep all of them inside the slack
    if (level != 1) // a first partitioning level touches no table: its level-2 calls size it
        CHGPU_TRY(agg_ensure_table(a, 2 * (max_units * (S + 1) + a->n_groups) + 2));
    for (int guard = 0; level != 1 && guard < 16 && a->t.capacity / 2 < max_units * (S + 1) + a->n_groups; ++guard)
    {
        AggCtrl c0;
        CHGPU_TRY(agg_read_ctrl(a, &c0));
        CHGPU_TRY(agg_grow(a, c0.n_groups, c0.has_zero != 0));
    }
    // one level, one argument word: the tile-sorted plan (two passes, streaming writes) where its shape fits
    const bool no_tiled = chgpu_opt(ctx, "tune_gb_no_tiled", 0) != 0;
    if (level == 0 && K == 1 && !no_tiled)
    {
        const int rc_t = agg_add_block_tiled(a, key_col, arg_cols, row_begin, n, P, S, cnt32, agg_mask, chunk_rows, probe_only);
        if (rc_t != CHGPU_ERR_NOT_IMPLEMENTED || local_pass) // (a local pass was sized for the tile-sorted plan alone: the caller falls back as a whole)
            return rc_t;
    }
    if (probe_only)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    const int gmajor_x = CHGPU_EXPERIMENT(ctx, "experiment_gmajor") ? 1 : 0; // timing experiment only (-DCHGPU_EXPERIMENTS builds): the aggregate pass still reads p-major
    // (carried tails measured SLOWER than plain runs -- 8.5 / 7.3 vs 6.4 ms at C3 -- and wrote more, not fewer, bytes (PMC WRITE_SIZE 21 GB
    //  vs 13 GB): a partition's line is then written by two instructions a barrier apart; kept selectable for A/B runs)
    const int carry_mode = chgpu_opt(ctx, "tune_gb_carry", 0);
    // (two scatter workgroups per CU in carry mode 2: the histogram is cut into the same row ranges)
    const bool carry_shape = carry_mode && K == 1 && P <= 256 && n < (1ull << 32);
    // (experiment: several smaller scatter workgroups per CU so that one's rank / scan phases overlap another's loads and stores)
    const int wgs_per_cu_x = chgpu_opt(ctx, "tune_gb_scatter_wgs", 1);
    const bool multi_wg = !carry_mode && (wgs_per_cu_x == 2 || wgs_per_cu_x == 4) && K == 1 && key32 && P <= 256;
    const u32 G = (u32)ctx->num_cus * (multi_wg ? (u32)wgs_per_cu_x : carry_shape && carry_mode == 2 ? 2 : GBP_WG_PER_CU);
    u64 rows_per_wg = (n + G - 1) / G;
    // the scatter's LDS image is tile*(8*K + key bytes) + 24*P bytes and must stay under ~159 KiB (160 KiB per workgroup, 64 B static)
    const size_t row_lds = 8 * K + (key32 ? 4 : 8);
    const u32 tile_cap = (u32)chgpu_opt(ctx, "tune_gb_tile", 12288);
    u32 tile = 4096;
    for (u32 cand : {8192u, 12288u})
        if (cand <= tile_cap && cand * row_lds + (size_t)P * 24 + 64 <= 159 * 1024)
            tile = cand;
    if (multi_wg)
        tile = 12288 / (u32)wgs_per_cu_x;
    rows_per_wg = (rows_per_wg + tile - 1) / tile * tile;
    const bool debug = chgpu_opt(ctx, "debug", 0) != 0;
    if (debug)
        fprintf(stderr, "chgpu: partitioned GROUP BY level=%d n=%llu hint=%llu S=%u P=%u G=%u tile=%u\n", level, (unsigned long long)n, (unsigned long long)a->size_hint, S, P, G, tile);

    // partition buffers (8-byte keys + K 8-byte words per row) and bookkeeping
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const u64 m = (u64)P * G;
    const size_t cnt_b = al(m * 4), off_b = al(m * 8 + 8), tmp_b = chgpu_scan_tmp_bytes(m), pend_b = al(((n + 63) / 64) * 8 + 8) + al((GBP_MAX_P + 2) * 4);
    // The partition buffers live in the context's scratch arena, which is kept between calls: a fresh hipMalloc of
    // 16 GB costs ~0.4 s, fifteen times the kernels it would serve.
    const u64 wstride = n + RP_SCATTER_SLACK; // rows per argument-word array (k_rp_scatter parks out-of-range rows in the slack)
    const size_t keys_b = al((size_t)wstride * (key32 ? 4 : 8));
    const size_t part_b = keys_b + al((size_t)wstride * 8 * K);
    const size_t own_b = al(cnt_b + off_b + 256 + tmp_b + pend_b + part_b);
    // a level-1 call reserves the region of its level-2 calls up front (growing the arena later would move it): the same
    // row count at most, bookkeeping for the largest partition count
    const u64 m2 = (u64)GBP_MAX_P * G;
    const size_t sub_b = level == 1 ? al(al(m2 * 4) + al(m2 * 8 + 8) + 256 + chgpu_scan_tmp_bytes(m2) + pend_b + part_b) + 4096 : 0;
    void * scratch_base = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, scratch_off + own_b + sub_b, &scratch_base));
    void * scratch = (char *)scratch_base + scratch_off;
    u32 * counts = (u32 *)scratch;
    u64 * offsets = (u64 *)((char *)scratch + cnt_b);
    u64 * total_dev = (u64 *)((char *)scratch + cnt_b + off_b);
    void * tmp = (char *)scratch + cnt_b + off_b + 256;
    u64 * pending = (u64 *)((char *)scratch + cnt_b + off_b + 256 + tmp_b);
    u32 * unit_start = (u32 *)((char *)pending + al(((n + 63) / 64) * 8 + 8)); // [P + 1] then the work counter
    u32 * unit_ctr = unit_start + GBP_MAX_P + 1;
    void * pkeys = (char *)scratch + cnt_b + off_b + 256 + tmp_b + pend_b; // keys (4 or 8 B) | word0 | word1
    u64 * pwords = (u64 *)((char *)pkeys + keys_b);

    GbpCols gc;
    gc.k = K;
    AggDesc d;
    agg_fill_desc(a, arg_cols, &d);
    u32 kk = 0;
    for (u32 j = 0; j < a->n_aggs; ++j)
    {
        if (a->kinds[j] == CHGPU_AGG_COUNT || !((agg_mask >> j) & 1))
            continue;
        gc.src[kk] = arg_cols[j]->data;
        gc.type[kk] = a->arg_types[j];
        gc.dst[kk] = pwords + (u64)kk * wstride;
        // the aggregate pass reads widened 8-byte words: integers were sign/zero-extended, Float64 kept its bits
        d.a[j].ptr = gc.dst[kk];
        d.a[j].arg_type = chgpu_type_is_float(a->arg_types[j]) ? CHGPU_F64 : CHGPU_U64;
        d.a[j].pre = kk;
        ++kk;
    }
    if (agg_mask != ~0u)
    {
        // keep only this call's functions in the descriptor (their state word indices stay the aggregator's own)
        u32 m = 0;
        for (u32 j = 0; j < a->n_aggs; ++j)
            if ((agg_mask >> j) & 1)
                d.a[m++] = d.a[j];
        d.n_aggs = m;
    }

    // wide loads need key/argument columns whose element width is the buffer width and a 16-byte aligned first row
    const size_t key_w = chgpu_type_size(a->key_type);
    bool wide = (key_w == 4 || key_w == 8) && ((uintptr_t)key_col->data + row_begin * key_w) % 16 == 0;
    for (u32 c = 0; c < K; ++c)
        wide = wide && chgpu_type_size(gc.type[c]) == 8 && ((uintptr_t)gc.src[c] + row_begin * 8) % 16 == 0;
    const bool no_wide = chgpu_opt(ctx, "tune_gb_nowide", 0) != 0;
    wide = wide && !no_wide;
    if (wide && key_w == 4)
        hipLaunchKernelGGL((k_rp_hist_wide<u32, GbpPartFn<u32>>), dim3(G), dim3(RP_THREADS), 0, ctx->stream, (const u32 *)key_col->data + row_begin, n, rows_per_wg, P, counts, GbpPartFn<u32>{P, mult}, gmajor_x);
    else if (wide)
        hipLaunchKernelGGL((k_rp_hist_wide<u64, GbpPartFn<u64>>), dim3(G), dim3(RP_THREADS), 0, ctx->stream, (const u64 *)key_col->data + row_begin, n, rows_per_wg, P, counts, GbpPartFn<u64>{P, mult});
    else
        hipLaunchKernelGGL(k_gb_hist, dim3(G), dim3(GBP_THREADS), 0, ctx->stream, (const void *)key_col->data, a->key_type, row_begin, n, rows_per_wg, P, counts, mult, key32 ? 1 : 0);
    int rc = chgpu_scan_exclusive_u32_u64(ctx, counts, offsets, m, total_dev, tmp, tmp_b);
    if (rc == CHGPU_OK)
    {
        const size_t lds_sc = (size_t)tile * row_lds + (size_t)P * 24 + 64;
#define GB_SCATTER(TILE_, KT_) do { if (wide) GB_SCATTER_W(TILE_, KT_, true); else GB_SCATTER_W(TILE_, KT_, false); } while (0)
#define GB_SCATTER_W(TILE_, KT_, W_)                                                                                                            \
    do                                                                                                                                          \
    {                                                                                                                                           \
        auto kern = k_gb_scatter<TILE_, KT_, W_>;                                                                                               \
        rc = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc) == hipSuccess ? CHGPU_OK : CHGPU_ERR_DEVICE; \
        if (rc == CHGPU_OK)                                                                                                                     \
            hipLaunchKernelGGL(kern, dim3(G), dim3(GBP_THREADS), lds_sc, ctx->stream, (const void *)key_col->data, a->key_type, row_begin, n, rows_per_wg, P, \
                               (const u64 *)offsets, gc, (KT_ *)pkeys, mult, gmajor_x);                                                              \
    } while (0)
        // carried-tail scatter (radix_partition.h): one 8-byte word, wide loads, P <= 256, < 2^32 rows.  carry_mode 2 = 512 threads x
        // 4096-row tiles x 8-row pieces, two workgroups per CU; 1 = 1024 x 8192 x 16-row pieces, one per CU
        const bool old_scatter = chgpu_opt(ctx, "tune_gb_old_scatter", 0) != 0;
        if (!carry_mode && !old_scatter && wide && K == 1 && n + RP_SCATTER_SLACK < (1ull << 32) && P + 1 <= 2 * RP_THREADS)
        {
            // the branch-free scatter (radix_partition.h): one 8-byte word, wide loads
            if (multi_wg)
            {
#define GB_MULTI(TILE_, THR_)                                                                                                                   \
    do                                                                                                                                          \
    {                                                                                                                                           \
        auto kern = k_rp_scatter<TILE_, u32, true, GbpPartFn<u32>, THR_>;                                                                        \
        const size_t lds_b = rp_scatter_lds_bytes(TILE_, P, 4, true);                                                                            \
        rc = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b) == hipSuccess ? CHGPU_OK : CHGPU_ERR_DEVICE; \
        if (rc == CHGPU_OK)                                                                                                                     \
            hipLaunchKernelGGL(kern, dim3(G), dim3(THR_), lds_b, ctx->stream, (const u32 *)key_col->data + row_begin, (const u64 *)gc.src[0] + row_begin, n, rows_per_wg, P, \
                               (const u64 *)offsets, (u32 *)pkeys, gc.dst[0], GbpPartFn<u32>{P, mult});                                          \
    } while (0)
                if (wgs_per_cu_x == 2) GB_MULTI(6144, 512); else GB_MULTI(3072, 256);
#undef GB_MULTI
            }
            else if (key32 && rp_scatter_lds_bytes(12288, P, 4, true) <= 159 * 1024)
            {
                auto kern = k_rp_scatter<12288, u32, true, GbpPartFn<u32>>;
                const size_t lds_b = rp_scatter_lds_bytes(12288, P, 4, true);
                rc = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b) == hipSuccess ? CHGPU_OK : CHGPU_ERR_DEVICE;
                if (rc == CHGPU_OK)
                    hipLaunchKernelGGL(kern, dim3(G), dim3(RP_THREADS), lds_b, ctx->stream, (const u32 *)key_col->data + row_begin, (const u64 *)gc.src[0] + row_begin, n, rows_per_wg, P,
                                       (const u64 *)offsets, (u32 *)pkeys, gc.dst[0], GbpPartFn<u32>{P, mult});
            }
            else if (key32)
            {
                auto kern = k_rp_scatter<8192, u32, true, GbpPartFn<u32>>;
                const size_t lds_b = rp_scatter_lds_bytes(8192, P, 4, true);
                rc = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b) == hipSuccess ? CHGPU_OK : CHGPU_ERR_DEVICE;
                if (rc == CHGPU_OK)
                    hipLaunchKernelGGL(kern, dim3(G), dim3(RP_THREADS), lds_b, ctx->stream, (const u32 *)key_col->data + row_begin, (const u64 *)gc.src[0] + row_begin, n, rows_per_wg, P,
                                       (const u64 *)offsets, (u32 *)pkeys, gc.dst[0], GbpPartFn<u32>{P, mult});
            }
            else
            {
                auto kern = k_rp_scatter<8192, u64, true, GbpPartFn<u64>>;
                const size_t lds_b = rp_scatter_lds_bytes(8192, P, 8, true);
                rc = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b) == hipSuccess ? CHGPU_OK : CHGPU_ERR_DEVICE;
                if (rc == CHGPU_OK)
                    hipLaunchKernelGGL(kern, dim3(G), dim3(RP_THREADS), lds_b, ctx->stream, (const u64 *)key_col->data + row_begin, (const u64 *)gc.src[0] + row_begin, n, rows_per_wg, P,
                                       (const u64 *)offsets, (u64 *)pkeys, gc.dst[0], GbpPartFn<u64>{P, mult});
            }
        }
        else if (carry_mode && wide && K == 1 && P <= 256 && n < (1ull << 32))
        {
#define GB_CARRY(KT_, TILE_, THR_, CG_)                                                                                                          \
    do                                                                                                                                          \
    {                                                                                                                                           \
        auto kern = k_rp_scatter_carry<TILE_, KT_, true, GbpPartFn<KT_>, THR_, CG_>;                                                             \
        const size_t lds_cy = rp_scatter_carry_lds_bytes(TILE_, P, CG_, sizeof(KT_), true);                                                      \
        rc = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cy) == hipSuccess ? CHGPU_OK : CHGPU_ERR_DEVICE; \
        if (rc == CHGPU_OK)                                                                                                                     \
            hipLaunchKernelGGL(kern, dim3(G), dim3(THR_), lds_cy, ctx->stream, (const KT_ *)key_col->data + row_begin, (const u64 *)gc.src[0] + row_begin, n, rows_per_wg, P, \
                               (const u64 *)offsets, (KT_ *)pkeys, gc.dst[0], GbpPartFn<KT_>{P, mult});                                          \
    } while (0)
            if (carry_mode == 2) { if (key32) GB_CARRY(u32, 4096, 512, 8); else GB_CARRY(u64, 2048, 512, 8); }
            else                 { if (key32) GB_CARRY(u32, 8192, 1024, 16); else GB_CARRY(u64, 4096, 1024, 16); }
#undef GB_CARRY
        }
        else if (tile == 12288) { if (key32) GB_SCATTER(12288, u32); else GB_SCATTER(12288, u64); }
        else if (tile == 8192) { if (key32) GB_SCATTER(8192, u32); else GB_SCATTER(8192, u64); }
        else                   { if (key32) GB_SCATTER(4096, u32); else GB_SCATTER(4096, u64); }
#undef GB_SCATTER
#undef GB_SCATTER_W
    }
    if (level == 1)
    {
        ctx->counters[6] += 2;
        if (rc == CHGPU_OK && hipGetLastError() != hipSuccess)
            rc = CHGPU_ERR_DEVICE;
        if (rc != CHGPU_OK)
        {
            (void)hipGetLastError();
            return chgpu_set_error(rc, "partitioned aggregation launch failed");
        }
        // partition boundaries: offsets[p * G] for p = 0..P1-1 (the read-back also orders the host behind the scatter)
        std::vector<u64> starts(P1 + 1);
        {
            void * stage = nullptr;
            CHGPU_TRY(chgpu_pinned(ctx, (size_t)P1 * 8, &stage));
            CHGPU_HIP(hipMemcpy2DAsync(stage, 8, offsets, (size_t)G * 8, 8, P1, hipMemcpyDeviceToHost, ctx->stream));
            CHGPU_HIP(hipStreamSynchronize(ctx->stream));
            memcpy(starts.data(), stage, (size_t)P1 * 8);
            starts[P1] = n;
        }
        // the partition buffers as columns: keys of the buffer width, arguments widened to 8 bytes (Float64 kept its bits)
        chgpu_col kc{};
        kc.ctx = ctx;
        kc.type = key32 ? CHGPU_U32 : CHGPU_U64;
        kc.rows = n;
        kc.data = pkeys;
        chgpu_col ac[GBP_MAX_K]{};
        const chgpu_col * sub_args[AGG_MAX_AGGS] = {};
        const int saved_key_type = a->key_type;
        int saved_arg_types[AGG_MAX_AGGS];
        const u64 saved_hint = a->size_hint;
        u32 c = 0;
        for (u32 j = 0; j < a->n_aggs; ++j)
        {
            saved_arg_types[j] = a->arg_types[j];
            if (a->kinds[j] == CHGPU_AGG_COUNT || !((agg_mask >> j) & 1))
                continue;
            ac[c].ctx = ctx;
            ac[c].type = chgpu_type_is_float(a->arg_types[j]) ? CHGPU_F64 : CHGPU_U64; // two's complement sums: width is what matters
            ac[c].rows = n;
            ac[c].data = pwords + (u64)c * wstride;
            a->arg_types[j] = ac[c].type;
            sub_args[j] = &ac[c];
            ++c;
        }
        a->key_type = kc.type;
        a->size_hint = saved_hint / P1 + saved_hint / P1 / 4 + 1024;
        for (u32 q = 0; q < P1 && rc == CHGPU_OK; ++q)
            if (starts[q + 1] > starts[q])
                rc = agg_add_block_partitioned(a, &kc, sub_args, starts[q], starts[q + 1] - starts[q], K, 2, scratch_off + own_b, agg_mask);
        a->key_type = saved_key_type;
        a->size_hint = saved_hint;
        for (u32 j = 0; j < a->n_aggs; ++j)
            a->arg_types[j] = saved_arg_types[j];
        return rc;
    }
    if (rc == CHGPU_OK)
        rc = hipMemsetAsync(pending, 0, pend_b, ctx->stream) == hipSuccess ? CHGPU_OK : CHGPU_ERR_DEVICE;
    if (rc == CHGPU_OK)
    {
        const size_t keys_lds = ((size_t)(key32 ? 4 : 8) * (S + 1) + 7) & ~(size_t)7;
        const size_t lds_ag = keys_lds + (size_t)(S + 1) * (8 * n8 + 4 * n4) + 16; // the kernel zeroes whole 8-byte words
        hipLaunchKernelGGL(k_gb_units, dim3(1), dim3(1024), 0, ctx->stream, (const u64 *)offsets, G, P, n, chunk_rows, unit_start, unit_ctr);
        const u64 rows_per_chunk = chunk_rows;
        u32 grid = (u32)ctx->num_cus;
        // the update of the state words as a compile-time code where the common shapes allow it (see k_agg_part_lds, OPS)
        u32 ops = 0;
        const bool no_ops = chgpu_opt(ctx, "tune_gb_noops", 0) != 0;
        {
            u32 word_op[AGG_MAX_WORDS] = {0};
            bool ok = !no_ops && a->n_words <= 4;
            for (u32 j = 0; j < d.n_aggs && ok; ++j)
            {
                const u32 w = d.a[j].word;
                const bool c32 = (cnt32 >> w) & 1;
                if (d.a[j].kind == CHGPU_AGG_COUNT)
                    word_op[w] = c32 ? 5 : 6;
                else
                {
                    const bool f = d.a[j].arg_type == CHGPU_F64;
                    ok = ok && d.a[j].pre < 2;
                    word_op[w] = (f ? 3 : 1) + d.a[j].pre;
                    if ((a->word_fx >> w) & 1)
                    {
                        ok = ok && d.a[j].pre == 0; // (a fixed-point sum of the second argument word: generic kernel)
                        word_op[w] = 7, word_op[a->fx_hi[w]] = 9;
                    }
                    if (d.a[j].kind == CHGPU_AGG_AVG)
                        word_op[w + 1] = ((cnt32 >> (w + 1)) & 1) ? 5 : 6;
                }
            }
            for (u32 w = 0; w < a->n_words && ok; ++w)
            {
                ok = ok && word_op[w] != 0; // (a pass over a subset of the functions leaves words untouched: generic kernel)
                ops |= word_op[w] << (4 * w);
            }
            if (!ok)
                ops = 0;
        }
#define GB_AGG(KT_, OPS_)                                                                                                                              \
    do                                                                                                                                                \
    {                                                                                                                                                 \
        auto kern = k_agg_part_lds<KT_, 8, KT_, false, OPS_>;                                                                                          \
        rc = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ag) == hipSuccess ? CHGPU_OK : CHGPU_ERR_DEVICE; \
        if (rc == CHGPU_OK)                                                                                                                           \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds_ag, ctx->stream, a->t, d, (const KT_ *)pkeys, (const void *)pwords, (const void *)(pwords + wstride), \
                               (const u64 *)offsets, G, P, n, pending, S, K, cnt32, rows_per_chunk, (const u32 *)unit_start, unit_ctr, (const u8 *)nullptr); \
    } while (0)
#define GB_AGG_OPS(KT_)                                      \
    switch (ops)                                             \
    {                                                        \
        case 0x51: GB_AGG(KT_, 0x51); break; /* sum, count */     \
        case 0x15: GB_AGG(KT_, 0x15); break; /* count, sum */     \
        case 0x1: GB_AGG(KT_, 0x1); break;   /* sum */            \
        case 0x5: GB_AGG(KT_, 0x5); break;   /* count */          \
        case 0x53: GB_AGG(KT_, 0x53); break; /* sum(Float64), count = avg(Float64) */ \
        case 0x3: GB_AGG(KT_, 0x3); break;   /* sum(Float64) */   \
        case 0x21: GB_AGG(KT_, 0x21); break; /* sum, sum */       \
        case 0x521: GB_AGG(KT_, 0x521); break; /* sum, sum, count */ \
        case 0x97: GB_AGG(KT_, 0x97); break; /* sum(Float64) as a fixed-point pair */ \
        case 0x957: GB_AGG(KT_, 0x957); break; /* the same + count: avg(Float64) */ \
        default: GB_AGG(KT_, 0); break;                      \
    }
        if (key32)
        {
            GB_AGG_OPS(u32)
        }
        else
        {
            GB_AGG_OPS(u64)
        }
#undef GB_AGG_OPS
#undef GB_AGG
    }
    ctx->counters[6] += 3;
    ctx->counters[5] += n;
    if (rc == CHGPU_OK && hipGetLastError() != hipSuccess)
        rc = CHGPU_ERR_DEVICE;
    if (rc == CHGPU_OK)
        rc = agg_finish_rounds(a, d, pkeys, key32 ? CHGPU_U32 : CHGPU_U64, 0, n, pending);
    else
    {
        (void)hipGetLastError(); // do not leave a sticky launch error behind for the next call
        chgpu_set_error(rc, "partitioned aggregation launch failed");
    }
    return rc;
}

static int agg_add_block_impl(chgpu_agg * a, const chgpu_col * key_col, const chgpu_col * const * arg_cols, u64 row_begin, u64 row_end,
                              const chgpu_col * filter);

extern "C" int chgpu_agg_add_block(chgpu_agg * a, const chgpu_col * key_col, const chgpu_col * const * arg_cols,
                                   uint64_t row_begin, uint64_t row_end)
{
    ChgpuDeviceGuard _dev_guard(a ? a->ctx : nullptr);
    return agg_add_block_impl(a, key_col, arg_cols, row_begin, row_end, nullptr);
}

extern "C" int chgpu_agg_add_block_filtered(chgpu_agg * a, const chgpu_col * key_col, const chgpu_col * const * arg_cols,
                                            uint64_t row_begin, uint64_t row_end, const chgpu_col * filter_u8)
{
    ChgpuDeviceGuard _dev_guard(a ? a->ctx : nullptr);
    return agg_add_block_impl(a, key_col, arg_cols, row_begin, row_end, filter_u8);
}

// The strategies that have no fused form: FilterTransform's work is done first (every column of the block filtered by the
// mask, chgpu_filter_columns) and the filtered block aggregated.
static int agg_add_block_materialised(chgpu_agg * a, const chgpu_col * key_col, const chgpu_col * const * arg_cols, u64 row_begin, u64 row_end,
                                      const chgpu_col * filter)
{
    chgpu_ctx * ctx = a->ctx;
    const u64 n = row_end - row_begin;
    const chgpu_col * src[1 + AGG_MAX_AGGS];
    chgpu_col * views[2 + AGG_MAX_AGGS] = {};
    u32 m = 0;
    int rc = CHGPU_OK;
    auto view = [&](const chgpu_col * c) {
        if (rc == CHGPU_OK)
            rc = chgpu_col_slice(ctx, c, row_begin, n, &views[m]);
        if (rc == CHGPU_OK)
            ++m;
    };
    view(filter);
    view(key_col);
    u32 arg_slot[AGG_MAX_AGGS];
    for (u32 j = 0; j < a->n_aggs; ++j)
        if (a->kinds[j] != CHGPU_AGG_COUNT)
        {
            arg_slot[j] = m - 1; // index among the data columns (key = 0)
            view(arg_cols[j]);
        }
    chgpu_col * outs[1 + AGG_MAX_AGGS] = {};
    u64 kept = 0;
    const u32 n_data = m ? m - 1 : 0;
    if (rc == CHGPU_OK)
    {
        for (u32 k = 0; k < n_data; ++k)
            src[k] = views[1 + k];
        rc = chgpu_filter_columns(ctx, n_data, src, views[0], -1, outs, &kept);
    }
    if (rc == CHGPU_OK && kept)
    {
        const chgpu_col * fargs[AGG_MAX_AGGS] = {};
        for (u32 j = 0; j < a->n_aggs; ++j)
            if (a->kinds[j] != CHGPU_AGG_COUNT)
                fargs[j] = outs[arg_slot[j]];
        rc = agg_add_block_impl(a, outs[0], fargs, 0, kept, nullptr);
    }
    for (u32 k = 0; k < n_data; ++k)
        chgpu_col_free(outs[k]);
    for (u32 k = 0; k < m; ++k)
        chgpu_col_free(views[k]);
    return rc;
}

static int agg_add_block_impl(chgpu_agg * a, const chgpu_col * key_col, const chgpu_col * const * arg_cols, u64 row_begin, u64 row_end,
                              const chgpu_col * filter)
{
    CHGPU_REQUIRE(a, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(row_begin <= row_end, CHGPU_ERR_BAD_ARGUMENTS, "row_begin > row_end");
    chgpu_ctx * ctx = a->ctx;
    const u64 n = row_end - row_begin;
    if (filter)
    {
        CHGPU_REQUIRE(filter->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS, "filter must be a UInt8 column");
        CHGPU_REQUIRE(row_end <= filter->rows, CHGPU_ERR_SIZES_MISMATCH, "filter has %llu rows, block ends at %llu",
                      (unsigned long long)filter->rows, (unsigned long long)row_end);
    }
    for (u32 j = 0; j < a->n_aggs; ++j)
    {
        if (a->kinds[j] == CHGPU_AGG_COUNT)
            continue;
        CHGPU_REQUIRE(arg_cols && arg_cols[j], CHGPU_ERR_BAD_ARGUMENTS, "argument column %u is NULL", j);
        CHGPU_REQUIRE(arg_cols[j]->type == a->arg_types[j], CHGPU_ERR_BAD_ARGUMENTS, "argument column %u has type %d, expected %d", j, arg_cols[j]->type, a->arg_types[j]);
        CHGPU_REQUIRE(row_end <= arg_cols[j]->rows, CHGPU_ERR_SIZES_MISMATCH, "argument column %u has %llu rows, block ends at %llu", j,
                      (unsigned long long)arg_cols[j]->rows, (unsigned long long)row_end);
    }
    if (a->key_type < 0)
    {
        // executeWithoutKeyImpl (Aggregator.cpp:1276-1321): addBatchSinglePlace per function
        u64 kept = n;
        if (filter)
        {
            // addBatchSinglePlace under a condition (addManyConditional, AggregateFunctionSum.h:138-236); count = countBytesInFilter
            chgpu_col * fv = nullptr;
            CHGPU_TRY(chgpu_col_slice(ctx, filter, row_begin, n, &fv));
            const int rc = chgpu_count_bytes_in_filter(ctx, fv, &kept);
            chgpu_col_free(fv);
            CHGPU_TRY(rc);
        }
        for (u32 j = 0; j < a->n_aggs; ++j)
        {
            u64 * st = &a->host_words[a->word_off[j]];
            if (a->kinds[j] == CHGPU_AGG_COUNT)
                st[0] += kept;
            else if (a->kinds[j] == CHGPU_AGG_MIN || a->kinds[j] == CHGPU_AGG_MAX || a->kinds[j] == CHGPU_AGG_ANY)
            {
                if (kept == 0 || n == 0)
                    continue;
                void * scratch = nullptr;
                CHGPU_TRY(chgpu_scratch(ctx, 256, &scratch));
                unsigned long long * dev = (unsigned long long *)scratch;
                const u8 * cond = filter ? (const u8 *)filter->data : nullptr;
                u64 v = 0;
                if (a->kinds[j] != CHGPU_AGG_ANY)
                {
                    CHGPU_HIP(hipMemsetAsync(dev, 0, 8, ctx->stream));
                    hipLaunchKernelGGL(k_nokey_extremum, dim3(chgpu_grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, (const void *)arg_cols[j]->data, a->arg_types[j], row_begin, n, cond,
                                       a->kinds[j] == CHGPU_AGG_MIN ? 1 : 0, dev);
                    ctx->counters[6] += 1;
                    CHGPU_HIP(hipGetLastError());
                    CHGPU_TRY(chgpu_read_back(ctx, dev, &v, 8));
                    st[0] = v > st[0] ? v : st[0]; // order keys under an unsigned max (see agg_order_key); zero = no value yet
                }
                else if (st[0] == 0) // setIfFirst: only a state without a value takes one
                {
                    CHGPU_HIP(hipMemsetAsync(dev, 0xFF, 8, ctx->stream));
                    hipLaunchKernelGGL(k_nokey_first_row, dim3(chgpu_grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, row_begin, n, cond, dev);
                    CHGPU_HIP(hipGetLastError());
                    CHGPU_TRY(chgpu_read_back(ctx, dev, &v, 8));
                    if (v != ~0ull)
                    {
                        hipLaunchKernelGGL(k_nokey_load, dim3(1), dim3(64), 0, ctx->stream, (const void *)arg_cols[j]->data, a->arg_types[j], row_begin + v, (u64 *)dev);
                        CHGPU_HIP(hipGetLastError());
                        u64 bits = 0;
                        CHGPU_TRY(chgpu_read_back(ctx, dev, &bits, 8));
                        st[0] = ~(a->any_seq + v);
                        st[1] = bits;
                    }
                    ctx->counters[6] += 2;
                }
            }
            else
            {
                if (filter)
                    CHGPU_TRY(chgpu_sum_add_many_conditional(ctx, arg_cols[j], filter, row_begin, row_end, st));
                else
                    CHGPU_TRY(chgpu_sum_add_many(ctx, arg_cols[j], row_begin, row_end, st));
                if (a->kinds[j] == CHGPU_AGG_AVG)
                    st[1] += kept;
            }
        }
        a->any_seq += n;
        a->nokey_kept += kept;
        return CHGPU_OK;
    }
    CHGPU_REQUIRE(key_col, CHGPU_ERR_BAD_ARGUMENTS, "key column is NULL");
    CHGPU_REQUIRE(key_col->type == a->key_type, CHGPU_ERR_BAD_ARGUMENTS, "key column has type %d, expected %d", key_col->type, a->key_type);
    CHGPU_REQUIRE(row_end <= key_col->rows, CHGPU_ERR_SIZES_MISMATCH, "key column has %llu rows, block ends at %llu",
                  (unsigned long long)key_col->rows, (unsigned long long)row_end);
    if (n == 0)
        return CHGPU_OK;
    // Strategy by promised/observed cardinality:
    //   groups <= what one workgroup's LDS table holds (~70 % of its cells)   -> LDS-staged (RANGE mode / k_agg_rows_lds)
    //   more, with enough rows to amortise two extra passes                   -> PARTITIONED
    //   otherwise (or hopelessly many groups)                                 -> DIRECT
    // Callers that gave no size hint (the reference adapts too: consecutive-key cache hit rate, Aggregator.cpp:944-958;
    // two-level conversion, :83-89): the first 1 Mi rows go through the LDS-staged kernel and the number of groups they
    // produced is extrapolated to the whole input.
    const u32 lds_cells = agg_part_max_cells(a->ctx, agg_part_cell_bytes(a, n, nullptr));
    const u64 lds_groups = (u64)lds_cells * 7 / 10;
    if (a->size_hint <= lds_groups && a->n_groups > lds_groups)
        a->size_hint = a->n_groups * 2; // the table already outgrew the LDS strategy
    if (a->size_hint == 0 && !a->hint_probed && n >= (8ull << 20))
    {
        a->hint_probed = true;
        const u64 probe_rows = 1ull << 20;
        const u64 before = a->n_groups;
        CHGPU_TRY(agg_add_block_impl(a, key_col, arg_cols, row_begin, row_begin + probe_rows, filter));
        if (a->n_groups > lds_groups / 2)
        {
            const u64 est = agg_estimate_groups(a->n_groups - before, probe_rows);
            a->size_hint = before + est + est / 4;
        }
        return agg_add_block_impl(a, key_col, arg_cols, row_begin + probe_rows, row_end, filter);
    }
    // a WHERE mask is fused only into the RANGE-mode kernel; every other strategy gets the filtered block materialised first
    if (filter)
    {
        const bool partitioned = a->size_hint > lds_groups && n >= (4u << 20) && !chgpu_opt(ctx, "agg_no_partition", 0);
        const bool will_range = !partitioned && a->size_hint <= 65536 && n < (1ull << 32) && !chgpu_opt(ctx, "tune_agg_no_ranged", 0) && !a->has_extremum;
        if (!will_range)
            return agg_add_block_materialised(a, key_col, arg_cols, row_begin, row_end, filter);
        // The aggregation kernel is issue-bound: it spends nearly the same time on a masked-out row as on a kept one, while
        // chgpu_filter_columns runs at HBM speed.  Measured break-even at ~30 % of the rows kept (1e9 rows, 1000 groups,
        // 10 % kept: 5.7 ms fused vs 5.3 ms materialised), so big blocks count the mask first (0.35 ms per 1e9 rows).
        if (n >= (16u << 20))
        {
            chgpu_col * fv = nullptr;
            CHGPU_TRY(chgpu_col_slice(ctx, filter, row_begin, n, &fv));
            u64 kept = 0;
            const int rc = chgpu_count_bytes_in_filter(ctx, fv, &kept);
            chgpu_col_free(fv);
            CHGPU_TRY(rc);
            if (kept == 0)
                return CHGPU_OK;
            if (kept * 10 < n * 3)
                return agg_add_block_materialised(a, key_col, arg_cols, row_begin, row_end, filter);
        }
    }
    CHGPU_TRY(agg_fx_prepare_block(a, arg_cols, row_begin, n)); // (may widen the fixed-point window: before the descriptor is filled)
    AggDesc d;
    agg_fill_desc(a, arg_cols, &d);

    const u64 n_words64 = (n + 63) / 64;
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, n_words64 * sizeof(u64) + 256, &scratch));
    u64 * pending = (u64 *)scratch;

    if (a->has_extremum)
    {
        // min / max / any states: one emplace + one atomic per state word and row (the LDS-staged and partitioned plans carry additive words only)
        CHGPU_TRY(agg_ensure_table(a));
        d.row_seq = a->any_seq - row_begin; // row i of the columns is the (any_seq + i - row_begin)-th row of the aggregation
        hipLaunchKernelGGL(k_agg_rows_direct<AGG_MODE_ALL>, dim3(chgpu_grid_for(ctx, n, AGG_THREADS, 8)), dim3(AGG_THREADS), 0, ctx->stream, a->t, d, key_col->data, a->key_type,
                           row_begin, n, pending);
        ctx->counters[6] += 1;
        ctx->counters[5] += n;
        CHGPU_HIP(hipGetLastError());
        CHGPU_TRY(agg_finish_rounds(a, d, key_col->data, a->key_type, row_begin, n, pending));
        if (a->word_any)
        {
            hipLaunchKernelGGL(k_agg_any_resolve, dim3(chgpu_grid_for(ctx, n, AGG_THREADS, 8)), dim3(AGG_THREADS), 0, ctx->stream, a->t, d, key_col->data, a->key_type, row_begin, n);
            ctx->counters[6] += 1;
            CHGPU_HIP(hipGetLastError());
            a->any_seq += n;
        }
        return CHGPU_OK;
    }

    // PARTITIONED strategy: large promised cardinality and enough rows to amortise two extra passes
    {
        u32 n_argwords = 0;
        for (u32 j = 0; j < a->n_aggs; ++j)
            if (a->kinds[j] != CHGPU_AGG_COUNT)
                ++n_argwords;
        // Two or more argument words: ONE PASS PER WORD through the tile-sorted plan (each pass sorts {key, its word} and aggregates into
        // cells that hold only its own state words; every count() rides in the first) -- 7 ms per word and 1e9 rows, against 22 ms for
        // the two-word scatter plan, whose 4096-row tiles leave 8-row runs (tools/bench_two_words.py).  A shape the plan does not take
        // answers NOT_IMPLEMENTED on its first pass, before anything was added: the older routes below take over.
        if (n_argwords >= 2 && a->size_hint > lds_groups && n >= (4u << 20) && !chgpu_opt(ctx, "agg_no_partition", 0) && !chgpu_opt(ctx, "tune_gb_no_tiled", 0)
            && !chgpu_opt(ctx, "tune_gb_no_word_passes", 0))
        {
            // every pass is asked first whether the plan takes it (nothing may be added before all of them are known to run)
            int rc = CHGPU_OK;
            for (int run = 0; run < 2 && rc == CHGPU_OK; ++run)
            {
                bool first = true;
                for (u32 j = 0; j < a->n_aggs && rc == CHGPU_OK; ++j)
                {
                    if (a->kinds[j] == CHGPU_AGG_COUNT)
                        continue;
                    u32 mask = 1u << j;
                    if (first)
                        for (u32 c = 0; c < a->n_aggs; ++c)
                            if (a->kinds[c] == CHGPU_AGG_COUNT)
                                mask |= 1u << c;
                    rc = agg_add_block_partitioned(a, key_col, arg_cols, row_begin, n, 1, 0, 0, mask, /*word_pass*/ run == 0 ? 2 : 1);
                    if (run == 1 && rc == CHGPU_ERR_NOT_IMPLEMENTED) // (the probe said yes: not reached)
                        rc = chgpu_set_error(CHGPU_ERR_LOGICAL, "a pass of a per-word GROUP BY was refused after its probe");
                    if (run == 1 && !first && rc == CHGPU_OK)
                        ctx->counters[5] -= n; // rows were counted once per pass
                    first = false;
                }
            }
            if (rc != CHGPU_ERR_NOT_IMPLEMENTED)
                return rc;
        }
        if (a->size_hint > lds_groups && n >= (4u << 20) && n_argwords <= GBP_MAX_K && !chgpu_opt(ctx, "agg_no_partition", 0))
        {
            int rc = agg_add_block_partitioned(a, key_col, arg_cols, row_begin, n, n_argwords);
            if (rc != CHGPU_ERR_NOT_IMPLEMENTED)
                return rc;
        }
        else if (a->size_hint > lds_groups && n >= (4u << 20) && !chgpu_opt(ctx, "agg_no_partition", 0))
        {
            // more argument columns than a partition buffer row carries: several partitioned calls over the same rows, each with
            // two of them (plus every count() in the first) -- ~12 ms per 1e9 rows and call, against one HBM atomic per row
            // and state word on the DIRECT path
            u32 masks[AGG_MAX_AGGS], ks[AGG_MAX_AGGS], n_calls = 0;
            u32 cur = 0, cur_k = 0;
            for (u32 j = 0; j < a->n_aggs; ++j)
            {
                if (a->kinds[j] == CHGPU_AGG_COUNT)
                    continue;
                cur |= 1u << j;
                if (++cur_k == GBP_MAX_K)
                {
                    masks[n_calls] = cur, ks[n_calls] = cur_k, ++n_calls;
                    cur = 0, cur_k = 0;
                }
            }
            if (cur_k)
                masks[n_calls] = cur, ks[n_calls] = cur_k, ++n_calls;
            for (u32 j = 0; j < a->n_aggs; ++j)
                if (a->kinds[j] == CHGPU_AGG_COUNT)
                    masks[0] |= 1u << j;
            int rc = CHGPU_OK;
            for (u32 c = 0; c < n_calls && rc == CHGPU_OK; ++c)
            {
                rc = agg_add_block_partitioned(a, key_col, arg_cols, row_begin, n, ks[c], 0, 0, masks[c]);
                if (rc == CHGPU_ERR_NOT_IMPLEMENTED && c == 0)
                    break; // nothing applied yet: fall through to the other strategies
            }
            if (rc != CHGPU_ERR_NOT_IMPLEMENTED)
            {
                ctx->counters[5] -= (u64)(n_calls - 1) * n; // rows were counted once per call
                return rc;
            }
        }
    }
    CHGPU_TRY(agg_ensure_table(a));
    // strategy: LDS-staged unless the caller promised a large cardinality (where nearly every key misses the LDS table)
    const bool use_lds = a->size_hint <= 65536; // beyond that nearly every key misses a workgroup's LDS table
    // RANGE mode of the partition-aggregate kernel: 4/8-byte keys; a launch takes at most GBP_MAX_K argument columns of one
    // width (8, 4 or 1 B), so the aggregate functions are split into PASSES over the same rows -- each pass re-reads the key
    // column and updates its own state words of the same groups (TPC-H Q1's seven sums and averages: 4 passes x ~20 B/row
    // instead of one trip through the generic kernel, which is 6x slower per row).
    bool ranged = use_lds && n < (1ull << 32) && !chgpu_opt(ctx, "tune_agg_no_ranged", 0); // keys of 1, 2, 4 or 8 bytes: every key type

    if (ranged)
    {
        struct Pass
        {
            u32 n = 0;           // argument columns in this pass
            u32 agg[GBP_MAX_K];  // their aggregate indices
            size_t aw = 8;
        };
        Pass passes[AGG_MAX_AGGS];
        u32 n_passes = 0;
        for (u32 j = 0; j < a->n_aggs; ++j)
        {
            if (a->kinds[j] == CHGPU_AGG_COUNT)
                continue;
            const size_t w = chgpu_type_size(a->arg_types[j]);
            u32 p = 0;
            for (; p < n_passes; ++p) // first pass of this width with a free slot
                if (passes[p].aw == w && passes[p].n < GBP_MAX_K)
                    break;
            if (p == n_passes)
            {
                passes[n_passes].aw = w;
                ++n_passes;
            }
            passes[p].agg[passes[p].n++] = j;
        }
        if (n_passes == 0)
            n_passes = 1; // only count(): one pass without argument columns

        const bool key32 = chgpu_type_size(a->key_type) <= 4, key8 = chgpu_type_size(a->key_type) == 1, key16 = chgpu_type_size(a->key_type) == 2;
        u32 cnt32 = 0;
        (void)agg_part_cell_bytes(a, n, &cnt32);
        const u32 n4 = (u32)__builtin_popcount(cnt32), n8 = a->n_words - n4;
        // cells: four times the promised groups (4096 when nothing was promised), bounded by ~150 KiB of LDS; tables of up
        // to ~76 KiB let two 1024-thread workgroups share a CU
        const u32 s_dflt = (u32)chgpu_opt(ctx, "tune_agg_ranged_s", 4096);
        u32 S = s_dflt;
        if (a->size_hint)
            for (S = 1024; S < 4 * a->size_hint && S < lds_cells; S <<= 1)
                ;
        if (S > lds_cells)
            S = lds_cells;
        const size_t keys_lds = ((size_t)(key32 ? 4 : 8) * (S + 1) + 7) & ~(size_t)7;
        const size_t lds_ag = keys_lds + (size_t)(S + 1) * (8 * n8 + 4 * n4) + 16;
        const u32 wg_per_cu = lds_ag <= 76 * 1024 ? 2 : 1;
        // flushes may claim up to grid * (S+1) cells above max fill: keep that inside the slack (capacity/2)
        const u64 max_grid = (a->t.capacity / 2) / (S + 1);
        u64 chunks = (u64)ctx->num_cus * wg_per_cu;
        if (chunks > max_grid)
            chunks = max_grid ? max_grid : 1;
        if (chunks > (n + 4095) / 4096)
            chunks = (n + 4095) / 4096;
        const u64 rows_per_chunk = ((n + chunks - 1) / chunks + 63) / 64 * 64;
        chunks = (n + rows_per_chunk - 1) / rows_per_chunk;
        const u8 * cond_ptr = filter ? (const u8 *)filter->data + row_begin : nullptr;
        for (u32 p = 0; p < n_passes; ++p)
        {
            // this pass's descriptor: its argument functions, plus every count() in the first pass; state word indices are
            // the aggregator's own, so all passes meet in the same cells
            AggDesc dp = d;
            dp.n_aggs = 0;
            const void * rwords[GBP_MAX_K] = {nullptr, nullptr};
            for (u32 c = 0; c < passes[p].n; ++c)
            {
                const u32 j = passes[p].agg[c];
                dp.a[dp.n_aggs] = d.a[j];
                dp.a[dp.n_aggs].pre = c;
                ++dp.n_aggs;
                rwords[c] = (const char *)arg_cols[j]->data + row_begin * passes[p].aw;
            }
            if (p == 0)
                for (u32 j = 0; j < a->n_aggs; ++j)
                    if (a->kinds[j] == CHGPU_AGG_COUNT)
                        dp.a[dp.n_aggs++] = d.a[j];
            const u32 rk = passes[p].n;
            const size_t aw = passes[p].aw;
            bool need_ext = false; // a signed narrow integer or Float32 argument in this pass
            for (u32 c = 0; c < passes[p].n; ++c)
            {
                const int at = a->arg_types[passes[p].agg[c]];
                need_ext = need_ext || at == CHGPU_I8 || at == CHGPU_I16 || at == CHGPU_I32 || at == CHGPU_F32;
            }
            CHGPU_HIP(hipMemsetAsync(pending, 0, n_words64 * sizeof(u64), ctx->stream));
#define RANGE_LAUNCH_X(KT_, AW_, KS_, X_)                                                                                                              \
    do                                                                                                                                                \
    {                                                                                                                                                 \
        CHGPU_HIP(hipFuncSetAttribute((const void *)k_agg_part_lds<KT_, AW_, KS_, X_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ag));     \
        hipLaunchKernelGGL((k_agg_part_lds<KT_, AW_, KS_, X_>), dim3((u32)chunks), dim3(1024), lds_ag, ctx->stream, a->t, dp, (const KS_ *)key_col->data + row_begin, \
                           rwords[0], rwords[1], (const u64 *)nullptr, 1u, (u32)chunks, n, pending, S, rk, cnt32, rows_per_chunk, (const u32 *)nullptr,  \
                           (u32 *)nullptr, cond_ptr);                                                                                                 \
    } while (0)
#define RANGE_LAUNCH_KS(KT_, AW_, KS_) do { if ((AW_) < 8 && need_ext) RANGE_LAUNCH_X(KT_, AW_, KS_, true); else RANGE_LAUNCH_X(KT_, AW_, KS_, false); } while (0)
#define RANGE_LAUNCH(KT_, AW_) RANGE_LAUNCH_KS(KT_, AW_, KT_)
            if (key8)
            {
                if (aw == 8) RANGE_LAUNCH_KS(u32, 8, u8); else if (aw == 4) RANGE_LAUNCH_KS(u32, 4, u8); else if (aw == 2) RANGE_LAUNCH_KS(u32, 2, u8); else RANGE_LAUNCH_KS(u32, 1, u8);
            }
            else if (key16) // UInt16 (Date) / Int16 keys
            {
                if (aw == 8) RANGE_LAUNCH_KS(u32, 8, u16); else if (aw == 4) RANGE_LAUNCH_KS(u32, 4, u16); else if (aw == 2) RANGE_LAUNCH_KS(u32, 2, u16); else RANGE_LAUNCH_KS(u32, 1, u16);
            }
            else if (key32)
            {
                if (aw == 8) RANGE_LAUNCH(u32, 8); else if (aw == 4) RANGE_LAUNCH(u32, 4); else if (aw == 2) RANGE_LAUNCH(u32, 2); else RANGE_LAUNCH(u32, 1);
            }
            else
            {
                if (aw == 8) RANGE_LAUNCH(u64, 8); else if (aw == 4) RANGE_LAUNCH(u64, 4); else if (aw == 2) RANGE_LAUNCH(u64, 2); else RANGE_LAUNCH(u64, 1);
            }
#undef RANGE_LAUNCH
#undef RANGE_LAUNCH_KS
#undef RANGE_LAUNCH_X
            ctx->counters[6] += 1;
            CHGPU_HIP(hipGetLastError());
            // rows this pass could not place (table at max fill) are retried with THIS pass's functions only
            CHGPU_TRY(agg_finish_rounds(a, dp, key_col->data, a->key_type, row_begin, n, pending));
        }
        ctx->counters[5] += n;
        return CHGPU_OK;
    }
    if (use_lds)
    {
        // LDS cells per workgroup: the largest power of two with (1 + n_words) * 8 * (S+1) <= AGG_LDS_BYTES
        u32 S = 4096;
        while ((size_t)(S + 1) * 8 * (1 + a->n_words) > AGG_LDS_BYTES && S > 64)
            S >>= 1;
        // flushes may claim up to grid * (S+1) cells above max fill: keep that inside the slack (capacity/2)
        u64 max_grid = (a->t.capacity / 2) / (S + 1);
        const u32 lds_threads = (u32)chgpu_opt(ctx, "tune_agg_lds_threads", 512);
        u32 grid = chgpu_grid_for(ctx, n, lds_threads, lds_threads >= 1024 ? 2 : 4);
        if (grid > max_grid)
            grid = (u32)(max_grid ? max_grid : 1);
        const size_t lds = (size_t)(S + 1) * 8 * (1 + a->n_words);
        hipLaunchKernelGGL(k_agg_rows_lds, dim3(grid), dim3(lds_threads), lds, ctx->stream, a->t, d, key_col->data, a->key_type, row_begin, n, pending, S);
    }
    else
    {
        const u32 grid = chgpu_grid_for(ctx, n, AGG_THREADS, 8);
        hipLaunchKernelGGL(k_agg_rows_direct<AGG_MODE_ALL>, dim3(grid), dim3(AGG_THREADS), 0, ctx->stream, a->t, d, key_col->data, a->key_type, row_begin, n, pending);
    }
    ctx->counters[6] += 1;
    ctx->counters[5] += n;
    CHGPU_HIP(hipGetLastError());

    return agg_finish_rounds(a, d, key_col->data, a->key_type, row_begin, n, pending);
}

// resize on overflow (HashTable.h:921-944): grow + rehash, then re-run only the rows left pending, until none is
static int agg_finish_rounds(chgpu_agg * a, const AggDesc & d, const void * keys, int key_type, u64 row_begin, u64 n, u64 * pending)
{
    chgpu_ctx * ctx = a->ctx;
    for (int round = 0; round < 64; ++round)
    {
        AggCtrl c;
        CHGPU_TRY(agg_read_ctrl(a, &c));
        if (!c.overflow && c.n_groups <= a->t.max_fill)
            return CHGPU_OK;
        CHGPU_TRY(agg_grow(a, c.n_groups, c.has_zero != 0));
        if (!c.overflow)
            return CHGPU_OK;
        const u32 grid = chgpu_grid_for(ctx, n, AGG_THREADS, 8);
        hipLaunchKernelGGL(k_agg_rows_direct<AGG_MODE_PENDING>, dim3(grid), dim3(AGG_THREADS), 0, ctx->stream, a->t, d, keys, key_type, row_begin, n, pending);
        ctx->counters[6] += 1;
        CHGPU_HIP(hipGetLastError());
    }
    return chgpu_set_error(CHGPU_ERR_LOGICAL, "aggregation table did not converge after 64 growth rounds");
}


// merge tuples (keys + state word columns) with overflow handling
static int agg_merge_tuples(chgpu_agg * a, const u64 * src_keys, const u64 * src_words, u64 src_stride, u64 n, int skip_zero_keys, u64 zero_slot_index)
{
    chgpu_ctx * ctx = a->ctx;
    if (n == 0)
        return CHGPU_OK;
    CHGPU_TRY(agg_ensure_table(a));
    const u64 n_words64 = (n + 63) / 64;
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, n_words64 * sizeof(u64) + 256, &scratch));
    u64 * pending = (u64 *)scratch;
    const u32 grid = chgpu_grid_for(ctx, n, AGG_THREADS, 8);
    hipLaunchKernelGGL(k_agg_tuples<AGG_MODE_ALL>, dim3(grid), dim3(AGG_THREADS), 0, ctx->stream, a->t, a->n_words, a->word_is_f64, agg_fx_words(a),
                       src_keys, src_words, src_stride, n, skip_zero_keys, zero_slot_index, 1, pending);
    ctx->counters[6] += 1;
    CHGPU_HIP(hipGetLastError());
    for (int round = 0; round < 64; ++round)
    {
        AggCtrl c;
        CHGPU_TRY(agg_read_ctrl(a, &c));
        if (!c.overflow)
            return CHGPU_OK;
        CHGPU_TRY(agg_grow(a, c.n_groups, c.has_zero != 0));
        hipLaunchKernelGGL(k_agg_tuples<AGG_MODE_PENDING>, dim3(grid), dim3(AGG_THREADS), 0, ctx->stream, a->t, a->n_words, a->word_is_f64, agg_fx_words(a),
                           src_keys, src_words, src_stride, n, skip_zero_keys, zero_slot_index, 1, pending);
        ctx->counters[6] += 1;
        CHGPU_HIP(hipGetLastError());
    }
    return chgpu_set_error(CHGPU_ERR_LOGICAL, "aggregation merge did not converge after 64 growth rounds");
}

// one without-key state word of `src_words` folded into dst (mergeWithoutKeyDataImpl, Aggregator.cpp:2584-2628); returns the words consumed
static u32 agg_merge_host_word(chgpu_agg * dst, u32 w, const u64 * src_words)
{
    if ((dst->word_any >> w) & 1)
    {
        if (dst->host_words[w] == 0 && src_words[w] != 0) // changeFirstTime: a state that has a value keeps it
        {
            dst->host_words[w] = src_words[w];
            dst->host_words[w + 1] = src_words[w + 1];
        }
        return 2;
    }
    if ((dst->word_is_f64 >> (16 + w)) & 1)
        dst->host_words[w] = src_words[w] > dst->host_words[w] ? src_words[w] : dst->host_words[w]; // min / max order keys
    else if ((dst->word_is_f64 >> w) & 1)
    {
        double x, y;
        memcpy(&x, &dst->host_words[w], 8);
        memcpy(&y, &src_words[w], 8);
        x += y;
        memcpy(&dst->host_words[w], &x, 8);
    }
    else
        dst->host_words[w] += src_words[w];
    return 1;
}

static bool agg_same_shape(const chgpu_agg * x, const chgpu_agg * y)
{
    if (x->key_type != y->key_type || x->n_aggs != y->n_aggs)
        return false;
    for (u32 j = 0; j < x->n_aggs; ++j)
        if (x->kinds[j] != y->kinds[j] || x->arg_types[j] != y->arg_types[j])
            return false;
    return true;
}

extern "C" int chgpu_agg_merge(chgpu_agg * dst, const chgpu_agg * src)
{
    ChgpuDeviceGuard _dev_guard(dst ? dst->ctx : nullptr);
    CHGPU_REQUIRE(dst && src, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(agg_same_shape(dst, src), CHGPU_ERR_BAD_ARGUMENTS, "cannot merge aggregation states of different shape");
    // variants of different pipeline streams live on different contexts: the source's kernels run on ITS stream and must have finished
    // before this context's stream reads its table (the reference merges after every stream has finished consuming)
    if (src->ctx != dst->ctx)
        CHGPU_HIP(hipStreamSynchronize(src->ctx->stream));
    if (dst->key_type < 0)
    {
        // mergeWithoutKeyDataImpl (Aggregator.cpp:2584-2628)
        for (u32 w = 0; w < dst->n_words;)
            w += agg_merge_host_word(dst, w, src->host_words);
        dst->nokey_kept += src->nokey_kept;
        return CHGPU_OK;
    }
    if (!src->table_mem)
        return CHGPU_OK;
    CHGPU_REQUIRE(dst->n_words == src->n_words, CHGPU_ERR_BAD_ARGUMENTS, "cannot merge aggregations created under different deterministic_float_sums settings");
    if (dst->word_fx || src->word_fx)
    {
        // fixed-point sums: both sides to ONE window first (the source's states are re-expressed in place: same values, possibly a coarser
        // unit -- the reference's merge consumes its source too), or both back to doubles when one of them met a NaN / infinity
        chgpu_agg * s = const_cast<chgpu_agg *>(src);
        if (!dst->word_fx || !s->word_fx)
        {
            CHGPU_TRY(agg_fx_to_plain(dst));
            CHGPU_TRY(agg_fx_to_plain(s));
        }
        else if (s->fx_base_set)
        {
            if (!dst->fx_base_set)
            {
                CHGPU_TRY(agg_fx_set_window(dst, s->fx_base, s->fx_log_cap));
                dst->fx_rows += s->fx_rows;
                dst->fx_emin = s->fx_emin;
            }
            else
            {
                int log_cap = dst->fx_log_cap > s->fx_log_cap ? dst->fx_log_cap : s->fx_log_cap;
                const u64 rows = dst->fx_rows + s->fx_rows;
                while (log_cap < 62 && rows > (1ull << log_cap))
                    log_cap += 8;
                const int bd = dst->fx_base + (log_cap - dst->fx_log_cap), bs = s->fx_base + (log_cap - s->fx_log_cap);
                const int base = bd > bs ? bd : bs;
                const int emin = dst->fx_emin < s->fx_emin ? dst->fx_emin : s->fx_emin;
                if (emin + 1 - FX_MIN_BITS < base)
                {
                    CHGPU_TRY(agg_fx_to_plain(dst));
                    CHGPU_TRY(agg_fx_to_plain(s));
                }
                else
                {
                    CHGPU_TRY(agg_fx_set_window(dst, base, log_cap));
                    CHGPU_TRY(agg_fx_set_window(s, base, log_cap));
                    dst->fx_emin = emin;
                    dst->fx_rows = rows;
                }
            }
        }
        if (src->ctx != dst->ctx)
            CHGPU_HIP(hipStreamSynchronize(src->ctx->stream)); // (the re-expression ran on the source's stream)
    }
    // the source's zero cell participates only when it is set
    AggCtrl sc;
    CHGPU_TRY(chgpu_read_back(dst->ctx, src->t.ctrl, &sc, sizeof(sc)));
    const u64 n = src->t.capacity + (sc.has_zero ? 1 : 0);
    return agg_merge_tuples(dst, src->t.keys, src->t.words, src->t.capacity + 1, n, 1, sc.has_zero ? src->t.capacity : ~0ull);
}

extern "C" int chgpu_agg_merge_states(chgpu_agg * dst, const chgpu_col * key_col, const chgpu_col * const * state_cols, uint64_t rows)
{
    ChgpuDeviceGuard _dev_guard(dst ? dst->ctx : nullptr);
    CHGPU_REQUIRE(dst && state_cols, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    chgpu_ctx * ctx = dst->ctx;
    for (u32 w = 0; w < dst->n_pub_words; ++w)
    {
        CHGPU_REQUIRE(state_cols[w], CHGPU_ERR_BAD_ARGUMENTS, "state column %u is NULL", w);
        CHGPU_REQUIRE(chgpu_type_size(state_cols[w]->type) == 8, CHGPU_ERR_BAD_ARGUMENTS, "state column %u must be 8 bytes wide", w);
        CHGPU_REQUIRE(state_cols[w]->rows >= rows, CHGPU_ERR_SIZES_MISMATCH, "state column %u shorter than %llu rows", w, (unsigned long long)rows);
    }
    if (dst->key_type < 0)
    {
        CHGPU_REQUIRE(rows <= 1, CHGPU_ERR_BAD_ARGUMENTS, "without_key states merge one row at a time");
        if (rows == 0)
            return CHGPU_OK;
        u64 in[AGG_MAX_WORDS] = {0};
        for (u32 w = 0; w < dst->n_words; ++w)
            CHGPU_TRY(chgpu_read_back(ctx, state_cols[w]->data, &in[w], 8));
        bool any_set = false;
        for (u32 w = 0; w < dst->n_words; ++w)
            any_set = any_set || in[w] != 0;
        for (u32 w = 0; w < dst->n_words;)
            w += agg_merge_host_word(dst, w, in);
        dst->nokey_kept += any_set ? 1 : 0; // (a partial state of an empty input is all zeros)
        return CHGPU_OK;
    }
    CHGPU_REQUIRE(key_col && key_col->rows >= rows, CHGPU_ERR_SIZES_MISMATCH, "key column shorter than %llu rows", (unsigned long long)rows);
    CHGPU_REQUIRE(key_col->type == dst->key_type, CHGPU_ERR_BAD_ARGUMENTS, "key column type mismatch");
    if (rows == 0)
        return CHGPU_OK;
    // Float64 sum states arriving for fixed-point sums: each is one value for the window (or the end of the fixed-point mode)
    if (dst->word_fx)
    {
        u32 emax = 0, emin = 2047;
        bool bad = false;
        for (u32 w = 0; w < dst->n_pub_words && !bad; ++w)
            if ((dst->word_fx >> w) & 1)
            {
                u32 e = 0, em = 2047;
                CHGPU_TRY(agg_fx_stats(ctx, state_cols[w]->data, CHGPU_F64, 0, rows, &e, &em, &bad));
                emax = e > emax ? e : emax;
                emin = em < emin ? em : emin;
            }
        if (bad)
            CHGPU_TRY(agg_fx_to_plain(dst));
        else
            CHGPU_TRY(agg_fx_admit(dst, emax, emin, rows));
    }
    // stage into one SoA buffer [keys u64][words...] so the tuple kernel sees a single stride
    chgpu_col * stage = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, rows * (1 + dst->n_words), &stage));
    u64 * sk = (u64 *)stage->data;
    int rc = CHGPU_OK;
    {
        const u32 grid = chgpu_grid_for(ctx, rows, 256, 8);
        // widen keys with the same zero-extension as the row path (reuse k_narrow in reverse via a tiny lambda kernel)
        switch (chgpu_type_size(key_col->type))
        {
            case 8: rc = hipMemcpyAsync(sk, key_col->data, rows * 8, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess ? CHGPU_OK : CHGPU_ERR_DEVICE; break;
            default:
            {
                extern __global__ void k_widen_keys(const void *, int, u64, u64 *);
                hipLaunchKernelGGL(k_widen_keys, dim3(grid), dim3(256), 0, ctx->stream, (const void *)key_col->data, key_col->type, (u64)rows, sk);
                break;
            }
        }
        for (u32 w = 0; w < dst->n_pub_words && rc == CHGPU_OK; ++w)
        {
            if ((dst->word_fx >> w) & 1)
            {
                hipLaunchKernelGGL(k_fx_from_double, dim3(grid), dim3(256), 0, ctx->stream, (const u64 *)state_cols[w]->data, (u64)rows, dst->fx_base,
                                   sk + (u64)(w + 1) * rows, sk + (u64)(dst->fx_hi[w] + 1) * rows);
                ctx->counters[6] += 1;
                continue;
            }
            rc = hipMemcpyAsync(sk + (u64)(w + 1) * rows, state_cols[w]->data, rows * 8, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess ? CHGPU_OK : CHGPU_ERR_DEVICE;
        }
    }
    if (rc == CHGPU_OK)
        rc = agg_merge_tuples(dst, sk, sk + rows, rows, rows, 0, ~0ull);
    else
        chgpu_set_error(CHGPU_ERR_DEVICE, "staging copy failed");
    chgpu_col_free(stage);
    return rc;
}

__global__ __launch_bounds__(256) void k_widen_keys(const void * keys, int type, u64 n, u64 * out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
        out[i] = load_key_zext(keys, type, i);
}

extern "C" int chgpu_agg_size(chgpu_agg * a, uint64_t * groups)
{
    ChgpuDeviceGuard _dev_guard(a ? a->ctx : nullptr);
    CHGPU_REQUIRE(a && groups, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    if (a->key_type < 0)
    {
        *groups = 1; // no-key aggregation always yields one row (AggregatingTransform.cpp:700-708)
        return CHGPU_OK;
    }
    if (!a->table_mem)
    {
        *groups = 0;
        return CHGPU_OK;
    }
    AggCtrl c;
    CHGPU_TRY(agg_read_ctrl(a, &c));
    *groups = c.n_groups;
    return CHGPU_OK;
}

// keys + raw state words compacted out of the table (table order; the reference's order is unspecified too)
static int agg_export(chgpu_agg * a, chgpu_col ** keys_out, chgpu_col ** word_cols /* [n_words] */, u64 * groups)
{
    chgpu_ctx * ctx = a->ctx;
    for (u32 w = 0; w < a->n_words; ++w)
        word_cols[w] = nullptr;
    if (a->key_type < 0)
    {
        for (u32 w = 0; w < a->n_words; ++w)
        {
            CHGPU_TRY(chgpu_col_upload(ctx, ((a->word_is_f64 >> w) & 1) ? CHGPU_F64 : CHGPU_U64, &a->host_words[w], 1, &word_cols[w]));
        }
        if (keys_out)
            *keys_out = nullptr;
        *groups = 1;
        return CHGPU_OK;
    }
    if (!a->table_mem)
    {
        if (keys_out)
            CHGPU_TRY(chgpu_col_new(ctx, a->key_type, 0, keys_out));
        for (u32 w = 0; w < a->n_pub_words; ++w)
            CHGPU_TRY(chgpu_col_new(ctx, (((a->word_is_f64 | a->word_fx) >> w) & 1) ? CHGPU_F64 : CHGPU_U64, 0, &word_cols[w]));
        *groups = 0;
        return CHGPU_OK;
    }
    const u64 cells = a->t.capacity + 1;
    chgpu_col * mask = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U8, cells, &mask));
    const u32 grid = chgpu_grid_for(ctx, cells, 256, 8);
    hipLaunchKernelGGL(k_fix_zero_key, dim3(1), dim3(64), 0, ctx->stream, a->t.keys, a->t.capacity);
    hipLaunchKernelGGL(k_occupied_mask, dim3(grid), dim3(256), 0, ctx->stream, a->t.keys, a->t.capacity, a->t.ctrl, (u8 *)mask->data);
    ctx->counters[6] += 2;
    int rc = CHGPU_OK;
    u64 n_out = 0;
    chgpu_col view;
    view.ctx = ctx;
    view.rows = cells;
    view.owns = false;
    // keys and every state word in ONE filter call: one count + scan + read-back instead of one per column (all 8-byte columns: they share
    // a compaction kernel)
    chgpu_col * k64 = nullptr;
    {
        chgpu_col views[1 + AGG_MAX_WORDS];
        const chgpu_col * vin[1 + AGG_MAX_WORDS];
        chgpu_col * vout[1 + AGG_MAX_WORDS] = {nullptr};
        for (u32 c = 0; c <= a->n_words; ++c)
        {
            views[c] = view;
            views[c].type = c == 0 ? CHGPU_U64 : (((a->word_is_f64 >> (c - 1)) & 1) ? CHGPU_F64 : CHGPU_U64);
            views[c].data = c == 0 ? (void *)a->t.keys : (void *)(a->t.words + (u64)(c - 1) * cells);
            vin[c] = &views[c];
        }
        rc = chgpu_filter_columns(ctx, 1 + a->n_words, vin, mask, 0, vout, &n_out);
        if (rc == CHGPU_OK)
        {
            k64 = vout[0];
            for (u32 w = 0; w < a->n_words; ++w)
                word_cols[w] = vout[1 + w];
        }
    }
    chgpu_col_free(mask);
    auto drop_words = [&]() {
        for (u32 w = 0; w < a->n_words; ++w)
        {
            chgpu_col_free(word_cols[w]); // the word columns already filtered when a later step failed
            word_cols[w] = nullptr;
        }
    };
    if (rc == CHGPU_OK)
    {
        // the fixed-point sums leave as the doubles they stand for (one rounding per group); the spare high words stay inside
        for (u32 w = 0; w < a->n_pub_words; ++w)
            if ((a->word_fx >> w) & 1)
            {
                if (n_out)
                {
                    hipLaunchKernelGGL(k_fx_to_double, dim3(chgpu_grid_for(ctx, n_out, 256, 8)), dim3(256), 0, ctx->stream, (u64 *)word_cols[w]->data,
                                       (u64 *)word_cols[a->fx_hi[w]]->data, n_out, a->fx_base, 0);
                    ctx->counters[6] += 1;
                }
                word_cols[w]->type = CHGPU_F64;
            }
        for (u32 w = a->n_pub_words; w < a->n_words; ++w)
        {
            chgpu_col_free(word_cols[w]); // pooled: reuse is stream-ordered behind the conversion
            word_cols[w] = nullptr;
        }
    }
    if (rc != CHGPU_OK)
    {
        chgpu_col_free(k64);
        drop_words();
        return rc;
    }
    if (keys_out)
    {
        if (chgpu_type_size(a->key_type) == 8)
        {
            k64->type = a->key_type;
            *keys_out = k64;
        }
        else
        {
            // insertKeyIntoColumns casts the UInt64 table key back to the column type
            chgpu_col * kn = nullptr;
            rc = chgpu_col_new(ctx, a->key_type, n_out, &kn);
            if (rc == CHGPU_OK && n_out)
            {
                const u32 g2 = chgpu_grid_for(ctx, n_out, 256, 8);
                if (chgpu_type_size(a->key_type) == 4)
                    hipLaunchKernelGGL(k_narrow_keys<u32>, dim3(g2), dim3(256), 0, ctx->stream, (const u64 *)k64->data, n_out, (u32 *)kn->data);
                else if (chgpu_type_size(a->key_type) == 2)
                    hipLaunchKernelGGL(k_narrow_keys<u16>, dim3(g2), dim3(256), 0, ctx->stream, (const u64 *)k64->data, n_out, (u16 *)kn->data);
                else
                    hipLaunchKernelGGL(k_narrow_keys<u8>, dim3(g2), dim3(256), 0, ctx->stream, (const u64 *)k64->data, n_out, (u8 *)kn->data);
                ctx->counters[6] += 1;
            }
            chgpu_col_free(k64); // pooled: any reuse is stream-ordered behind the narrow kernel
            if (rc != CHGPU_OK)
            {
                drop_words();
                return rc;
            }
            *keys_out = kn;
        }
    }
    else
        chgpu_col_free(k64);
    *groups = n_out;
    return CHGPU_OK;
}

extern "C" int chgpu_agg_export_states(chgpu_agg * a, chgpu_col ** keys_out, chgpu_col ** state_cols, uint64_t * groups)
{
    ChgpuDeviceGuard _dev_guard(a ? a->ctx : nullptr);
    CHGPU_REQUIRE(a && state_cols && groups, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    return agg_export(a, keys_out, state_cols, groups);
}

// Two-level form of the partial states: rows ordered by the reference's bucket number, so that block b of a CPU initiator's
// MergingAggregatedMemoryEfficientTransform is a row range.  (The device table itself stays single-level)
extern "C" int chgpu_agg_export_states_two_level(chgpu_agg * a, chgpu_col ** keys_out, chgpu_col ** state_cols, uint64_t * groups, uint64_t * bucket_counts)
{
    ChgpuDeviceGuard _dev_guard(a ? a->ctx : nullptr);
    CHGPU_REQUIRE(a && keys_out && state_cols && groups && bucket_counts, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(a->key_type >= 0, CHGPU_ERR_BAD_ARGUMENTS, "an aggregation without key has no buckets");
    chgpu_col * keys = nullptr;
    chgpu_col * words[AGG_MAX_WORDS] = {nullptr};
    u64 n = 0;
    CHGPU_TRY(agg_export(a, &keys, words, &n));
    const u32 nw = a->n_pub_words;
    chgpu_col * sorted[AGG_MAX_WORDS + 1] = {nullptr};
    int rc = CHGPU_OK;
    // <= 8 columns per partition call; the key column rides in the first
    for (u32 lo = 0; lo < nw + 1 && rc == CHGPU_OK; lo += 8)
    {
        const chgpu_col * in[8];
        chgpu_col * out[8] = {nullptr};
        u32 k = 0;
        for (u32 c = lo; c < nw + 1 && k < 8; ++c, ++k)
            in[k] = c == 0 ? keys : words[c - 1];
        rc = chgpu_partition_by_hash(a->ctx, keys, 256, k, in, out, bucket_counts);
        for (u32 j = 0; j < k && rc == CHGPU_OK; ++j)
            sorted[lo + j] = out[j];
    }
    chgpu_col_free(keys);
    for (u32 w = 0; w < nw; ++w)
        chgpu_col_free(words[w]);
    if (rc != CHGPU_OK)
    {
        for (u32 c = 0; c < nw + 1; ++c)
            if (sorted[c])
                chgpu_col_free(sorted[c]);
        return rc;
    }
    *keys_out = sorted[0];
    for (u32 w = 0; w < nw; ++w)
        state_cols[w] = sorted[w + 1];
    *groups = n;
    return CHGPU_OK;
}

// min / max state words (order keys; complemented for min) -> values of the argument's type
__global__ __launch_bounds__(256) void k_extremum_decode(const u64 * __restrict__ words, u64 n, int type, int is_min, void * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    {
        const u64 bits = is_min == 2 ? words[i] : agg_order_key_inverse(is_min ? ~words[i] : words[i], type);
        switch (type)
        {
            case CHGPU_I64: case CHGPU_U64: case CHGPU_F64: ((u64 *)out)[i] = bits; break;
            case CHGPU_U32: case CHGPU_I32: ((u32 *)out)[i] = (u32)bits; break;
            case CHGPU_U16: case CHGPU_I16: ((u16 *)out)[i] = (u16)bits; break;
            case CHGPU_U8: case CHGPU_I8: ((u8 *)out)[i] = (u8)bits; break;
            case CHGPU_F32: ((float *)out)[i] = (float)__longlong_as_double((long long)bits); break; // the widening was exact: so is this
            default: break;
        }
    }
}

extern "C" int chgpu_agg_finalize(chgpu_agg * a, chgpu_col ** keys_out, chgpu_col ** res_cols, uint64_t * groups)
{
    ChgpuDeviceGuard _dev_guard(a ? a->ctx : nullptr);
    CHGPU_REQUIRE(a && res_cols && groups, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    chgpu_ctx * ctx = a->ctx;
    chgpu_col * words[AGG_MAX_WORDS] = {nullptr};
    u64 n = 0;
    CHGPU_TRY(agg_export(a, keys_out, words, &n));
    int rc = CHGPU_OK;
    for (u32 j = 0; j < a->n_aggs; ++j)
    {
        const u32 w = a->word_off[j];
        if (a->kinds[j] == CHGPU_AGG_COUNT)
        {
            words[w]->type = CHGPU_U64;
            res_cols[j] = words[w];
            words[w] = nullptr;
        }
        else if (a->kinds[j] == CHGPU_AGG_SUM)
        {
            words[w]->type = chgpu_sum_result_type(a->arg_types[j]); // SumSimple: Int64 / UInt64 / Float64
            res_cols[j] = words[w];
            words[w] = nullptr;
        }
        else if (a->kinds[j] == CHGPU_AGG_MIN || a->kinds[j] == CHGPU_AGG_MAX || a->kinds[j] == CHGPU_AGG_ANY)
        {
            // insertResultInto: the value itself, in the argument's type (AggregateFunctionsMinMax.cpp)
            chgpu_col * r = nullptr;
            rc = chgpu_col_new(ctx, a->arg_types[j], n, &r);
            if (rc != CHGPU_OK)
                break;
            if (a->key_type < 0 && a->nokey_kept == 0)
                CHGPU_HIP(hipMemsetAsync(r->data, 0, chgpu_type_size(a->arg_types[j]), ctx->stream)); // a state without a value: the type's default
            else if (n)
            {
                // (any: the value word, as loaded -- no order key to undo)
                hipLaunchKernelGGL(k_extremum_decode, dim3(chgpu_grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream,
                                   (const u64 *)words[a->kinds[j] == CHGPU_AGG_ANY ? w + 1 : w]->data, n, a->arg_types[j],
                                   a->kinds[j] == CHGPU_AGG_MIN ? 1 : a->kinds[j] == CHGPU_AGG_ANY ? 2 : 0, r->data);
                ctx->counters[6] += 1;
            }
            res_cols[j] = r;
        }
        else
        {
            chgpu_col * r = nullptr;
            rc = chgpu_col_new(ctx, CHGPU_F64, n, &r);
            if (rc != CHGPU_OK)
                break;
            if (n)
            {
                hipLaunchKernelGGL(k_avg_divide, dim3(chgpu_grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, (const u64 *)words[w]->data,
                                   (const u64 *)words[w + 1]->data, n, chgpu_sum_result_type(a->arg_types[j]), (double *)r->data);
                ctx->counters[6] += 1;
            }
            res_cols[j] = r;
        }
    }
    for (u32 w = 0; w < a->n_pub_words; ++w)
        if (words[w])
            chgpu_col_free(words[w]); // pooled: reuse is stream-ordered behind k_avg_divide
    *groups = n;
    return rc;
}
