// join_chain.h — a chain of filter-form joins probed in ONE sweep over the left Block (included at the end of join_kernels.hip).
//
// Reference shape replaced: a star join is a pipeline of JoiningTransforms, each running HashJoin::joinBlock -> joinRightColumns
// (src/Interpreters/HashJoin/HashJoinMethodsImpl.h:68-202) and then `block.filter(filter)` (:122-123) over EVERY left column, so the
// fact columns are copied once per join.  When every join of the chain is of the filter form (need_filter, JoinFeatures.h:32, or ALL over a
// build side without duplicate keys: at most one right row per left row), the surviving left rows are exactly the AND of the joins' filters
// -- no join changes the multiplicity of a row -- so the chain can be answered before any left column is touched (late materialisation):
// the surviving row numbers (`filterToIndices`, src/Columns/FilterDescription.cpp:113-116), the matched right row of every join that adds
// columns, and the left columns gathered at the survivors (`IColumn::index`) come out of one pass over the survivors only.
//
// Kernels:
//   k_chain_lds   steps whose key set is an exact dense bitmap of <= 4 LDS slices (dimension surrogate keys): a workgroup takes a PART of
//                 64 Ki rows; a thread keeps its 64 keys of the step's column in registers (16 x 16-byte loads in flight), tests them
//                 against every slice of the step's bitmap staged in LDS (a random bit read served by LDS, not by 64 separate L2 requests
//                 per wave instruction), and while it tests the last slice it reloads every register it is done with from the NEXT step's
//                 column (or the next part's first column), so the memory pipe stays busy through the LDS reloads.  Output: one 64-bit
//                 word of alive bits per thread and part -- 1 bit per row instead of a filter byte.
//   k_chain_tail  the remaining steps (bitmaps too large for LDS, hash tables) over the rows still alive: a wave compacts the alive rows
//                 of a UNIT (4096 consecutive rows = the 64 words one wave of k_chain_lds wrote) into an LDS queue and runs the steps over
//                 dense lanes, several rows per lane in flight.  Output: the chain's result as a plain bitmask over the rows (bit r of
//                 word r / 64) + the number of survivors per unit.
//   k_chain_indexes  bitmask + scanned unit counts -> ascending row numbers (and the filter bytes when asked for)
//   k_chain_gather   one thread per survivor: the matched right row of every join that adds columns, the left columns gathered there
#pragma once

static constexpr u32 JC_MAX_STEPS = 8;
static constexpr u32 JC_MAX_CARRY = 8;
static constexpr u32 JC_SLICE_BYTES = 142u * 1024u;
static constexpr u32 JC_SLICE_BITS = JC_SLICE_BYTES * 8u;
static constexpr u32 JC_MAX_SLICES = 4;
static constexpr u32 JC_QPT = 16;                  // quads (of four rows) per thread and part
static constexpr u32 JC_THREADS = 1024;
static constexpr u32 JC_PART_Q = JC_THREADS * JC_QPT; // quads per part
static constexpr u32 JC_PART_ROWS = JC_PART_Q * 4;
static constexpr u32 JC_UNIT_ROWS = 64 * JC_QPT * 4;  // rows of one wave of a part: 4096 consecutive rows
static constexpr u32 JC_UNITS_PER_PART = JC_THREADS / 64;

typedef u32 jc_v4u __attribute__((ext_vector_type(4)));

struct ChainLdsStep
{
    const u32 * keys;    // left key column (4-byte keys, 16-byte aligned)
    const u8 * null_map; // or NULL (4-byte aligned)
    const u32 * pf;      // exact dense bitmap of the build keys
    u32 dense_bits;      // multiple of 32, > max key; bits beyond the largest key are zero
    u32 n_slices;
    int anti;
    int has_zero;
};
struct ChainLdsArgs
{
    ChainLdsStep s[JC_MAX_STEPS];
    u32 n_steps;
};

struct ChainTailStep
{
    const void * keys;
    const u8 * null_map;
    const u32 * pf;   // prefilter words or NULL
    const u64 * kv;   // {key, value} cells of the hash table
    u64 pf_mask;
    u64 max_key;
    u64 capacity;
    int key_type;
    int anti;
    int has_zero;
    int dense;        // pf is an exact bitmap over [0, max_key]
    const u32 * dense_row; // or NULL: dense_row[key] = the build row of the key (one build block), 0xFFFFFFFF = absent -- instead of kv
};
struct ChainTailArgs
{
    ChainTailStep s[JC_MAX_STEPS];
    u32 n_steps;    // the filtering steps, LDS steps first
    u32 first_tail; // steps [first_tail, n_steps) are probed by k_chain_tail for units k_chain_lds covered; all of them for the others
};

// bit `rel` of the staged slice; rel >= nb (a key outside the slice, or below it: the subtraction wrapped) reads the zero pad word behind it
__device__ __forceinline__ u32 jc_bit(const u32 * __restrict__ bits, u32 rel, u32 nb)
{
    const u32 r = rel < nb ? rel : nb;
    return (bits[r >> 5] >> (r & 31u)) & 1u;
}

// A part's keys are fetched with buffer loads: the part's base lives in a scalar resource descriptor, a lane contributes ONE 32-bit offset
// register for all sixteen loads and the quad's offset is a scalar operand.  (With global loads hipcc hoists the sixteen addresses out of
// the slice loop as sixteen per-lane 64-bit pointers, which spills the key registers.)
// Row mapping inside a part: wave w, lane l, quad kb -> quad w * 1024 + kb * 64 + l: a wave owns 4096 consecutive rows (a UNIT), one load
// instruction of a wave covers 1 KiB of consecutive addresses.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t jc_part_rsrc(const void * part_base)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)part_base, 0, (int)(JC_PART_ROWS * 4u), 0x00020000);
}
__device__ __forceinline__ u32 jc_lane_quad() { return (threadIdx.x >> 6) * (64u * JC_QPT) + (threadIdx.x & 63u); }
__device__ __forceinline__ jc_v4u jc_load_quad(__amdgpu_buffer_rsrc_t rsrc, u32 kb)
{
    return __builtin_bit_cast(jc_v4u, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(jc_lane_quad() * 16u), (int)(kb * 64u * 16u), 0));
}

template <bool LAST>
__device__ __forceinline__ void jc_slice_pass(jc_v4u (&kk)[JC_QPT], u32 & f_lo, u32 & f_hi, const u32 * __restrict__ bits, u32 lo, u32 nb,
                                              __amdgpu_buffer_rsrc_t next)
{
#pragma unroll
    for (u32 kb = 0; kb < JC_QPT; ++kb)
    {
        const jc_v4u k = kk[kb];
        const u32 h = jc_bit(bits, k.x - lo, nb) | (jc_bit(bits, k.y - lo, nb) << 1) | (jc_bit(bits, k.z - lo, nb) << 2) | (jc_bit(bits, k.w - lo, nb) << 3);
        if (kb < 8)
            f_lo |= h << (4 * kb);
        else
            f_hi |= h << (4 * (kb - 8));
        if constexpr (LAST)
        {
            kk[kb] = jc_load_quad(next, kb); // this register is done with the step: refill it from the next column
        }
    }
}

__global__ __launch_bounds__(JC_THREADS) void k_chain_lds(ChainLdsArgs a, u64 n, u64 * __restrict__ alive_words)
{
    extern __shared__ __attribute__((aligned(16))) u32 jc_bits[];
    // whole parts only: the rows behind the last whole part go through k_chain_tail's generic path, so no load here needs a clamp
    const u64 n_parts = n / JC_PART_ROWS;
    if (blockIdx.x >= n_parts)
        return;
    const u32 L = a.n_steps;
    jc_v4u kk[JC_QPT];
    {
        const __amdgpu_buffer_rsrc_t first = jc_part_rsrc((const jc_v4u *)a.s[0].keys + (u64)blockIdx.x * JC_PART_Q);
#pragma unroll
        for (u32 kb = 0; kb < JC_QPT; ++kb)
            kk[kb] = jc_load_quad(first, kb);
    }
    u32 loaded = ~0u;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // stage slice `sl` of step `st`'s bitmap (+ a zero pad word behind it).  Whole KiB pieces go global -> LDS directly (LDS-DMA: one
    // instruction per wave and KiB, no register round trip, no ds_write issue); the ragged end and the pad word are written by hand.
    auto stage = [&](const ChainLdsStep & st, u32 id, u32 sl, u32 lo, u32 nb) {
        if (loaded == id)
            return;
        __syncthreads(); // every wave has finished probing the slice that is about to be replaced
        const u32 n_words = nb / 32; // (dense_bits and JC_SLICE_BITS are multiples of 32)
        const u32 * src = st.pf + lo / 32;
        const u32 full = n_words / 256;
        for (u32 c = wave; c < full; c += JC_THREADS / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + c * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)(jc_bits + c * 256), 16, 0, 0);
        for (u32 x = full * 256 + threadIdx.x; x < n_words + 1; x += JC_THREADS)
            jc_bits[x] = x < n_words ? src[x] : 0u; // the pad word answers every key outside the slice
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): this wave's pieces have landed (the keys prefetched before them too)
        if (sl == 0 && threadIdx.x == 0)    // the zero key lives out of line (HashTable.h:874-898): bit 0 stands for it
            jc_bits[0] = (jc_bits[0] & ~1u) | (st.has_zero ? 1u : 0u);
        __syncthreads();
        loaded = id;
    };
    u32 iter = 0;
    for (u64 part = blockIdx.x; part < n_parts; part += gridDim.x, ++iter)
    {
        // odd turns walk the steps and the slices backwards: the slice a part ends with is the one the next part starts with (one stage saved
        // per part); the AND over the steps does not care about their order
        const bool rev = iter & 1u;
        const bool more = part + gridDim.x < n_parts;
        u32 alive_lo = ~0u, alive_hi = ~0u;
        for (u32 si = 0; si < L; ++si)
        {
            const u32 s = rev ? L - 1 - si : si;
            const ChainLdsStep & st = a.s[s];
            u32 f_lo = 0, f_hi = 0;
            const bool last_step = si + 1 == L;
            // what the key registers are refilled with during the step's last pass: the next step of this part, or the first step of the
            // next part (which walks the other way); nothing left: this part's column again, harmlessly
            const u32 nx_s = !last_step ? (rev ? s - 1 : s + 1) : (more ? (rev ? 0u : L - 1) : s);
            const u64 nx_part = (last_step && more) ? part + gridDim.x : part;
            const __amdgpu_buffer_rsrc_t next_base = jc_part_rsrc((const jc_v4u *)a.s[nx_s].keys + nx_part * JC_PART_Q);
            // (two separate loops, not one loop with an if / else around the two pass flavours: hipcc hoists the code the flavours share --
            //  64 subtractions -- above the branch and the keys no longer fit the register file)
            u32 k = 0;
            for (; k + 1 < st.n_slices; ++k)
            {
                const u32 sl = rev ? st.n_slices - 1 - k : k;
                const u32 lo = sl * JC_SLICE_BITS;
                const u32 nb = st.dense_bits - lo < JC_SLICE_BITS ? st.dense_bits - lo : JC_SLICE_BITS;
                stage(st, s * JC_MAX_SLICES + sl, sl, lo, nb);
                jc_slice_pass<false>(kk, f_lo, f_hi, jc_bits, lo, nb, next_base);
            }
            {
                const u32 sl = rev ? st.n_slices - 1 - k : k;
                const u32 lo = sl * JC_SLICE_BITS;
                const u32 nb = st.dense_bits - lo < JC_SLICE_BITS ? st.dense_bits - lo : JC_SLICE_BITS;
                stage(st, s * JC_MAX_SLICES + sl, sl, lo, nb);
                jc_slice_pass<true>(kk, f_lo, f_hi, jc_bits, lo, nb, next_base);
            }
            if (st.null_map)
            {
                // a NULL key matches nothing (HashJoinMethodsImpl.h:451-452)
                const u32 * nm = (const u32 *)st.null_map + part * JC_PART_Q + jc_lane_quad();
#pragma unroll 4
                for (u32 kb = 0; kb < JC_QPT; ++kb)
                {
                    const u32 w = nm[kb * 64u];
                    const u32 nul = ((w & 0xffu) ? 1u : 0u) | ((w & 0xff00u) ? 2u : 0u) | ((w & 0xff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u);
                    if (kb < 8)
                        f_lo &= ~(nul << (4 * kb));
                    else
                        f_hi &= ~(nul << (4 * (kb - 8)));
                }
            }
            if (st.anti) // :515-519, :535-536
            {
                f_lo = ~f_lo;
                f_hi = ~f_hi;
            }
            alive_lo &= f_lo;
            alive_hi &= f_hi;
        }
        alive_words[part * JC_THREADS + threadIdx.x] = ((u64)alive_hi << 32) | alive_lo;
    }
}

// slot of `key` in the step's hash table (capacity = the zero key's out-of-line cell), NO_SLOT when absent
__device__ __forceinline__ u32 jc_find_slot(const ChainTailStep & st, u64 key)
{
    if (key == 0)
        return st.has_zero ? (u32)st.capacity : NO_SLOT;
    const u64 mask = st.capacity - 1;
    u64 slot = dev_intHash64(key) & mask;
    for (u64 step = 0; step < st.capacity; ++step)
    {
        const u64 k = st.kv[2 * slot];
        if (k == key)
            return (u32)slot;
        if (k == 0)
            return NO_SLOT;
        slot = (slot + 1) & mask;
    }
    return NO_SLOT;
}

__device__ __forceinline__ bool jc_tail_found(const ChainTailStep & st, u64 key)
{
    if (key == 0)
        return st.has_zero != 0;
    if (st.dense)
        return key <= st.max_key && ((st.pf[key >> 5] >> (key & 31)) & 1u);
    if (st.pf)
    {
        const u64 pos = ((key * 0x9E3779B97F4A7C15ull) >> 32) & st.pf_mask;
        if (!((st.pf[pos >> 5] >> (pos & 31)) & 1u))
            return false;
    }
    return jc_find_slot(st, key) != NO_SLOT;
}

static constexpr u32 JCT_THREADS = 256, JCT_WAVES = JCT_THREADS / 64, JCT_U = 4;
static constexpr u32 JCT_QUEUE = 1024;   // queue entries per wave: a quarter of a unit (more alive rows than that: four passes, quarter by quarter)
static constexpr u32 JCT_COAL_MIN = 96;  // alive rows of a unit from which a dense-bitmap step reads the unit's keys whole instead of row by row

__device__ __forceinline__ u32 jc_wave_sum(u32 v)
{
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1)
        v += __shfl_xor(v, dlt, 64);
    return v;
}

// One dense-bitmap step over a whole unit whose keys are 4 bytes wide: the unit's 16 KiB of keys are read with full-width loads (a lane
// takes its own sixteen quads, as in k_chain_lds), and only the bitmap words are gathered -- for the alive rows; a dead row reads word 0, so
// every load is unconditional and a wave instruction costs the cache a line per alive lane.  (Row by row, the keys of the alive rows are
// gathers too: at 8 % alive they touch nearly every line of the unit anyway, through 64 separate requests per instruction.)
__device__ __forceinline__ u64 jc_tail_step_coalesced(const ChainTailStep & st, u64 word, u64 row0, u32 lane)
{
    const jc_v4u * kq = (const jc_v4u *)st.keys + row0 / 4 + lane;
    const u32 * nq = st.null_map ? (const u32 *)st.null_map + row0 / 4 + lane : nullptr;
    const u32 max_key = (u32)st.max_key;
#pragma unroll 1
    for (u32 g = 0; g < 4; ++g)
    {
        jc_v4u k[4];
        u32 nm[4];
#pragma unroll
        for (u32 j = 0; j < 4; ++j)
        {
            k[j] = __builtin_nontemporal_load(kq + (4 * g + j) * 64);
            nm[j] = nq ? nq[(4 * g + j) * 64] : 0u;
        }
        const u32 bits = (u32)(word >> (16 * g)) & 0xffffu;
        u32 w[16];
#pragma unroll
        for (u32 j = 0; j < 4; ++j)
        {
            const u32 kv4[4] = {k[j].x, k[j].y, k[j].z, k[j].w};
#pragma unroll
            for (u32 b = 0; b < 4; ++b)
            {
                const bool look = ((bits >> (4 * j + b)) & 1u) && kv4[b] <= max_key;
                w[4 * j + b] = st.pf[(look ? kv4[b] : 0u) >> 5];
            }
        }
        u32 out = 0;
#pragma unroll
        for (u32 j = 0; j < 4; ++j)
        {
            const u32 kv4[4] = {k[j].x, k[j].y, k[j].z, k[j].w};
#pragma unroll
            for (u32 b = 0; b < 4; ++b)
            {
                const u32 key = kv4[b];
                u32 found = key <= max_key ? (w[4 * j + b] >> (key & 31u)) & 1u : 0u;
                found = key == 0 ? (st.has_zero ? 1u : 0u) : found;          // the zero key lives out of line (HashTable.h:874-898)
                found = ((nm[j] >> (8 * b)) & 0xffu) ? 0u : found;           // HashJoinMethodsImpl.h:451-452
                out |= (st.anti ? found ^ 1u : found) << (4 * j + b);        // :515-519, :535-536
            }
        }
        word &= ~(0xffffull << (16 * g)) | ((u64)(bits & out) << (16 * g));
    }
    return word;
}

// alive_words: one word per thread and WHOLE part of k_chain_lds (n_lds_units = 16 x parts units), or NULL when no LDS step ran.  Units
// behind the last whole part start with every row alive and take every step.  mask_words[r / 64] bit r % 64 = row r survives the chain.
__global__ __launch_bounds__(JCT_THREADS) void k_chain_tail(ChainTailArgs a, const u64 * __restrict__ alive_words, u64 n_lds_units, u64 n, u64 * __restrict__ mask_words,
                                                            u32 * __restrict__ unit_counts)
{
    __shared__ u16 queue[JCT_WAVES][JCT_QUEUE];
    __shared__ u32 surv[JCT_WAVES][JC_UNIT_ROWS / 32];
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u64 n_units = (n + JC_UNIT_ROWS - 1) / JC_UNIT_ROWS;
    u16 * q = queue[wv];
    u32 * sv = surv[wv];
    for (u64 unit = (u64)blockIdx.x * JCT_WAVES + wv; unit < n_units; unit += (u64)gridDim.x * JCT_WAVES)
    {
        const u64 row0 = unit * JC_UNIT_ROWS;
        const bool from_lds = alive_words && unit < n_lds_units;
        u32 s = from_lds ? a.first_tail : 0u;
        u64 word = from_lds ? alive_words[unit * 64 + lane] : ~0ull; // bit 4 * kb + b: row (kb * 64 + lane) * 4 + b of the unit
        const bool whole = row0 + JC_UNIT_ROWS <= n;
        if (!whole)
        {
            // rows beyond the end do not exist (the last unit only: a rolled loop keeps the common path's registers low)
#pragma unroll 1
            for (u32 bit = 0; bit < 64; ++bit)
                if (row0 + ((bit >> 2) * 64 + lane) * 4 + (bit & 3) >= n)
                    word &= ~(1ull << bit);
        }
        // steps over an exact bitmap while many rows are alive: the unit's keys read whole
        u32 total = jc_wave_sum((u32)__popcll(word));
        while (s < a.n_steps && whole && total >= JCT_COAL_MIN && a.s[s].dense && (a.s[s].key_type == CHGPU_U32 || a.s[s].key_type == CHGPU_I32)
               && ((uintptr_t)a.s[s].keys & 15) == 0 && ((uintptr_t)a.s[s].null_map & 3) == 0)
        {
            word = jc_tail_step_coalesced(a.s[s], word, row0, lane);
            total = jc_wave_sum((u32)__popcll(word));
            ++s;
        }
        sv[lane] = 0;
        sv[lane + 64] = 0;
        if (s < a.n_steps)
        {
            // the other steps: the alive rows compacted into a queue (entry = the row's offset inside the unit), dense lanes, JCT_U rows per
            // lane in flight.  More alive rows than the queue holds: quarter by quarter.
            const u32 n_pass = total <= JCT_QUEUE ? 1u : 4u;
            for (u32 pass = 0; pass < n_pass; ++pass)
            {
                const u64 wpart = n_pass == 1 ? word : (word >> (16 * pass)) & 0xffffull;
                const u32 bit0 = n_pass == 1 ? 0u : 16u * pass;
                const u32 c = (u32)__popcll(wpart);
                u32 incl = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1)
                {
                    const u32 o = __shfl_up(incl, d, 64);
                    if ((int)lane >= d)
                        incl += o;
                }
                const u32 ptotal = __shfl(incl, 63, 64);
                u32 pos = incl - c;
                for (u64 w = wpart; w; w &= w - 1)
                {
                    const u32 b = bit0 + (u32)__ffsll((unsigned long long)w) - 1;
                    q[pos++] = (u16)(((b >> 2) * 64 + lane) * 4 + (b & 3));
                }
                // (LDS operations of one wave complete in order: no barrier between the queue writes and the reads below)
                for (u32 i0 = 0; i0 < ptotal; i0 += 64 * JCT_U)
                {
                    u32 e[JCT_U];
                    bool al[JCT_U];
#pragma unroll
                    for (u32 u = 0; u < JCT_U; ++u)
                    {
                        const u32 idx = i0 + u * 64 + lane;
                        al[u] = idx < ptotal;
                        e[u] = al[u] ? q[idx] : 0;
                    }
                    for (u32 t = s; t < a.n_steps; ++t)
                    {
                        const ChainTailStep & st = a.s[t];
                        u64 key[JCT_U];
                        bool ok[JCT_U];
#pragma unroll
                        for (u32 u = 0; u < JCT_U; ++u)
                        {
                            key[u] = al[u] ? jload_key(st.keys, st.key_type, row0 + e[u]) : 0;
                            ok[u] = al[u] && !(st.null_map && st.null_map[row0 + e[u]]); // HashJoinMethodsImpl.h:451-452
                        }
#pragma unroll
                        for (u32 u = 0; u < JCT_U; ++u)
                            if (al[u])
                            {
                                const bool found = ok[u] && jc_tail_found(st, key[u]);
                                al[u] = st.anti ? !found : found;
                            }
                    }
#pragma unroll
                    for (u32 u = 0; u < JCT_U; ++u)
                        if (al[u])
                            atomicOr(&sv[e[u] >> 5], 1u << (e[u] & 31u));
                }
            }
        }
        else
        {
            // no step left: the thread-layout word goes to the row-major bitmap as it is, nibble by nibble
#pragma unroll
            for (u32 kb = 0; kb < JC_QPT; ++kb)
            {
                const u32 nib = (u32)(word >> (4 * kb)) & 15u;
                const u32 bit = (kb * 64 + lane) * 4;
                if (nib)
                    atomicOr(&sv[bit >> 5], nib << (bit & 31u));
            }
        }
        const u64 out = ((u64)sv[2 * lane + 1] << 32) | sv[2 * lane]; // rows row0 + 64 * lane ... + 63
        mask_words[unit * 64 + lane] = out;
        const u32 cnt = jc_wave_sum((u32)__popcll(out));
        if (lane == 0)
            unit_counts[unit] = cnt;
    }
}

struct ChainEmitArgs
{
    ChainTailStep s[JC_MAX_STEPS]; // the steps whose matched right row is wanted
    u64 * rowid_out[JC_MAX_STEPS];
    // or, instead of the row id, the value of one right column at the matched row (one build block: row = the id's low 32 bits; a miss
    // gives the type's default 0, insertDefault): payload_in != NULL
    const void * payload_in[JC_MAX_STEPS];
    void * payload_out[JC_MAX_STEPS];
    u32 payload_size[JC_MAX_STEPS];
    const void * carry_in[JC_MAX_CARRY];
    void * carry_out[JC_MAX_CARRY];
    u32 carry_size[JC_MAX_CARRY];  // element bytes: 1, 2, 4, 8
    u32 n_rowid;
    u32 n_carry;
};

// bitmask -> the survivors' row numbers (ascending) and, when asked for, the filter bytes.  One wave per unit of 4096 rows (64 mask
// words); unit_offsets = exclusive scan of k_chain_tail's counts.
__global__ __launch_bounds__(JCT_THREADS) void k_chain_indexes(const u64 * __restrict__ mask_words, const u64 * __restrict__ unit_offsets, u64 n,
                                                               u64 * __restrict__ indexes, u8 * __restrict__ filter)
{
    __shared__ u16 queue[JCT_WAVES][JC_UNIT_ROWS];
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u64 n_units = (n + JC_UNIT_ROWS - 1) / JC_UNIT_ROWS;
    u16 * q = queue[wv];
    for (u64 unit = (u64)blockIdx.x * JCT_WAVES + wv; unit < n_units; unit += (u64)gridDim.x * JCT_WAVES)
    {
        const u64 row0 = unit * JC_UNIT_ROWS;
        const u64 word = mask_words[unit * 64 + lane];
        if (filter)
        {
            // this lane's 64 rows -> 64 filter bytes
            const u64 r = row0 + 64 * (u64)lane;
            if (r < n)
            {
#pragma unroll
                for (u32 g = 0; g < 4; ++g)
                {
                    const u32 bits16 = (u32)(word >> (16 * g)) & 0xffffu;
                    u32 w4[4];
#pragma unroll
                    for (u32 x = 0; x < 4; ++x)
                    {
                        const u32 nib = (bits16 >> (4 * x)) & 15u;
                        w4[x] = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
                    }
                    jc_v4u v;
                    v.x = w4[0]; v.y = w4[1]; v.z = w4[2]; v.w = w4[3];
                    if (r + 16 * g + 16 <= n)
                        *(jc_v4u *)(filter + r + 16 * g) = v;
                    else
                        for (u32 x = 0; x < 16 && r + 16 * g + x < n; ++x)
                            filter[r + 16 * g + x] = (u8)((bits16 >> x) & 1u);
                }
            }
        }
        if (!indexes)
            continue;
        const u32 c = (u32)__popcll(word);
        u32 incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1)
        {
            const u32 o = __shfl_up(incl, d, 64);
            if ((int)lane >= d)
                incl += o;
        }
        const u32 total = __shfl(incl, 63, 64);
        if (total == 0)
            continue;
        u32 pos = incl - c;
        for (u64 w = word; w; w &= w - 1)
            q[pos++] = (u16)(64 * lane + (u32)__ffsll((unsigned long long)w) - 1);
        const u64 base = unit_offsets[unit];
        for (u32 i = lane; i < total; i += 64)
            indexes[base + i] = row0 + q[i];
    }
}

// one thread per survivor: the matched right rows and the left columns gathered at the survivor.  Dense lanes, no LDS: the dependent
// reads (key -> cell -> value) are hidden by occupancy.
__global__ __launch_bounds__(JT) void k_chain_gather(ChainEmitArgs a, const u64 * __restrict__ indexes, u64 kept)
{
    for (u64 o = (u64)blockIdx.x * JT + threadIdx.x; o < kept; o += (u64)gridDim.x * JT)
    {
        const u64 row = indexes[o];
        for (u32 cc = 0; cc < a.n_carry; ++cc)
        {
            switch (a.carry_size[cc])
            {
                case 1: ((u8 *)a.carry_out[cc])[o] = ((const u8 *)a.carry_in[cc])[row]; break;
                case 2: ((u16 *)a.carry_out[cc])[o] = ((const u16 *)a.carry_in[cc])[row]; break;
                case 4: ((u32 *)a.carry_out[cc])[o] = ((const u32 *)a.carry_in[cc])[row]; break;
                default: ((u64 *)a.carry_out[cc])[o] = ((const u64 *)a.carry_in[cc])[row]; break;
            }
        }
        for (u32 s = 0; s < a.n_rowid; ++s)
        {
            const ChainTailStep & st = a.s[s];
            u64 rid = NO_ROW;
            if (!st.anti && !(st.null_map && st.null_map[row]) && st.dense_row)
            {
                // dense surrogate keys: the row sits at its key (one 4-byte read in a table the Infinity Cache holds, no hash, no probing)
                const u64 key = jload_key(st.keys, st.key_type, row);
                const u32 r = key <= st.max_key ? st.dense_row[key] : 0xFFFFFFFFu;
                rid = r == 0xFFFFFFFFu ? NO_ROW : (u64)r; // (block 0 << 32 | row)
            }
            else if (!st.anti && !(st.null_map && st.null_map[row]))
            {
                const u64 key = jload_key(st.keys, st.key_type, row);
                if (key == 0)
                {
                    if (st.has_zero)
                        rid = st.kv[2 * st.capacity + 1];
                }
                else
                {
                    const u64 mask = st.capacity - 1;
                    u64 slot = dev_intHash64(key) & mask;
                    for (u64 step = 0; step < st.capacity; ++step)
                    {
                        const jv2 cell = *(const jv2 *)(st.kv + 2 * slot); // {key, value}: one 16-byte read
                        if (cell.x == key)
                        {
                            rid = cell.y;
                            break;
                        }
                        if (cell.x == 0)
                            break;
                        slot = (slot + 1) & mask;
                    }
                }
            }
            if (a.payload_in[s])
            {
                const u64 r = rid == NO_ROW ? 0 : (rid & 0xFFFFFFFFull);
                switch (a.payload_size[s])
                {
                    case 1: ((u8 *)a.payload_out[s])[o] = rid == NO_ROW ? (u8)0 : ((const u8 *)a.payload_in[s])[r]; break;
                    case 2: ((u16 *)a.payload_out[s])[o] = rid == NO_ROW ? (u16)0 : ((const u16 *)a.payload_in[s])[r]; break;
                    case 4: ((u32 *)a.payload_out[s])[o] = rid == NO_ROW ? 0u : ((const u32 *)a.payload_in[s])[r]; break;
                    default: ((u64 *)a.payload_out[s])[o] = rid == NO_ROW ? 0ull : ((const u64 *)a.payload_in[s])[r]; break;
                }
            }
            else
                a.rowid_out[s][o] = rid;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Key set without a hash table.  A SEMI / ANTI step only asks whether the key is present; for a build side of <= 4-byte keys whose
// largest key is below 2^25 the exact bitmap answers that, and building it costs one pass over the build keys instead of the table's
// claim / finalise passes (SSB's supplier and part sides: 0.2-0.3 ms each for the tables, ~0.03 ms for the bitmaps).  The largest key and
// the zero key's presence come from the staging pass of addBlockToJoin (k_join_stage_keys).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(JT) void k_join_keyset_fill(u32 * __restrict__ pf, const u64 * __restrict__ keys, const u8 * __restrict__ valid, u64 n)
{
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
    {
        const u64 k = keys[i];
        if (k && !(valid && !valid[i]))
            atomicOr(&pf[k >> 5], 1u << (k & 31));
    }
}

// CHGPU_OK: j->ks_* / max_key / has_zero are set.  CHGPU_ERR_NOT_IMPLEMENTED (no error text): the key domain does not fit; build the table.
static int join_build_keyset(chgpu_join * j, bool fill = true)
{
    if (j->ks_ready)
        return CHGPU_OK;
    if (chgpu_type_size(j->key_type) > 4)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    chgpu_ctx * ctx = j->ctx;
    j->build_closed = true;
    // the largest key and the zero key's presence were gathered while the keys were staged (k_join_stage_keys)
    u64 r[2] = {j->stats_host[0], j->stats_host[1]};
    if (j->key_stats && !j->stats_known)
        CHGPU_TRY(chgpu_read_back(ctx, j->key_stats, r, sizeof(r)));
    if (r[0] >= (1ull << 25))
        return CHGPU_ERR_NOT_IMPLEMENTED;
    u64 bits = 1ull << 16;
    while (bits <= r[0] + 32)
        bits <<= 1;
    void * m = nullptr;
    CHGPU_TRY(chgpu_pool_alloc(ctx, bits / 8 + 256, &m, &j->ks_class));
    j->ks_pf = (u32 *)m;
    j->ks_bits = bits;
    CHGPU_HIP(hipMemsetAsync(m, 0, bits / 8 + 256, ctx->stream));
    for (const BuildBlock & b : j->blocks)
        if (b.rows && fill) // (!fill: the caller derives the bits from its row map, join_build_dense)
        {
            hipLaunchKernelGGL(k_join_keyset_fill, dim3(chgpu_grid_for(ctx, b.rows, JT, 8)), dim3(JT), 0, ctx->stream, j->ks_pf, (const u64 *)b.keys, (const u8 *)b.valid, b.rows);
            ctx->counters[6] += 1;
        }
    CHGPU_HIP(hipGetLastError());
    j->max_key = r[0];
    j->has_zero = r[1] != 0;
    j->ks_ready = true;
    return CHGPU_OK;
}

// The key set plus the build row of every key: for a chain step that adds right columns over a build side with dense, UNIQUE keys in one
// block (a dimension table under its surrogate key) the bitmap filters and dm_rows[key] names the matched row -- no hash table is built
// (SSB's customer side: 0.46 ms of table build, and a 16-byte cell read in a 0.5 GB table per survivor, become 0.1 ms and a 4-byte read
// in 120 MB).  CHGPU_ERR_NOT_IMPLEMENTED (no error text): not this shape -- build the table.
__global__ __launch_bounds__(JT) void k_join_dense_fill(u32 * __restrict__ dm, const u64 * __restrict__ keys, const u8 * __restrict__ valid, u64 n, u32 * __restrict__ dup)
{
    for (u64 i = (u64)blockIdx.x * JT + threadIdx.x; i < n; i += (u64)gridDim.x * JT)
    {
        if (valid && !valid[i])
            continue;
        if (atomicExch(&dm[keys[i]], (u32)i) != 0xFFFFFFFFu)
            *dup = 1; // a second row with this key: the reference's maps keep both (ALL) or one by a rule (ANY): the table's business
    }
}
// bit k of the key set = "dm[k] names a row": a streaming pass with one ballot per 64 cells instead of one atomicOr per build row
__global__ __launch_bounds__(JT) void k_join_bitmap_from_dense(const u32 * __restrict__ dm, u64 cells, u64 * __restrict__ pf64)
{
    const u64 n64 = (cells + 63) / 64;
    for (u64 w = ((u64)blockIdx.x * JT + threadIdx.x) >> 6; w < n64; w += ((u64)gridDim.x * JT) >> 6)
    {
        const u64 c = w * 64 + (threadIdx.x & 63);
        const u64 b = __ballot(c < cells && dm[c] != 0xFFFFFFFFu);
        if ((threadIdx.x & 63) == 0)
            pf64[w] = b;
    }
}
// the duplicate flag of a pending row map is known: the map becomes usable, or (a second row for some key) goes away -- NOT_IMPLEMENTED
// then, no error text: the table path takes over (the key set stays valid)
static int join_finish_dense(chgpu_join * j, u32 dup)
{
    if (!j->dm_pending)
        return j->dm_ready ? CHGPU_OK : CHGPU_ERR_NOT_IMPLEMENTED;
    j->dm_pending = false;
    if (dup)
    {
        chgpu_pool_free(j->ctx, j->dm_rows, j->dm_class);
        j->dm_rows = nullptr;
        return CHGPU_ERR_NOT_IMPLEMENTED;
    }
    j->dm_ready = true;
    return CHGPU_OK;
}

// defer_dup: the duplicate flag (key_stats[2]) is left on the device and the map marked pending -- a chain looks at the flags of all its
// joins with one read-back (join_finish_dense)
static int join_build_dense(chgpu_join * j, bool defer_dup = false)
{
    if (j->dm_ready || j->dm_pending)
        return CHGPU_OK;
    if (j->blocks.size() != 1 || j->blocks[0].rows >= 0xFFFFFFFFull || !j->key_stats)
        return CHGPU_ERR_NOT_IMPLEMENTED;
    const bool had_keyset = j->ks_ready;
    const int krc = join_build_keyset(j, /*fill*/ false);
    if (krc != CHGPU_OK)
        return krc;
    chgpu_ctx * ctx = j->ctx;
    const u64 cells = j->max_key + 1;
    void * m = nullptr;
    size_t mclass = 0;
    CHGPU_TRY(chgpu_pool_alloc(ctx, cells * 4 + 256, &m, &mclass));
    u32 * dup_dev = (u32 *)(j->key_stats + 2);
    hipError_t e = hipMemsetAsync(m, 0xFF, cells * 4, ctx->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(dup_dev, 0, 8, ctx->stream);
    if (e == hipSuccess)
    {
        const BuildBlock & b = j->blocks[0];
        hipLaunchKernelGGL(k_join_dense_fill, dim3(chgpu_grid_for(ctx, b.rows, JT, 8)), dim3(JT), 0, ctx->stream, (u32 *)m, (const u64 *)b.keys, (const u8 *)b.valid, b.rows, dup_dev);
        ctx->counters[6] += 1;
        if (!had_keyset)
        {
            // the key set's bits (the bitmap is 8-byte aligned and padded: ks_bits is a multiple of 64); a set does not mind duplicate keys
            hipLaunchKernelGGL(k_join_bitmap_from_dense, dim3(chgpu_grid_for(ctx, cells, JT, 8)), dim3(JT), 0, ctx->stream, (const u32 *)m, cells, (u64 *)j->ks_pf);
            ctx->counters[6] += 1;
        }
        e = hipGetLastError();
    }
    if (e != hipSuccess)
    {
        chgpu_pool_free(ctx, m, mclass);
        return chgpu_set_error(CHGPU_ERR_DEVICE, "dense join map: %s", hipGetErrorString(e));
    }
    j->dm_rows = (u32 *)m;
    j->dm_class = mclass;
    j->dm_pending = true;
    if (defer_dup)
        return CHGPU_OK;
    u32 dup = 0;
    CHGPU_TRY(chgpu_read_back(ctx, dup_dev, &dup, 4));
    return join_finish_dense(j, dup);
}

/* See include/chgpu.h.  right_payload_cols[s] != NULL (with want_right_rows[s]): right_rowid_u64[s] receives that column's values at the matched
   rows instead of the row ids. */
static int join_chain_impl(uint32_t n_steps, chgpu_join * const * joins, const chgpu_col * const * key_cols, const chgpu_col * const * null_maps,
                           const int * want_right_rows, const chgpu_col * const * right_payload_cols, uint32_t n_carry, const chgpu_col * const * carry_cols,
                           chgpu_col ** indexes_u64, chgpu_col ** right_rowid_u64, chgpu_col ** carry_out, chgpu_col ** filter_u8, uint64_t * n_kept)
{
    CHGPU_REQUIRE(n_steps >= 1 && joins && key_cols && n_kept, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(n_steps <= JC_MAX_STEPS, CHGPU_ERR_NOT_IMPLEMENTED, "a join chain of %u steps (at most %u)", n_steps, JC_MAX_STEPS);
    CHGPU_REQUIRE(n_carry <= JC_MAX_CARRY, CHGPU_ERR_NOT_IMPLEMENTED, "%u carried columns (at most %u per call)", n_carry, JC_MAX_CARRY);
    CHGPU_REQUIRE(n_carry == 0 || (carry_cols && carry_out), CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    for (u32 s = 0; s < n_steps; ++s)
        CHGPU_REQUIRE(joins[s] && key_cols[s], CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    chgpu_ctx * ctx = joins[0]->ctx;
    ChgpuDeviceGuard _dev_guard(ctx);
    *n_kept = 0;
    if (indexes_u64) *indexes_u64 = nullptr;
    if (filter_u8) *filter_u8 = nullptr;
    for (u32 s = 0; s < n_steps && right_rowid_u64; ++s)
        right_rowid_u64[s] = nullptr;
    for (u32 c = 0; c < n_carry; ++c)
        carry_out[c] = nullptr;
    const u64 n = key_cols[0]->rows;
    for (u32 s = 0; s < n_steps; ++s)
    {
        chgpu_join * j = joins[s];
        CHGPU_REQUIRE(j->ctx == ctx, CHGPU_ERR_BAD_ARGUMENTS, "the joins of a chain must live on one context");
        CHGPU_REQUIRE(key_cols[s]->type == j->key_type, CHGPU_ERR_BAD_ARGUMENTS, "left key column %u has type %d, expected %d", s, key_cols[s]->type, j->key_type);
        CHGPU_REQUIRE(key_cols[s]->rows == n, CHGPU_ERR_SIZES_MISMATCH, "Size of key column %u doesn't match the chain's (%llu vs %llu rows)", s,
                      (unsigned long long)key_cols[s]->rows, (unsigned long long)n);
        if (null_maps && null_maps[s])
            CHGPU_REQUIRE(null_maps[s]->type == CHGPU_U8 && null_maps[s]->rows == n, CHGPU_ERR_SIZES_MISMATCH, "null map size mismatch");
        // RIGHT / FULL keep per-row used flags and INNER ANY consumes a right row once (setUsedOnce): stateful, not a pure filter
        CHGPU_REQUIRE(!jf_track_used(j), CHGPU_ERR_NOT_IMPLEMENTED, "RIGHT / FULL joins in a chain");
        CHGPU_REQUIRE(!(j->kind == CHGPU_JOIN_INNER && j->strictness == CHGPU_STRICT_ANY), CHGPU_ERR_NOT_IMPLEMENTED, "INNER ANY in a chain");
        if (want_right_rows && want_right_rows[s])
            CHGPU_REQUIRE(right_rowid_u64, CHGPU_ERR_BAD_ARGUMENTS, "right row ids wanted but right_rowid_u64 is NULL");
        if (right_payload_cols && right_payload_cols[s])
        {
            CHGPU_REQUIRE(want_right_rows && want_right_rows[s], CHGPU_ERR_BAD_ARGUMENTS, "a right column for step %u, which adds no right rows", s);
            CHGPU_REQUIRE(j->blocks.size() == 1, CHGPU_ERR_NOT_IMPLEMENTED, "right columns are gathered inside the chain for one-block build sides: take the row ids");
            CHGPU_REQUIRE(right_payload_cols[s]->rows == j->blocks[0].rows, CHGPU_ERR_SIZES_MISMATCH, "Size of the right column of step %u doesn't match the build block", s);
        }
    }
    // The right sides that still have to be built.  Their key statistics (gathered while the keys were staged) come over in ONE read-back,
    // the builds are queued, and the duplicate flags of the row maps come over in a second one: two waits for the whole chain instead of
    // one or two per join.
    {
        chgpu_join * need[JC_MAX_STEPS];
        u32 n_need = 0;
        for (u32 s = 0; s < n_steps; ++s)
        {
            chgpu_join * j = joins[s];
            bool seen = false;
            for (u32 q = 0; q < n_need; ++q)
                seen = seen || need[q] == j;
            if (!seen && !j->finished && !j->ks_ready && !j->stats_known && j->key_stats)
                need[n_need++] = j;
        }
        if (n_need)
        {
            void * stage = nullptr;
            CHGPU_TRY(chgpu_pinned(ctx, 16 * JC_MAX_STEPS, &stage));
            for (u32 q = 0; q < n_need; ++q)
                CHGPU_HIP(hipMemcpyAsync((char *)stage + 16 * q, need[q]->key_stats, 16, hipMemcpyDeviceToHost, ctx->stream));
            CHGPU_HIP(hipStreamSynchronize(ctx->stream));
            for (u32 q = 0; q < n_need; ++q)
            {
                memcpy(need[q]->stats_host, (char *)stage + 16 * q, 16);
                need[q]->stats_known = true;
            }
        }
    }
    bool to_table[JC_MAX_STEPS] = {false};
    for (u32 s = 0; s < n_steps; ++s)
    {
        chgpu_join * j = joins[s];
        if (j->finished)
            continue;
        // a SEMI / ANTI step that adds no column only needs the key SET: the exact bitmap, without the hash table
        int krc = CHGPU_ERR_NOT_IMPLEMENTED;
        if ((j->strictness == CHGPU_STRICT_SEMI || j->strictness == CHGPU_STRICT_ANTI) && !(want_right_rows && want_right_rows[s]))
            krc = join_build_keyset(j);
        else if (want_right_rows && want_right_rows[s] && j->strictness != CHGPU_STRICT_ANTI && !chgpu_opt(ctx, "tune_join_no_dense_map", 0))
            krc = join_build_dense(j, /*defer_dup*/ true); // right rows over dense unique keys: the key set + a direct row map
        if (krc != CHGPU_OK && krc != CHGPU_ERR_NOT_IMPLEMENTED)
            return krc;
        to_table[s] = krc != CHGPU_OK;
    }
    {
        chgpu_join * pend[JC_MAX_STEPS];
        u32 n_pend = 0;
        for (u32 s = 0; s < n_steps; ++s)
        {
            bool seen = false;
            for (u32 q = 0; q < n_pend; ++q)
                seen = seen || pend[q] == joins[s];
            if (!seen && joins[s]->dm_pending)
                pend[n_pend++] = joins[s];
        }
        if (n_pend)
        {
            void * stage = nullptr;
            CHGPU_TRY(chgpu_pinned(ctx, 16 * JC_MAX_STEPS, &stage));
            for (u32 q = 0; q < n_pend; ++q)
                CHGPU_HIP(hipMemcpyAsync((char *)stage + 16 * q, pend[q]->key_stats + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
            CHGPU_HIP(hipStreamSynchronize(ctx->stream));
            for (u32 q = 0; q < n_pend; ++q)
            {
                u32 dup = 0;
                memcpy(&dup, (char *)stage + 16 * q, 4);
                (void)join_finish_dense(pend[q], dup); // (a duplicate key: dm_ready stays false, the table is built below)
            }
        }
    }
    for (u32 s = 0; s < n_steps; ++s)
    {
        chgpu_join * j = joins[s];
        if (!j->finished)
        {
            const bool wants_rows = want_right_rows && want_right_rows[s];
            const bool have = wants_rows ? j->dm_ready : (j->ks_ready && (j->strictness == CHGPU_STRICT_SEMI || j->strictness == CHGPU_STRICT_ANTI));
            if (to_table[s] || !have)
                CHGPU_TRY(join_build_table(j));
        }
        // ALL over duplicate build keys replicates left rows: the chain's result is then not a filter
        CHGPU_REQUIRE(j->strictness != CHGPU_STRICT_ALL || j->unique_keys || (!j->finished && j->dm_ready), CHGPU_ERR_NOT_IMPLEMENTED,
                      "ALL join over duplicate build keys in a chain");
    }
    for (u32 c = 0; c < n_carry; ++c)
        CHGPU_REQUIRE(carry_cols[c] && carry_cols[c]->rows == n, CHGPU_ERR_SIZES_MISMATCH, "Size of carried column %u doesn't match the chain's", c);

    // which steps filter at all (LEFT ANY / LEFT ALL keep every left row), and which of them fit the LDS sweep
    ChainLdsArgs la{};
    ChainTailArgs ta{};
    std::vector<u32> lds_steps, tail_steps;
    for (u32 s = 0; s < n_steps; ++s)
    {
        chgpu_join * j = joins[s];
        const bool filters = jf_left_kind(j) == CHGPU_JOIN_INNER || j->strictness == CHGPU_STRICT_SEMI || j->strictness == CHGPU_STRICT_ANTI;
        if (!filters)
            continue;
        const chgpu_col * nm = null_maps ? null_maps[s] : nullptr;
        const u64 dense_bits = (j->max_key + 32) / 32 * 32;
        const bool dense = j->finished ? (j->t.pf && j->max_key <= j->t.pf_mask) : j->ks_ready;
        const bool lds = dense && chgpu_type_size(j->key_type) == 4 && dense_bits <= (u64)JC_MAX_SLICES * JC_SLICE_BITS && n >= (1u << 20)
            && (uintptr_t)key_cols[s]->data % 16 == 0 && (!nm || (uintptr_t)nm->data % 4 == 0);
        (lds ? lds_steps : tail_steps).push_back(s);
    }
    auto fill_tail = [&](ChainTailStep & t, u32 s) {
        chgpu_join * j = joins[s];
        t.keys = key_cols[s]->data;
        t.null_map = null_maps && null_maps[s] ? (const u8 *)null_maps[s]->data : nullptr;
        t.pf = j->finished ? j->t.pf : j->ks_pf;
        t.kv = j->finished ? j->t.kv : nullptr;
        t.pf_mask = j->finished ? j->t.pf_mask : j->ks_bits - 1;
        t.max_key = j->max_key;
        t.capacity = j->finished ? j->t.capacity : 0;
        t.key_type = j->key_type;
        t.anti = j->strictness == CHGPU_STRICT_ANTI ? 1 : 0;
        t.has_zero = j->has_zero ? 1 : 0;
        t.dense = (j->finished ? (j->t.pf && j->max_key <= j->t.pf_mask) : j->ks_ready) ? 1 : 0;
        t.dense_row = (!j->finished && j->dm_ready) ? j->dm_rows : nullptr;
    };
    for (u32 s : lds_steps)
    {
        chgpu_join * j = joins[s];
        ChainLdsStep & l = la.s[la.n_steps++];
        l.keys = (const u32 *)key_cols[s]->data;
        l.null_map = null_maps && null_maps[s] ? (const u8 *)null_maps[s]->data : nullptr;
        l.pf = j->finished ? j->t.pf : j->ks_pf;
        l.dense_bits = (u32)((j->max_key + 32) / 32 * 32);
        l.n_slices = (l.dense_bits + JC_SLICE_BITS - 1) / JC_SLICE_BITS;
        l.anti = j->strictness == CHGPU_STRICT_ANTI ? 1 : 0;
        l.has_zero = j->has_zero ? 1 : 0;
        fill_tail(ta.s[ta.n_steps++], s);
    }
    ta.first_tail = ta.n_steps;
    for (u32 s : tail_steps)
        fill_tail(ta.s[ta.n_steps++], s);

    chgpu_col * idx = nullptr, * fcol = nullptr;
    chgpu_col * rid[JC_MAX_STEPS] = {nullptr};
    chgpu_col * car[JC_MAX_CARRY] = {nullptr};
    auto fail = [&](int code) {
        chgpu_col_free(idx);
        chgpu_col_free(fcol);
        for (u32 s = 0; s < JC_MAX_STEPS; ++s)
            chgpu_col_free(rid[s]);
        for (u32 c = 0; c < JC_MAX_CARRY; ++c)
            chgpu_col_free(car[c]);
        return code;
    };
    int rc = CHGPU_OK;
    u64 kept = 0;
    const u64 n_units = (n + JC_UNIT_ROWS - 1) / JC_UNIT_ROWS;
    const u64 n_lds_units = la.n_steps ? n / JC_PART_ROWS * JC_UNITS_PER_PART : 0; // k_chain_lds sweeps whole parts; the rest starts alive in k_chain_tail
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t alive_b = al(n_lds_units * 64 * 8), mask_b = al(n_units * 64 * 8), cnt_b = al(n_units * 4 + 4), off_b = al(n_units * 8 + 8), tmp_b = chgpu_scan_tmp_bytes(n_units);
    u64 * mask_words = nullptr, * unit_offsets = nullptr;
    if (n)
    {
        void * scratch = nullptr;
        CHGPU_TRY(chgpu_scratch(ctx, 256 + alive_b + mask_b + cnt_b + off_b + al(tmp_b), &scratch));
        u64 * total_dev = (u64 *)scratch;
        u64 * alive_words = (u64 *)((char *)scratch + 256);
        mask_words = (u64 *)((char *)scratch + 256 + alive_b);
        u32 * unit_counts = (u32 *)((char *)scratch + 256 + alive_b + mask_b);
        unit_offsets = (u64 *)((char *)scratch + 256 + alive_b + mask_b + cnt_b);
        void * tmp = (char *)scratch + 256 + alive_b + mask_b + cnt_b + off_b;
        if (n_lds_units)
        {
            const size_t lds_b = JC_SLICE_BYTES + 16;
            CHGPU_HIP(hipFuncSetAttribute((const void *)k_chain_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
            const u64 n_parts = n_lds_units / JC_UNITS_PER_PART;
            const u32 grid = (u32)(n_parts < (u64)ctx->num_cus ? n_parts : (u64)ctx->num_cus);
            hipLaunchKernelGGL(k_chain_lds, dim3(grid), dim3(JC_THREADS), lds_b, ctx->stream, la, n, alive_words);
            ctx->counters[6] += 1;
        }
        {
            const u64 want = (n_units + JCT_WAVES - 1) / JCT_WAVES;
            const u64 cap = (u64)ctx->num_cus * 4;
            hipLaunchKernelGGL(k_chain_tail, dim3((u32)(want < cap ? want : cap)), dim3(JCT_THREADS), 0, ctx->stream, ta, n_lds_units ? (const u64 *)alive_words : (const u64 *)nullptr,
                               n_lds_units, n, mask_words, unit_counts);
            ctx->counters[6] += 1;
        }
        CHGPU_HIP(hipGetLastError());
        CHGPU_TRY(chgpu_scan_exclusive_u32_u64(ctx, unit_counts, unit_offsets, n_units, total_dev, tmp, tmp_b));
        CHGPU_TRY(chgpu_read_back(ctx, total_dev, &kept, sizeof(kept)));
    }
    // outputs, sized by the survivors
    if (indexes_u64 && (rc = chgpu_col_new(ctx, CHGPU_U64, kept, &idx)) != CHGPU_OK)
        return fail(rc);
    if (filter_u8 && (rc = chgpu_col_new(ctx, CHGPU_U8, n, &fcol)) != CHGPU_OK)
        return fail(rc);
    ChainEmitArgs ea{};
    for (u32 s = 0; s < n_steps; ++s)
        if (want_right_rows && want_right_rows[s])
        {
            const chgpu_col * pay = right_payload_cols ? right_payload_cols[s] : nullptr;
            if ((rc = chgpu_col_new(ctx, pay ? pay->type : CHGPU_U64, kept, &rid[s])) != CHGPU_OK)
                return fail(rc);
            fill_tail(ea.s[ea.n_rowid], s);
            ea.rowid_out[ea.n_rowid] = pay ? nullptr : (u64 *)rid[s]->data;
            ea.payload_in[ea.n_rowid] = pay ? pay->data : nullptr;
            ea.payload_out[ea.n_rowid] = pay ? rid[s]->data : nullptr;
            ea.payload_size[ea.n_rowid] = pay ? (u32)chgpu_type_size(pay->type) : 0;
            ++ea.n_rowid;
        }
    for (u32 c = 0; c < n_carry; ++c)
    {
        if ((rc = chgpu_col_new(ctx, carry_cols[c]->type, kept, &car[c])) != CHGPU_OK)
            return fail(rc);
        ea.carry_in[c] = carry_cols[c]->data;
        ea.carry_out[c] = car[c]->data;
        ea.carry_size[c] = (u32)chgpu_type_size(carry_cols[c]->type);
    }
    ea.n_carry = n_carry;
    const bool gather = kept && (ea.n_rowid || ea.n_carry);
    if (!idx && gather && (rc = chgpu_col_new(ctx, CHGPU_U64, kept, &idx)) != CHGPU_OK) // the gather runs over the index list
        return fail(rc);
    if (n && ((kept && idx) || fcol))
    {
        const u64 want = (n_units + JCT_WAVES - 1) / JCT_WAVES;
        const u64 cap = (u64)ctx->num_cus * 8;
        hipLaunchKernelGGL(k_chain_indexes, dim3((u32)(want < cap ? want : cap)), dim3(JCT_THREADS), 0, ctx->stream, (const u64 *)mask_words, (const u64 *)unit_offsets, n,
                           (kept && idx) ? (u64 *)idx->data : (u64 *)nullptr, fcol ? (u8 *)fcol->data : (u8 *)nullptr);
        ctx->counters[6] += 1;
        if (gather)
        {
            hipLaunchKernelGGL(k_chain_gather, dim3(chgpu_grid_for(ctx, kept, JT, 16)), dim3(JT), 0, ctx->stream, ea, (const u64 *)idx->data, kept);
            ctx->counters[6] += 1;
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return fail(chgpu_set_error(CHGPU_ERR_DEVICE, "join chain launch: %s", hipGetErrorString(e)));
    }
    if (!indexes_u64 && idx)
    {
        chgpu_col_free(idx); // (stream-ordered pool: the kernels above still own the buffer until they finish)
        idx = nullptr;
    }
    for (u32 s = 0; s < n_steps; ++s)
        joins[s]->left_seq += n;
    if (indexes_u64) *indexes_u64 = idx;
    if (filter_u8) *filter_u8 = fcol;
    for (u32 s = 0; s < n_steps; ++s)
        if (rid[s])
            right_rowid_u64[s] = rid[s];
    for (u32 c = 0; c < n_carry; ++c)
        carry_out[c] = car[c];
    *n_kept = kept;
    ctx->counters[3] += n * n_steps;
    ctx->counters[4] += kept;
    return CHGPU_OK;
}

extern "C" int chgpu_join_probe_chain(uint32_t n_steps, chgpu_join * const * joins, const chgpu_col * const * key_cols, const chgpu_col * const * null_maps,
                                      const int * want_right_rows, uint32_t n_carry, const chgpu_col * const * carry_cols, chgpu_col ** indexes_u64,
                                      chgpu_col ** right_rowid_u64, chgpu_col ** carry_out, chgpu_col ** filter_u8, uint64_t * n_kept)
{
    return join_chain_impl(n_steps, joins, key_cols, null_maps, want_right_rows, nullptr, n_carry, carry_cols, indexes_u64, right_rowid_u64, carry_out, filter_u8, n_kept);
}

extern "C" int chgpu_join_probe_chain_columns(uint32_t n_steps, chgpu_join * const * joins, const chgpu_col * const * key_cols, const chgpu_col * const * null_maps,
                                              const int * want_right_rows, const chgpu_col * const * right_cols, uint32_t n_carry, const chgpu_col * const * carry_cols,
                                              chgpu_col ** indexes_u64, chgpu_col ** right_out, chgpu_col ** carry_out, chgpu_col ** filter_u8, uint64_t * n_kept)
{
    return join_chain_impl(n_steps, joins, key_cols, null_maps, want_right_rows, right_cols, n_carry, carry_cols, indexes_u64, right_out, carry_out, filter_u8, n_kept);
}
