// radix_partition.h — the two kernels of an exact (histogram-first) radix partition with carried tails, shared by the GROUP BY
// (agg_kernels.hip: rows -> partitions that fit an LDS table) and the join probe (join_kernels.hip: probe keys -> table regions that
// fit an XCD's L2).  The partition function is a functor so both callers run the same code.
#pragma once
#include "chgpu_internal.h"

static constexpr u32 RP_THREADS = 1024;
static constexpr u32 RP_MAX_P = 1024;
static constexpr u32 RP_SCATTER_SLACK = 12288; // rows of slack k_rp_scatter needs behind each of its output arrays

// Same histogram with 16-byte nontemporal key loads (4- and 8-byte keys whose first row is 16-byte aligned): four loads
// per lane are issued before the first LDS atomic.
template <typename KT, typename PartFn>
__global__ __launch_bounds__(RP_THREADS) void k_rp_hist_wide(const KT * __restrict__ keys, u64 n, u64 rows_per_wg, u32 P, u32 * __restrict__ counts, PartFn part_fn, int gmajor = 0)
{
    constexpr u32 GBP_THREADS = RP_THREADS;
    constexpr u32 GBP_MAX_P = RP_MAX_P;
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    constexpr u32 VEC = 16 / sizeof(KT);
    constexpr int HU = 4;
    __shared__ u32 cnt[GBP_MAX_P];
    for (u32 p = threadIdx.x; p < P; p += GBP_THREADS)
        cnt[p] = 0;
    __syncthreads();
    const u64 r0 = (u64)blockIdx.x * rows_per_wg;
    const u64 r1 = r0 + rows_per_wg < n ? r0 + rows_per_wg : n;
    u64 i = r0;
    constexpr u64 STEP = (u64)HU * GBP_THREADS * VEC;
    for (; i + STEP <= r1; i += STEP)
    {
        v4u v[HU];
#pragma unroll
        for (int q = 0; q < HU; ++q)
            v[q] = __builtin_nontemporal_load((const v4u *)(keys + i + ((u64)q * GBP_THREADS + threadIdx.x) * VEC));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < HU; ++q)
        {
            if constexpr (sizeof(KT) == 4)
            {
                atomicAdd(&cnt[part_fn((KT)v[q].x)], 1u);
                atomicAdd(&cnt[part_fn((KT)v[q].y)], 1u);
                atomicAdd(&cnt[part_fn((KT)v[q].z)], 1u);
                atomicAdd(&cnt[part_fn((KT)v[q].w)], 1u);
            }
            else
            {
                atomicAdd(&cnt[part_fn((KT)((u64)v[q].x | ((u64)v[q].y << 32)))], 1u);
                atomicAdd(&cnt[part_fn((KT)((u64)v[q].z | ((u64)v[q].w << 32)))], 1u);
            }
        }
    }
    for (i += threadIdx.x; i < r1; i += GBP_THREADS)
        atomicAdd(&cnt[part_fn(keys[i])], 1u);
    __syncthreads();
    for (u32 p = threadIdx.x; p < P; p += GBP_THREADS)
        counts[gmajor ? (u64)blockIdx.x * P + p : (u64)p * gridDim.x + blockIdx.x] = cnt[p];
}

// The same pass with the partial lines of every partition run CARRIED from tile to tile in LDS, so that global memory only ever
// sees whole, aligned 16-row pieces of a partition's output (64 B of 4-byte keys, 128 B of 8-byte words): a run of ~32-48 rows
// that starts and ends anywhere leaves a partial line at both ends, the L2 evicts it before the next tile (22 us and 4.7 MB of
// stores per XCD later) completes it, and HBM pays a read-modify-write for it -- 1.5-1.85x the bytes of the rows (PMC WRITE_SIZE).
// Per partition the workgroup keeps: base = the output row where its carried rows start (16-row aligned after the first flush),
// ccnt <= 15 carried rows in carry_key / carry_word.  A tile's sorted rows of partition p continue at base + ccnt; everything
// below the last 16-row boundary is written, the tail becomes the new carry.  Old carries are written by 16 consecutive lanes
// per partition together with the stage rows that complete their line (same tile, microseconds apart: the L2 merges them).
// One 8-byte argument word, WIDE loads only (the shape of config C3); P <= 256.
// dynamic LDS: stage_word u64[TILE] | carry_word u64[P*CG] (both only with HAS_WORD) | stage_key KT[TILE] | carry_key KT[P*CG] |
//              base | delta | obase | ccnt | tile_cnt | tile_off | ocnt, each u32[P]      (rp_scatter_carry_lds_bytes)
// PartFn: u32 operator()(KT key) const -> partition in [0, P).  HAS_WORD = false: keys only (words / out_words unused).
// THREADS x TILE: 1024 x 8192 runs one workgroup per CU; 512 x 4096 with CG = 8 fits two (<= 80 KiB of LDS each), whose memory and
// LDS phases then overlap -- a single workgroup drains its stores (s_waitcnt vmcnt(0)) before every tile.  Output rows are 32-bit
// (the callers bound a call below 2^32 rows).
template <u32 GBP_TILE, typename KT, bool HAS_WORD, typename PartFn, u32 THREADS = RP_THREADS, u32 CG = 16>
__global__ __launch_bounds__(THREADS) void k_rp_scatter_carry(const KT * __restrict__ keys, const u64 * __restrict__ words, u64 n, u64 rows_per_wg, u32 P,
                                                              const u64 * __restrict__ offsets, KT * __restrict__ out_keys, u64 * __restrict__ out_words, PartFn part_fn)
{
    constexpr u32 GBP_THREADS = THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
    u64 * stage_word = (u64 *)gb_lds;
    u64 * carry_word = stage_word + (HAS_WORD ? GBP_TILE : 0);
    KT * stage_key = (KT *)(carry_word + (HAS_WORD ? (size_t)P * CG : 0));
    KT * carry_key = stage_key + GBP_TILE;
    u32 * base = (u32 *)(carry_key + (size_t)P * CG);
    u32 * delta = base + P;
    u32 * obase = delta + P;
    u32 * ccnt = obase + P;
    u32 * tile_cnt = ccnt + P;
    u32 * tile_off = tile_cnt + P;
    u32 * ocnt = tile_off + P;
    __shared__ u32 wave_tot[GBP_THREADS / 64];

    for (u32 p = threadIdx.x; p < P; p += GBP_THREADS)
    {
        base[p] = (u32)offsets[(u64)p * gridDim.x + blockIdx.x];
        ccnt[p] = 0;
        tile_cnt[p] = 0;
    }
    __syncthreads();
    const u64 r0 = (u64)blockIdx.x * rows_per_wg;
    const u64 r1 = r0 + rows_per_wg < n ? r0 + rows_per_wg : n;
    constexpr u32 RPT = GBP_TILE / GBP_THREADS;
    static_assert(RPT % 2 == 0, "row pairs");
    KT key[RPT];
    u64 argw[HAS_WORD ? RPT : 1];
    typedef u64 v2q __attribute__((ext_vector_type(2)));
    typedef u32 v2d __attribute__((ext_vector_type(2)));
    auto row_of = [&](u64 tb, u32 j) -> u64 { return tb + (u64)(j >> 1) * (2 * GBP_THREADS) + 2 * threadIdx.x + (j & 1); };
    auto load_tile = [&](u64 tb) {
#pragma unroll
        for (u32 j = 0; j < RPT; j += 2)
        {
            const u64 i = row_of(tb, j);
            if (i + 1 < r1)
            {
                if constexpr (sizeof(KT) == 4)
                {
                    const v2d kk = __builtin_nontemporal_load((const v2d *)(keys + i));
                    key[j] = kk.x, key[j + 1] = kk.y;
                }
                else
                {
                    const v2q kk = __builtin_nontemporal_load((const v2q *)(keys + i));
                    key[j] = kk.x, key[j + 1] = kk.y;
                }
                if constexpr (HAS_WORD)
                {
                    const v2q a = __builtin_nontemporal_load((const v2q *)(words + i));
                    argw[j] = a.x, argw[j + 1] = a.y;
                }
            }
            else
            {
                const bool in = i < r1;
                key[j] = in ? keys[i] : (KT)0;
                key[j + 1] = 0;
                if constexpr (HAS_WORD)
                {
                    argw[j] = in ? words[i] : 0;
                    argw[j + 1] = 0;
                }
            }
        }
    };
    if (r0 < r1)
        load_tile(r0);
    for (u64 tbase = r0; tbase < r1; tbase += GBP_TILE)
    {
        u32 part[RPT], rank[RPT];
        // 1. a rank inside the tile's partition bucket
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
        {
            part[j] = ~0u;
            if (row_of(tbase, j) < r1)
            {
                part[j] = part_fn(key[j]);
                rank[j] = atomicAdd(&tile_cnt[part[j]], 1u);
            }
        }
        __syncthreads();
        // 2. exclusive scan of tile_cnt[P] -> tile_off[P] (P <= 2 * threads); per partition: where its rows go and what stays
        {
            const u32 e0 = threadIdx.x * 2, e1 = e0 + 1;
            const u32 c0 = e0 < P ? tile_cnt[e0] : 0, c1 = e1 < P ? tile_cnt[e1] : 0;
            const u32 v = c0 + c1;
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
            u32 off = inc - v;
            for (u32 w = 0; w < wave; ++w)
                off += wave_tot[w];
            auto plan = [&](u32 p, u32 toff, u32 nnew) {
                const u32 b = base[p];
                const u32 c = ccnt[p];
                const u32 end = b + c + nnew;
                const u32 fe = end & ~(CG - 1);
                const bool flush = fe > b;
                tile_off[p] = toff;
                delta[p] = b + c - toff;          // stage position -> output row (modulo 2^32: the sum with a position is exact)
                obase[p] = b;
                ocnt[p] = flush ? c : 0;          // old carry rows that leave now (all of them: they sit below fe)
                base[p] = flush ? fe : b;         // rows at or above it stay in LDS as carry slot (row - base)
                ccnt[p] = end - (flush ? fe : b);
                tile_cnt[p] = 0;
            };
            if (e0 < P)
                plan(e0, off, c0);
            if (e1 < P)
                plan(e1, off + c0, c1);
            static_assert(2 * THREADS >= 256, "the scan covers two partitions per thread");
        }
        __syncthreads();
        // 3. counting sort into the LDS staging arrays; the carried rows that leave are written out by 16 lanes per partition
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
        {
            if (part[j] == ~0u)
                continue;
            const u32 pos = tile_off[part[j]] + rank[j];
            stage_key[pos] = key[j];
            if constexpr (HAS_WORD)
                stage_word[pos] = argw[j];
        }
        for (u32 slot = threadIdx.x; slot < P * CG; slot += GBP_THREADS)
        {
            const u32 p = slot / CG, i = slot % CG;
            if (i < ocnt[p])
            {
                const u32 dst = obase[p] + i;
                out_keys[dst] = carry_key[slot];
                if constexpr (HAS_WORD)
                    out_words[dst] = carry_word[slot];
            }
        }
        if (tbase + GBP_TILE < r1)
            load_tile(tbase + GBP_TILE); // prefetch: lands while this tile is written out
        __syncthreads();
        // 4. stage rows below their partition's new base go to global memory (consecutive lanes -> consecutive rows of a run), the
        //    rest becomes the partition's carry
        const u32 tile_rows = (u32)(r1 - tbase < GBP_TILE ? r1 - tbase : GBP_TILE);
        for (u32 pos = threadIdx.x; pos < tile_rows; pos += GBP_THREADS)
        {
            const KT k = stage_key[pos];
            const u32 p = part_fn(k);
            const u32 dst = delta[p] + pos;
            const u32 nb = base[p];
            if (dst < nb)
            {
                out_keys[dst] = k;
                if constexpr (HAS_WORD)
                    out_words[dst] = stage_word[pos];
            }
            else
            {
                const u32 cs = p * CG + (dst - nb);
                carry_key[cs] = k;
                if constexpr (HAS_WORD)
                    carry_word[cs] = stage_word[pos];
            }
        }
        // no barrier: the next tile's step 1 touches only tile_cnt[] (cleared in step 2); its step 2 -- the first writer of the
        // per-partition plan -- and its step 3 -- the first reader of the carries written above -- sit behind the barrier that ends
        // step 1, which every wave reaches only after it has finished step 4 of this tile
    }
    __syncthreads();
    // the last partial pieces
    for (u32 slot = threadIdx.x; slot < P * CG; slot += GBP_THREADS)
    {
        const u32 p = slot / CG, i = slot % CG;
        if (i < ccnt[p])
        {
            const u32 dst = base[p] + i;
            out_keys[dst] = carry_key[slot];
            if constexpr (HAS_WORD)
                out_words[dst] = carry_word[slot];
        }
    }
}


// ---------------------------------------------------------------------------------------------
// k_rp_scatter: the partition pass without carried tails, written so that NO branch surrounds an LDS or memory operation inside a
// step.  The first version of this pass handled every row inside its own `if (row < end)` region; hipcc then waits for each LDS
// round trip before it issues the next (s_waitcnt lgkmcnt(0) at every control-flow join), a wave spends its time parked
// (SQ_WAIT_ANY = 72 % of its cycles, PMC) and with one 1024-thread workgroup per CU there is nobody else to run.  Here rows beyond
// the end of the range go to a dummy partition P that sorts behind every real one and is never written, so each step is a straight
// run of independent operations: RPT ranks, RPT offset reads, 2*RPT staging writes, and a write-out whose LDS reads are all issued
// before the first global store.  Output rows are 32-bit (the callers bound a call below 2^32 rows).
// dynamic LDS: stage_word u64[TILE] (HAS_WORD) | stage_key KT[TILE] | cursor u32[P] | delta u32[P + 1] | tile_cnt u32[P + 1] | tile_off u32[P + 1]
// ---------------------------------------------------------------------------------------------
template <u32 GBP_TILE, typename KT, bool HAS_WORD, typename PartFn, u32 THREADS = RP_THREADS>
__global__ __launch_bounds__(THREADS) void k_rp_scatter(const KT * __restrict__ keys, const u64 * __restrict__ words, u64 n, u64 rows_per_wg, u32 P,
                                                        const u64 * __restrict__ offsets, KT * __restrict__ out_keys, u64 * __restrict__ out_words, PartFn part_fn)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
    u64 * stage_word = (u64 *)gb_lds;
    KT * stage_key = (KT *)(stage_word + (HAS_WORD ? GBP_TILE : 0));
    u32 * cursor = (u32 *)(stage_key + GBP_TILE);
    u32 * delta = cursor + P;
    u32 * tile_cnt = delta + (P + 1);
    u32 * tile_off = tile_cnt + (P + 1);
    __shared__ u32 wave_tot[THREADS / 64];

    for (u32 p = threadIdx.x; p <= P; p += THREADS)
    {
        if (p < P)
            cursor[p] = (u32)offsets[(u64)p * gridDim.x + blockIdx.x];
        tile_cnt[p] = 0;
    }
    __syncthreads();
    const u64 r0 = (u64)blockIdx.x * rows_per_wg;
    const u64 r1 = r0 + rows_per_wg < n ? r0 + rows_per_wg : n;
    constexpr u32 RPT = GBP_TILE / THREADS;
    static_assert(RPT % 2 == 0, "row pairs");
    KT key[RPT];
    u64 argw[HAS_WORD ? RPT : 1];
    typedef u64 v2q __attribute__((ext_vector_type(2)));
    typedef u32 v2d __attribute__((ext_vector_type(2)));
    auto row_of = [&](u64 tb, u32 j) -> u64 { return tb + (u64)(j >> 1) * (2 * THREADS) + 2 * threadIdx.x + (j & 1); };
    // The loads are unconditional and all vector loads: a pair that would start beyond the range re-reads the range's last whole pair
    // (its rows are invalid -- row_of() >= r1 -- and go to the dummy partition); the lone last row of an odd range is fetched as
    // the second element of the pair that ENDS with it (one row to the left: the hardware takes the misaligned address).  No
    // conditional load anywhere: a load that may or may not have been issued forces s_waitcnt vmcnt(0) on every later wait.
    const bool odd_tail = ((r1 - r0) & 1) != 0;
    // (a workgroup left with a single row re-reads the pair that ENDS with it, not the one that starts with it: never past the column)
    const u64 last_pair = r1 >= r0 + 2 ? ((r1 - r0 - 2) & ~(u64)1) + r0 : (r0 ? r0 - 1 : r0);
    auto load_tile = [&](u64 tb) {
#pragma unroll
        for (u32 j = 0; j < RPT; j += 2)
        {
            const u64 i = row_of(tb, j);
            const bool tail = odd_tail && i + 1 == r1;
            const u64 li = tail ? i - 1 : (i + 1 < r1 ? i : last_pair);
            if constexpr (sizeof(KT) == 4)
            {
                const v2d kk = __builtin_nontemporal_load((const v2d *)(keys + li));
                key[j] = tail ? kk.y : kk.x, key[j + 1] = kk.y;
            }
            else
            {
                const v2q kk = __builtin_nontemporal_load((const v2q *)(keys + li));
                key[j] = tail ? kk.y : kk.x, key[j + 1] = kk.y;
            }
            if constexpr (HAS_WORD)
            {
                const v2q a = __builtin_nontemporal_load((const v2q *)(words + li));
                argw[j] = tail ? a.y : a.x, argw[j + 1] = a.y;
            }
        }
    };
    if (r0 >= r1)
        return; // (the whole workgroup: no barrier below is left waiting)
    u32 part[RPT], rank[RPT];
    // 1. a rank inside the tile's partition bucket (rows past the end: the dummy bucket P)
    auto step_rank = [&](u64 tb) {
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
            part[j] = row_of(tb, j) < r1 ? part_fn(key[j]) : P;
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
            rank[j] = atomicAdd(&tile_cnt[part[j]], 1u);
    };
    // 2. exclusive scan of tile_cnt[P + 1] -> tile_off (P + 1 <= 2 * threads); between two barriers
    auto step_scan = [&]() {
        __syncthreads();
        const u32 e0 = threadIdx.x * 2, e1 = e0 + 1;
        const u32 c0 = e0 <= P ? tile_cnt[e0] : 0, c1 = e1 <= P ? tile_cnt[e1] : 0;
        const u32 v = c0 + c1;
        const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        u32 inc = v;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1)
        {
            const u32 o = __shfl_up(inc, dlt, 64);
            if (lane >= (u32)dlt)
                inc += o;
        }
        if (lane == 63)
            wave_tot[wave] = inc;
        __syncthreads();
        u32 off = inc - v;
        for (u32 w = 0; w < wave; ++w)
            off += wave_tot[w];
        if (e0 <= P)
        {
            tile_off[e0] = off;
            tile_cnt[e0] = 0;
            if (e0 < P)
            {
                const u32 c = cursor[e0];
                delta[e0] = c - off; // stage position -> output row (modulo 2^32: the sum with a position is exact)
                cursor[e0] = c + c0;
            }
            else
                delta[e0] = (u32)n - off; // rows past the end land in the slack behind the n output rows (never read)
        }
        if (e1 <= P)
        {
            tile_off[e1] = off + c0;
            tile_cnt[e1] = 0;
            if (e1 < P)
            {
                const u32 c = cursor[e1];
                delta[e1] = c - (off + c0);
                cursor[e1] = c + c1;
            }
            else
                delta[e1] = (u32)n - (off + c0);
        }
        __syncthreads();
    };
    // The pipeline: a tile's loads are issued BEFORE the previous tile's stores and first used AFTER them, all in one iteration, so
    // hipcc waits for them with s_waitcnt vmcnt(<the stores behind them>) and the stores drain while the next tile is ranked, scanned
    // and staged.  (With the loads' first use at the top of the next iteration the wait count is the minimum over the paths into the
    // loop -- the entry path has no stores behind the loads -- and every tile waited for ALL its predecessor's stores to be
    // acknowledged: 72 % of the wave cycles parked, PMC SQ_WAIT_ANY.)  The prologue issues as many harmless stores into the slack
    // as a write-out does, so that the entry path looks like the back edge to that analysis too.
    load_tile(r0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (u32 j = 0; j < RPT; ++j)
    {
        const u32 dst = (u32)n + j * THREADS + threadIdx.x;
        out_keys[dst] = 0;
        if constexpr (HAS_WORD)
            out_words[dst] = 0;
    }
    __builtin_amdgcn_sched_barrier(0);
    step_rank(r0);
    step_scan();
    for (u64 tbase = r0; tbase < r1; tbase += GBP_TILE)
    {
        // 3. counting sort into the LDS staging arrays
        u32 pos3[RPT];
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
            pos3[j] = tile_off[part[j]];
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
        {
            stage_key[pos3[j] + rank[j]] = key[j];
            if constexpr (HAS_WORD)
                stage_word[pos3[j] + rank[j]] = argw[j];
        }
        __builtin_amdgcn_sched_barrier(0);
        load_tile(tbase + GBP_TILE); // the next tile, unconditional (beyond the range it re-reads the last pair)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // 4. write the partition runs: consecutive lanes -> consecutive addresses inside a run; in halves to bound registers.
        //    EVERY staged row is stored, unconditionally: the rows past the end of the range sort behind the real ones and their
        //    delta points into RP_SCATTER_SLACK rows of slack behind the output, so no branch surrounds a store.
        constexpr u32 HALF = RPT / 2;
        const u32 real_rows = tile_off[P];
#pragma unroll
        for (u32 h = 0; h < 2; ++h)
        {
            KT k4[HALF];
            u64 w4[HAS_WORD ? HALF : 1];
            u32 d4[HALF];
#pragma unroll
            for (u32 j = 0; j < HALF; ++j)
            {
                const u32 pos = (h * HALF + j) * THREADS + threadIdx.x;
                k4[j] = stage_key[pos];
                if constexpr (HAS_WORD)
                    w4[j] = stage_word[pos];
            }
#pragma unroll
            for (u32 j = 0; j < HALF; ++j)
            {
                const u32 pos = (h * HALF + j) * THREADS + threadIdx.x;
                const u32 p = pos < real_rows ? part_fn(k4[j]) : P;
                d4[j] = delta[p];
            }
#pragma unroll
            for (u32 j = 0; j < HALF; ++j)
            {
                const u32 pos = (h * HALF + j) * THREADS + threadIdx.x;
                const u32 dst = d4[j] + pos;
                out_keys[dst] = k4[j];
                if constexpr (HAS_WORD)
                    out_words[dst] = w4[j];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next tile: ranks now (its keys have arrived; the stores above keep draining), then the scan.  The ranks only touch
        // tile_cnt[] (cleared by the last scan); the scan -- the first writer of tile_off[] / delta[] -- sits behind its own barrier,
        // which every wave reaches only after it has finished the write-out above.  One tile too many is ranked at the end (all its
        // rows in the dummy bucket): harmless and branch-free.
        step_rank(tbase + GBP_TILE);
        step_scan();
    }
}

// ---------------------------------------------------------------------------------------------
// k_rp_tilesort: the partition pass WITHOUT a scatter.  Every TILE-row tile of the input is counting-sorted by partition in LDS and
// written back to the SAME row range of the output arrays, so the pass writes whole lines in row order (a streaming copy: the
// scatter of 48-row runs to P different places is what held k_rp_scatter at 3.3-3.9 TB/s where a copy moves 5.4 TB/s,
// tools/mall_bench.hip); next to it goes tile_index[t][0..P] (u16): where each partition's run starts inside tile t, [P] = the
// number of real rows of the tile.  The consumer of partition p gathers one run per tile (tools/gather_runs_bench.hip: ~5 TB/s
// for 48-row runs) -- short runs are cheap to READ (no partial-line write-back, no merge window).  No histogram pass and no offset
// scan precede it; part_total[p] += the rows of partition p (one atomic per partition and workgroup, for the consumer's work split).
// rows_per_wg is a multiple of TILE; out arrays hold ceil(n / TILE) * TILE rows (rows past n sort behind the real rows of the last tile).
// dynamic LDS: stage_word u64[TILE] | stage_key KT[TILE] | tile_cnt u32[P + 1] | tile_off u32[P + 1] | wg_total u32[P + 1]
// ---------------------------------------------------------------------------------------------
// AT / EX: the argument column as stored -- u64 (EX 0), or a 4-byte type widened on the way into LDS: EX 0 zero-extended (UInt32),
// 3 sign-extended (Int32), 4 Float32 -> Float64 bits; the sorted copy always holds 8-byte words.
// AOS: the sorted copy is ONE array of {word, key} records -- 12 bytes for 4-byte keys, 16 for 8-byte keys (out_words = its base,
// out_keys unused) -- instead of a key array and a word array: a partition's run is then one contiguous piece per tile, not two -- the consumer's gather touches
// (576 + 124) / 128 = 5.5 lines per 48-row run instead of 2.5 + 3.9.
template <u32 GBP_TILE, typename KT, typename PartFn, u32 THREADS = RP_THREADS, typename AT = u64, int EX = 0, bool AOS = false>
__global__ __launch_bounds__(THREADS) void k_rp_tilesort(const KT * __restrict__ keys, const AT * __restrict__ words, u64 n, u64 rows_per_wg, u32 P,
                                                         KT * __restrict__ out_keys, u64 * __restrict__ out_words, unsigned short * __restrict__ tile_index,
                                                         unsigned long long * __restrict__ part_total, PartFn part_fn)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
    u64 * stage_word = (u64 *)gb_lds;
    KT * stage_key = (KT *)(stage_word + GBP_TILE);
    u32 * tile_cnt = (u32 *)(stage_key + GBP_TILE);
    u32 * tile_off = tile_cnt + (P + 1);
    u32 * wg_total = tile_off + (P + 1);
    __shared__ u32 wave_tot[THREADS / 64];

    for (u32 p = threadIdx.x; p <= P; p += THREADS)
        tile_cnt[p] = 0, wg_total[p] = 0;
    __syncthreads();
    const u64 r0 = (u64)blockIdx.x * rows_per_wg;
    if (r0 >= n)
        return; // (the whole workgroup: no barrier below is left waiting)
    // Everything below addresses memory as a workgroup-uniform base + a 32-bit byte offset (the host keeps rows_per_wg * 8 under
    // 2^32): one VGPR per address instead of two, and the scalar-base form of the load / store instructions.
    const u32 nrel = (u32)(r0 + rows_per_wg < n ? rows_per_wg : n - r0); // rows of this workgroup
    // (the load bases sit two rows BEFORE the workgroup's first row unless that is row 0: a workgroup left with a single row reads it
    //  as the second element of the pair that ends with it -- one row to the left, never one past the end of the column)
    const u32 shift = r0 ? 2u : 0u;
    const char * kbase = (const char *)(keys + r0 - shift);
    const char * wbase = (const char *)(words + r0 - shift);
    char * okbase = (char *)(out_keys + r0);
    char * owbase = AOS ? (char *)out_words + r0 * (8 + sizeof(KT)) : (char *)(out_words + r0);
    char * ixbase = (char *)(tile_index + (r0 / GBP_TILE) * (u64)(P + 1));
    constexpr u32 RPT = GBP_TILE / THREADS;
    static_assert(RPT % 4 == 0, "whole 16-byte pieces per thread");
    typedef u64 v2q __attribute__((ext_vector_type(2)));
    typedef u32 v2d __attribute__((ext_vector_type(2)));
    typedef u32 v4d __attribute__((ext_vector_type(4)));
    typedef typename std::conditional<sizeof(KT) == 4, v2d, v2q>::type kpair;
    auto rel_of = [&](u32 trel, u32 j) -> u32 { return trel + (j >> 1) * (2 * THREADS) + 2 * threadIdx.x + (j & 1); };
    // Unconditional vector loads into raw pair registers, as in k_rp_scatter: a pair beyond the range re-reads the last whole pair; the
    // lone last row of an odd range comes as the second element of the pair that ends with it.  The tail select happens where the
    // row is used (a select scheduled right behind the loads would make the wave wait for every store issued before them).
    const bool odd_tail = (nrel & 1) != 0;
    const u32 last_pair = (nrel >= 2 ? (nrel - 2) & ~1u : shift ? ~0u : 0u) + shift; // in shifted rows (a lone row: the pair that ends with it)
    kpair kraw[RPT / 2];
    typedef typename std::conditional<sizeof(AT) == 4, v2d, v2q>::type apair;
    apair wraw[RPT / 2];
    auto load_tile = [&](u32 trel) {
#pragma unroll
        for (u32 j = 0; j < RPT; j += 2)
        {
            const u32 i = rel_of(trel, j);
            const bool tail = odd_tail && i + 1 == nrel;
            const u32 li = tail ? i + shift - 1 : (i + 1 < nrel ? i + shift : last_pair);
            kraw[j / 2] = __builtin_nontemporal_load((const kpair *)(kbase + li * (u32)sizeof(KT)));
            wraw[j / 2] = __builtin_nontemporal_load((const apair *)(wbase + li * (u32)sizeof(AT)));
        }
    };
    auto key_at = [&](u32 trel, u32 j) -> KT {
        const bool tail = odd_tail && rel_of(trel, j & ~1u) + 1 == nrel;
        return ((j & 1) || tail) ? (KT)kraw[j / 2].y : (KT)kraw[j / 2].x;
    };
    auto word_at = [&](u32 trel, u32 j) -> u64 {
        const bool tail = odd_tail && rel_of(trel, j & ~1u) + 1 == nrel;
        const auto raw = ((j & 1) || tail) ? wraw[j / 2].y : wraw[j / 2].x;
        if constexpr (EX == 3)
            return (u64)(i64)(i32)(u32)raw;
        else if constexpr (EX == 4)
            return (u64)__double_as_longlong((double)__uint_as_float((u32)raw));
        else
            return (u64)raw;
    };
    u32 part[RPT], rank[RPT];
    auto step_rank = [&](u32 trel) {
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
            part[j] = rel_of(trel, j) < nrel ? part_fn(key_at(trel, j)) : P;
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
            rank[j] = atomicAdd(&tile_cnt[part[j]], 1u);
    };
    // exclusive scan of tile_cnt[P + 1] -> tile_off (P + 1 <= 2 * threads); between two barriers
    auto step_scan = [&]() {
        __syncthreads();
        const u32 e0 = threadIdx.x * 2, e1 = e0 + 1;
        const u32 c0 = e0 <= P ? tile_cnt[e0] : 0, c1 = e1 <= P ? tile_cnt[e1] : 0;
        const u32 v = c0 + c1;
        const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        u32 inc = v;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1)
        {
            const u32 o = __shfl_up(inc, dlt, 64);
            if (lane >= (u32)dlt)
                inc += o;
        }
        if (lane == 63)
            wave_tot[wave] = inc;
        __syncthreads();
        u32 off = inc - v;
        for (u32 w = 0; w < wave; ++w)
            off += wave_tot[w];
        if (e0 <= P)
        {
            tile_off[e0] = off;
            tile_cnt[e0] = 0;
            wg_total[e0] += c0;
        }
        if (e1 <= P)
        {
            tile_off[e1] = off + c0;
            tile_cnt[e1] = 0;
            wg_total[e1] += c1;
        }
        __syncthreads();
    };
    constexpr u32 KPP = 16 / sizeof(KT);                 // keys per 16-byte piece
    constexpr u32 KPIECES = GBP_TILE / KPP / THREADS;    // key pieces per thread and tile
    constexpr u32 WPIECES = GBP_TILE / 2 / THREADS;      // word pieces per thread and tile
    static_assert(GBP_TILE % (KPP * THREADS) == 0 && GBP_TILE % (2 * THREADS) == 0, "whole pieces");
    const u32 e_idx = threadIdx.x < P ? threadIdx.x : P; // this thread's entry of a tile's index (the threads beyond P repeat entry P)
    // The same pipeline as k_rp_scatter: a tile's loads are issued before the previous tile's stores and first used after them.  The
    // prologue issues as many stores as a write-out does (zeros into this workgroup's first tile, by the very threads that
    // overwrite them in the first write-out) so that the loop's entry path looks like its back edge to the wait-count analysis.
    load_tile(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (u32 q = 0; q < KPIECES; ++q)
        __builtin_nontemporal_store(v4d{0, 0, 0, 0}, (v4d *)((AOS ? owbase + WPIECES * THREADS * 16u : okbase) + (q * THREADS + threadIdx.x) * 16u));
#pragma unroll
    for (u32 q = 0; q < WPIECES; ++q)
        __builtin_nontemporal_store(v2q{0, 0}, (v2q *)(owbase + (q * THREADS + threadIdx.x) * 16u));
    *(unsigned short *)(ixbase + e_idx * 2u) = 0;
    __builtin_amdgcn_sched_barrier(0);
    step_rank(0);
    step_scan();
    u32 tile_no = 0;
    for (u32 trel = 0; trel < nrel; trel += GBP_TILE, ++tile_no)
    {
        // counting sort into the LDS staging arrays
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
        {
            const u32 pos = tile_off[part[j]] + rank[j];
            if constexpr (AOS && sizeof(KT) == 4)
            {
                u32 * rec = (u32 *)gb_lds + 3 * pos;
                const u64 w = word_at(trel, j);
                rec[0] = (u32)w; // word first: the consumer's 12-byte load then puts the 8-byte word into an even-aligned register pair
                rec[1] = (u32)(w >> 32);
                rec[2] = (u32)key_at(trel, j);
            }
            else if constexpr (AOS)
            {
                u64 * rec = (u64 *)gb_lds + 2 * pos; // 16-byte records {word, key}
                rec[0] = word_at(trel, j);
                rec[1] = (u64)key_at(trel, j);
            }
            else
            {
                stage_key[pos] = key_at(trel, j);
                stage_word[pos] = word_at(trel, j);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_tile(trel + GBP_TILE); // the next tile, unconditional (beyond the range it re-reads the last pair)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // the sorted tile goes out in row order: 16-byte pieces, consecutive lanes -> consecutive pieces
        {
            v4d kq[KPIECES];
            v2q wq[WPIECES];
#pragma unroll
            for (u32 q = 0; q < KPIECES; ++q)
                kq[q] = *(const v4d *)((const char *)stage_key + (q * THREADS + threadIdx.x) * 16u);
#pragma unroll
            for (u32 q = 0; q < WPIECES; ++q)
                wq[q] = *(const v2q *)((const char *)stage_word + (q * THREADS + threadIdx.x) * 16u);
            const u32 eo = tile_off[e_idx];
#pragma unroll
            // (AOS: the LDS image -- word region then key region -- is one record array and goes out as it lies, to one base)
            for (u32 q = 0; q < KPIECES; ++q)
                __builtin_nontemporal_store(kq[q], (v4d *)((AOS ? owbase + trel * (8u + (u32)sizeof(KT)) + WPIECES * THREADS * 16u : okbase + trel * (u32)sizeof(KT)) + (q * THREADS + threadIdx.x) * 16u));
#pragma unroll
            for (u32 q = 0; q < WPIECES; ++q)
                __builtin_nontemporal_store(wq[q], (v2q *)(owbase + trel * (AOS ? 8u + (u32)sizeof(KT) : 8u) + (q * THREADS + threadIdx.x) * 16u));
            *(unsigned short *)(ixbase + (tile_no * (P + 1) + e_idx) * 2u) = (unsigned short)eo; // no branch around a store
        }
        __builtin_amdgcn_sched_barrier(0);
        step_rank(trel + GBP_TILE);
        step_scan();
    }
    // (the surplus tile ranked at the end put all its rows into the dummy bucket P)
    for (u32 p = threadIdx.x; p < P; p += THREADS)
        if (wg_total[p])
            atomicAdd(&part_total[p], (unsigned long long)wg_total[p]);
}

// ---------------------------------------------------------------------------------------------
// k_rp_tilesort_keys: the tile sort for 8-byte keys alone, with the bucket of a row taken RELATIVE to the tile's first row --
// the second level of a two-level partition: the rows arrive partitioned once (contiguous first-level partitions), a tile lies inside
// one first-level partition or straddles two, and bucket_fn(key, first_key_of_the_tile) numbers the second-level buckets of those
// (at most) two partitions 0 .. P-1; anything else goes to the dummy bucket (the caller guarantees there is none).
// Output as k_rp_tilesort: the sorted tile in its own row range + tile_index[t][0..P] (u16).  16384-row tiles: 128 KiB of LDS.
// dynamic LDS: stage_key u64[TILE] | tile_cnt u32[P + 1] | tile_off u32[P + 1]
// ---------------------------------------------------------------------------------------------
// HAS_WORD: every key carries an 8-byte word (the join build: the row id); LDS then holds TILE x 16 bytes (8192-row tiles).
template <u32 GBP_TILE, typename BucketFn, u32 THREADS = RP_THREADS, bool HAS_WORD = false>
__global__ __launch_bounds__(THREADS) void k_rp_tilesort_keys(const u64 * __restrict__ keys, u64 n, u64 rows_per_wg, u32 P, u64 * __restrict__ out_keys,
                                                              unsigned short * __restrict__ tile_index, BucketFn bucket_fn, u32 * __restrict__ stray_flag,
                                                              const u64 * __restrict__ words = nullptr, u64 * __restrict__ out_words = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
    u64 * stage_key = (u64 *)gb_lds;
    u64 * stage_word = stage_key + GBP_TILE; // (HAS_WORD)
    u32 * tile_cnt = (u32 *)(stage_key + (HAS_WORD ? 2 : 1) * GBP_TILE);
    u32 * tile_off = tile_cnt + (P + 1);
    __shared__ u32 wave_tot[THREADS / 64];
    for (u32 p = threadIdx.x; p <= P; p += THREADS)
        tile_cnt[p] = 0;
    __syncthreads();
    const u64 r0 = (u64)blockIdx.x * rows_per_wg;
    if (r0 >= n)
        return; // (the whole workgroup: no barrier below is left waiting)
    const u32 nrel = (u32)(r0 + rows_per_wg < n ? rows_per_wg : n - r0);
    const u32 shift = r0 ? 2u : 0u; // see k_rp_tilesort: a lone last row is read as the second element of the pair that ends with it
    const char * kbase = (const char *)(keys + r0 - shift);
    char * okbase = (char *)(out_keys + r0);
    const char * wbase = (const char *)(words + r0 - shift);
    char * owbase = (char *)(out_words + r0);
    char * ixbase = (char *)(tile_index + (r0 / GBP_TILE) * (u64)(P + 1));
    constexpr u32 RPT = GBP_TILE / THREADS;
    static_assert(RPT % 2 == 0, "row pairs");
    typedef u64 v2q __attribute__((ext_vector_type(2)));
    auto rel_of = [&](u32 trel, u32 j) -> u32 { return trel + (j >> 1) * (2 * THREADS) + 2 * threadIdx.x + (j & 1); };
    const bool odd_tail = (nrel & 1) != 0;
    const u32 last_pair = (nrel >= 2 ? (nrel - 2) & ~1u : shift ? ~0u : 0u) + shift;
    v2q kraw[RPT / 2];
    v2q wraw[HAS_WORD ? RPT / 2 : 1];
    u64 first_key = 0;
    auto load_tile = [&](u32 trel) {
#pragma unroll
        for (u32 j = 0; j < RPT; j += 2)
        {
            const u32 i = rel_of(trel, j);
            const bool tail = odd_tail && i + 1 == nrel;
            const u32 li = tail ? i + shift - 1 : (i + 1 < nrel ? i + shift : last_pair);
            kraw[j / 2] = __builtin_nontemporal_load((const v2q *)(kbase + li * 8u));
            if constexpr (HAS_WORD)
                wraw[j / 2] = __builtin_nontemporal_load((const v2q *)(wbase + li * 8u));
        }
        first_key = *(const u64 *)(kbase + ((trel < nrel ? trel : nrel - 1) + shift) * 8u); // the tile's first row (every lane the same address)
    };
    auto key_at = [&](u32 trel, u32 j) -> u64 {
        const bool tail = odd_tail && rel_of(trel, j & ~1u) + 1 == nrel;
        return ((j & 1) || tail) ? kraw[j / 2].y : kraw[j / 2].x;
    };
    auto word_at = [&](u32 trel, u32 j) -> u64 {
        const bool tail = odd_tail && rel_of(trel, j & ~1u) + 1 == nrel;
        return ((j & 1) || tail) ? wraw[j / 2].y : wraw[j / 2].x;
    };
    u32 part[RPT], rank[RPT];
    bool stray = false; // a real row whose bucket is out of range (a tile spanning more than two first-level partitions): the caller must not use the result
    auto step_rank = [&](u32 trel) {
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
        {
            const u32 b = bucket_fn(key_at(trel, j), first_key);
            const bool real = rel_of(trel, j) < nrel;
            part[j] = (real && b < P) ? b : P;
            stray |= real && b >= P;
        }
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
            rank[j] = atomicAdd(&tile_cnt[part[j]], 1u);
    };
    auto step_scan = [&]() {
        __syncthreads();
        const u32 e0 = threadIdx.x * 2, e1 = e0 + 1;
        const u32 c0 = e0 <= P ? tile_cnt[e0] : 0, c1 = e1 <= P ? tile_cnt[e1] : 0;
        const u32 v = c0 + c1;
        const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        u32 inc = v;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1)
        {
            const u32 o = __shfl_up(inc, dlt, 64);
            if (lane >= (u32)dlt)
                inc += o;
        }
        if (lane == 63)
            wave_tot[wave] = inc;
        __syncthreads();
        u32 off = inc - v;
        for (u32 w = 0; w < wave; ++w)
            off += wave_tot[w];
        if (e0 <= P)
        {
            tile_off[e0] = off;
            tile_cnt[e0] = 0;
        }
        if (e1 <= P)
        {
            tile_off[e1] = off + c0;
            tile_cnt[e1] = 0;
        }
        __syncthreads();
    };
    constexpr u32 KPIECES = GBP_TILE / 2 / THREADS;
    const u32 e_idx = threadIdx.x < P ? threadIdx.x : P;
    load_tile(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (u32 q = 0; q < KPIECES; ++q)
    {
        __builtin_nontemporal_store(v2q{0, 0}, (v2q *)(okbase + (q * THREADS + threadIdx.x) * 16u));
        if constexpr (HAS_WORD)
            __builtin_nontemporal_store(v2q{0, 0}, (v2q *)(owbase + (q * THREADS + threadIdx.x) * 16u));
    }
    *(unsigned short *)(ixbase + e_idx * 2u) = 0;
    __builtin_amdgcn_sched_barrier(0);
    step_rank(0);
    step_scan();
    u32 tile_no = 0;
    for (u32 trel = 0; trel < nrel; trel += GBP_TILE, ++tile_no)
    {
#pragma unroll
        for (u32 j = 0; j < RPT; ++j)
        {
            const u32 pos = tile_off[part[j]] + rank[j];
            stage_key[pos] = key_at(trel, j);
            if constexpr (HAS_WORD)
                stage_word[pos] = word_at(trel, j);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_tile(trel + GBP_TILE);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        {
            v2q kq[KPIECES];
            v2q wq[HAS_WORD ? KPIECES : 1];
#pragma unroll
            for (u32 q = 0; q < KPIECES; ++q)
            {
                kq[q] = *(const v2q *)((const char *)stage_key + (q * THREADS + threadIdx.x) * 16u);
                if constexpr (HAS_WORD)
                    wq[q] = *(const v2q *)((const char *)stage_word + (q * THREADS + threadIdx.x) * 16u);
            }
            const u32 eo = tile_off[e_idx];
#pragma unroll
            for (u32 q = 0; q < KPIECES; ++q)
            {
                __builtin_nontemporal_store(kq[q], (v2q *)(okbase + trel * 8u + (q * THREADS + threadIdx.x) * 16u));
                if constexpr (HAS_WORD)
                    __builtin_nontemporal_store(wq[q], (v2q *)(owbase + trel * 8u + (q * THREADS + threadIdx.x) * 16u));
            }
            *(unsigned short *)(ixbase + (tile_no * (P + 1) + e_idx) * 2u) = (unsigned short)eo;
        }
        __builtin_amdgcn_sched_barrier(0);
        step_rank(trel + GBP_TILE);
        step_scan();
    }
    if (stray)
        *stray_flag = 1;
}

static inline size_t rp_tilesort_lds_bytes(u32 tile, u32 P, size_t key_bytes)
{
    return (size_t)tile * (key_bytes + 8) + (size_t)(P + 1) * 12 + 64;
}

static inline size_t rp_scatter_lds_bytes(u32 tile, u32 P, size_t key_bytes, bool has_word)
{
    return (size_t)tile * (key_bytes + (has_word ? 8 : 0)) + (size_t)P * 4 + (size_t)(P + 1) * 12 + 64;
}

static inline size_t rp_scatter_carry_lds_bytes(u32 tile, u32 P, u32 cg, size_t key_bytes, bool has_word)
{
    return (size_t)tile * (key_bytes + (has_word ? 8 : 0)) + (size_t)P * cg * (key_bytes + (has_word ? 8 : 0)) + (size_t)P * 28 + 64;
}
