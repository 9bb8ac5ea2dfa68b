// radix_partition.h — the two kernels of an exact (histogram-first) radix partition with carried tails, shared by the GROUP BY
// (agg_kernels.hip: rows -> partitions that fit an LDS table) and the join probe (join_kernels.hip: probe keys -> table regions that
// fit an XCD's L2).  The partition function is a functor so both callers run the same code.
#pragma once
#include "chgpu_internal.h"

static constexpr u32 RP_THREADS = 1024;
static constexpr u32 RP_MAX_P = 1024;

// Same histogram with 16-byte nontemporal key loads (4- and 8-byte keys whose first row is 16-byte aligned): four loads
// per lane are issued before the first LDS atomic.
template <typename KT, typename PartFn>
__global__ __launch_bounds__(RP_THREADS) void k_rp_hist_wide(const KT * __restrict__ keys, u64 n, u64 rows_per_wg, u32 P, u32 * __restrict__ counts, PartFn part_fn)
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
        counts[(u64)p * gridDim.x + blockIdx.x] = cnt[p];
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
// dynamic LDS: stage_word u64[TILE] | carry_word u64[P*16] (both only with HAS_WORD) | base u64[P] | delta u64[P] | obase u64[P] |
//              stage_key KT[TILE] | carry_key KT[P*16] | ccnt u32[P] | tile_cnt u32[P] | tile_off u32[P] | ocnt u32[P]
// PartFn: u32 operator()(KT key) const -> partition in [0, P).  HAS_WORD = false: keys only (words / out_words unused).
template <u32 GBP_TILE, typename KT, bool HAS_WORD, typename PartFn>
__global__ __launch_bounds__(RP_THREADS) void k_rp_scatter_carry(const KT * __restrict__ keys, const u64 * __restrict__ words, u64 n, u64 rows_per_wg, u32 P,
                                                                 const u64 * __restrict__ offsets, KT * __restrict__ out_keys, u64 * __restrict__ out_words, PartFn part_fn)
{
    constexpr u32 GBP_THREADS = RP_THREADS;
    constexpr u32 CG = 16; // carry granularity in rows
    extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
    u64 * stage_word = (u64 *)gb_lds;
    u64 * carry_word = stage_word + (HAS_WORD ? GBP_TILE : 0);
    u64 * base = carry_word + (HAS_WORD ? (size_t)P * CG : 0);
    u64 * delta = base + P;
    u64 * obase = delta + P;
    KT * stage_key = (KT *)(obase + P);
    KT * carry_key = stage_key + GBP_TILE;
    u32 * ccnt = (u32 *)(carry_key + (size_t)P * CG);
    u32 * tile_cnt = ccnt + P;
    u32 * tile_off = tile_cnt + P;
    u32 * ocnt = tile_off + P;
    __shared__ u32 wave_tot[GBP_THREADS / 64];

    for (u32 p = threadIdx.x; p < P; p += GBP_THREADS)
    {
        base[p] = offsets[(u64)p * gridDim.x + blockIdx.x];
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
                const u64 b = base[p];
                const u32 c = ccnt[p];
                const u64 end = b + c + nnew;
                const u64 fe = end & ~(u64)(CG - 1);
                const bool flush = fe > b;
                tile_off[p] = toff;
                delta[p] = b + c - toff;          // stage position -> output row
                obase[p] = b;
                ocnt[p] = flush ? c : 0;          // old carry rows that leave now (all of them: they sit below fe)
                base[p] = flush ? fe : b;         // rows at or above it stay in LDS as carry slot (row - base)
                ccnt[p] = (u32)(end - (flush ? fe : b));
                tile_cnt[p] = 0;
            };
            if (e0 < P)
                plan(e0, off, c0);
            if (e1 < P)
                plan(e1, off + c0, c1);
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
                const u64 dst = obase[p] + i;
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
            const u64 dst = delta[p] + pos;
            const u64 nb = base[p];
            if (dst < nb)
            {
                out_keys[dst] = k;
                if constexpr (HAS_WORD)
                    out_words[dst] = stage_word[pos];
            }
            else
            {
                const u32 cs = p * CG + (u32)(dst - nb);
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
            const u64 dst = base[p] + i;
            out_keys[dst] = carry_key[slot];
            if constexpr (HAS_WORD)
                out_words[dst] = carry_word[slot];
        }
    }
}

