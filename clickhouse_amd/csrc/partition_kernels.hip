// partition_kernels.hip — externally visible hashing (bit-exact CRC32-C) and stable hash partitioning.
//
// Reference loops replaced (file:line in the reference checkout):
//   k_weak_hash32      ColumnVector<T>::getWeakHash32                 src/Columns/ColumnVector.cpp:78-95 (hashCRC32, Hash.h:276-288)
//   k_selector         ConcurrentHashJoin calculateHashes + hashToSelector
//                                                                      src/Interpreters/ConcurrentHashJoin.cpp:426-452,
//                      TwoLevelHashTable::getBucketFromHash            src/Common/HashTable/TwoLevelHashTable.h:53
//   k_part_*           IColumn::scatter / scatterBlocksByCopying       src/Columns/IColumn.cpp:245-269, ConcurrentHashJoin.cpp:494-536
//
// CDNA has no crc32c instruction: CRC32-C of a 64-bit key is computed from eight 256-entry byte tables staged in LDS
// (crc is GF(2)-affine in the message).  The partition is a stable three-pass split (per-tile histogram -> scan ->
// ordered scatter through LDS, see k_part_scatter_lds), so each shard's rows keep their input order exactly like the
// reference's sequential insertFrom loop.
#include "chgpu_internal.h"

#include <algorithm>

static constexpr u32 PT = 256;
static constexpr u32 MAX_SHARDS = 256;
static constexpr u32 MAX_PART_COLS = 8;


__device__ __forceinline__ u64 pload_key(const void * keys, int type, u64 i)
{
    switch (type)
    {
        case CHGPU_U32: case CHGPU_I32: return ((const u32 *)keys)[i];
        case CHGPU_U16: case CHGPU_I16: return ((const u16 *)keys)[i];
        case CHGPU_U8: case CHGPU_I8: return ((const u8 *)keys)[i];
        default: return ((const u64 *)keys)[i];
    }
}

__device__ __forceinline__ void stage_lut(u32 * lds_lut, const u32 * __restrict__ lut)
{
    for (u32 k = threadIdx.x; k < 2049; k += blockDim.x)
        lds_lut[k] = lut[k];
    __syncthreads();
}

__global__ __launch_bounds__(PT) void k_weak_hash32(const void * __restrict__ data, int type, u64 n, const u32 * __restrict__ lut, u32 * __restrict__ hash)
{
    __shared__ u32 l[2049];
    stage_lut(l, lut);
    for (u64 i = (u64)blockIdx.x * PT + threadIdx.x; i < n; i += (u64)gridDim.x * PT)
    {
        const u32 seed = hash[i];
        // crc(seed, x) = crc(seed, 0) ^ tab(x); crc(-1, 0) is precomputed, other seeds are folded bit by bit
        const u32 base = seed == 0xFFFFFFFFu ? l[2048] : dev_crc32c_zero8(seed);
        hash[i] = base ^ dev_crc32c_tab(l, pload_key(data, type, i));
    }
}

__global__ __launch_bounds__(PT) void k_selector(const void * __restrict__ keys, int type, u64 n, const u32 * __restrict__ lut, u32 shard_mask, u32 * __restrict__ sel)
{
    __shared__ u32 l[2049];
    stage_lut(l, lut);
    for (u64 i = (u64)blockIdx.x * PT + threadIdx.x; i < n; i += (u64)gridDim.x * PT)
    {
        const u32 crc = l[2048] ^ dev_crc32c_tab(l, pload_key(keys, type, i));
        sel[i] = ((crc >> 24) & 0xFF) & shard_mask; // getBucketFromHash(h) & (num_shards - 1)
    }
}

struct PartCols
{
    u32 n_cols;
    u32 elem_size[MAX_PART_COLS];
    const void * src[MAX_PART_COLS];
    void * dst[MAX_PART_COLS];
};

// ---------------------------------------------------------------------------------------------
// LDS-staged stable partition.  A first version (one wave per tile, every row stored straight to its destination: 64 scattered
// 4-8 byte stores per wave instruction) was bound by the rate of partial-line writes (3.06 ms per 256-way pass over 1e8 16-byte
// rows against 1.6 ms); here a
// workgroup sorts a tile of PL_TILE rows by shard in LDS and writes each shard's run with consecutive lanes on consecutive
// addresses.  Stability: rows are taken wave-striped (step j of wave w covers 64 consecutive rows), a row's rank inside
// its wave = the wave's running per-shard counter + the number of lower lanes of this step with the same shard (per-bit
// ballots), and waves are ordered by an exclusive prefix over their per-shard totals.
// The shard of a row comes from a UInt32 selector column or, for the radix sort, straight from a key byte (SelSrc).
// ---------------------------------------------------------------------------------------------
struct SelSrc
{
    const void * p;
    u32 mode;  // 0: UInt32 selector column; 1/2/4/8: key column of that many bytes, shard = (key >> shift) & 0xFF
    u32 shift;
};
__device__ __forceinline__ u32 sel_at(const SelSrc & s, u64 i, u32 num_shards)
{
    u32 v;
    switch (s.mode)
    {
        case 0: v = ((const u32 *)s.p)[i]; break;
        case 1: v = ((const u8 *)s.p)[i]; break;
        case 2: v = (((const u16 *)s.p)[i] >> s.shift) & 0xFFu; break;
        case 4: v = (((const u32 *)s.p)[i] >> s.shift) & 0xFFu; break;
        default: v = (u32)(((const u64 *)s.p)[i] >> s.shift) & 0xFFu; break;
    }
    return v < num_shards ? v : 0; // a selector beyond the shard count is a caller bug: folded into shard 0
}

#ifndef PL_RPT_V
#define PL_RPT_V 16
#endif
static constexpr u32 PL_RPT = PL_RPT_V;       // rows per thread
static constexpr u32 PL_TILE = PT * PL_RPT;   // 4096 rows per workgroup tile (A/B on 1e8-row sorts: 16 rows/thread 13.1 ms, 32: 14.7 ms, 8: 14.1 ms)
static constexpr u32 PL_WAVE_ROWS = PL_TILE / (PT / 64);

// (a wave-per-tile variant without workgroup barriers was measured 3-5 % slower on the 256-shard radix passes, same box A/B)
__global__ __launch_bounds__(PT) void k_part_hist_lds(SelSrc sel, u64 n, u32 num_shards, u64 n_tiles, u32 * __restrict__ counts)
{
    __shared__ u32 hist[MAX_SHARDS];
    for (u64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
    {
        for (u32 s = threadIdx.x; s < (num_shards < 8 ? 8u : num_shards); s += PT)
            hist[s] = 0;
        __syncthreads();
        const u64 base = tile * PL_TILE;
        u32 v[PL_RPT];
        // the histogram does not care which thread sees which row: full tiles of 4- and 8-byte sources are read 16 bytes per lane
        // (4-byte loads reach about half the rate of 16-byte ones on this part)
        const bool full = base + PL_TILE <= n && ((uintptr_t)sel.p & 15) == 0;
        if (full && (sel.mode == 0 || sel.mode == 4))
        {
            const u32 sh = sel.mode == 0 ? 0 : sel.shift, mk = sel.mode == 0 ? ~0u : 0xFFu;
#pragma unroll
            for (u32 j = 0; j < PL_RPT / 4; ++j)
            {
                const uint4 q = *(const uint4 *)((const u32 *)sel.p + base + ((u64)j * PT + threadIdx.x) * 4);
                const u32 a[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (u32 r = 0; r < 4; ++r)
                {
                    const u32 x = (a[r] >> sh) & mk;
                    v[j * 4 + r] = x < num_shards ? x : 0;
                }
            }
        }
        else if (full && sel.mode == 8)
        {
#pragma unroll
            for (u32 j = 0; j < PL_RPT / 2; ++j)
            {
                const ulonglong2 q = *(const ulonglong2 *)((const u64 *)sel.p + base + ((u64)j * PT + threadIdx.x) * 2);
                const u32 x0 = (u32)(q.x >> sel.shift) & 0xFFu, x1 = (u32)(q.y >> sel.shift) & 0xFFu;
                v[j * 2] = x0 < num_shards ? x0 : 0;
                v[j * 2 + 1] = x1 < num_shards ? x1 : 0;
            }
        }
        else
        {
#pragma unroll
            for (u32 j = 0; j < PL_RPT; ++j)
            {
                const u64 i = base + (u64)j * PT + threadIdx.x;
                v[j] = i < n ? sel_at(sel, i, num_shards) : ~0u;
            }
        }
        if (num_shards <= 8)
        {
            // few shards (the 2/4/8-GPU split): 256 lanes hammering 8 LDS counters serialise; count in registers, reduce per wave
            u32 c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (u32 j = 0; j < PL_RPT; ++j)
#pragma unroll
                for (u32 q = 0; q < 8; ++q)
                    c[q] += v[j] == q ? 1u : 0u;
#pragma unroll
            for (u32 q = 0; q < 8; ++q)
            {
                const u32 t = wave_reduce_add_u32(c[q]);
                if ((threadIdx.x & 63) == 0 && t)
                    atomicAdd(&hist[q], t);
            }
        }
        else
        {
#pragma unroll
            for (u32 j = 0; j < PL_RPT; ++j)
                if (v[j] != ~0u)
                    atomicAdd(&hist[v[j]], 1u);
        }
        __syncthreads();
        for (u32 s = threadIdx.x; s < num_shards; s += PT)
            counts[(u64)s * n_tiles + tile] = hist[s];
        __syncthreads();
    }
}

// one column of the tile: rows (64 apart per step, from `src`) -> LDS at their destination position -> runs written with
// consecutive lanes on consecutive addresses.  Loads go eight at a time (the whole 32 at once cost 256 VGPRs).
template <typename T>
__device__ __forceinline__ void part_move_column(const T * __restrict__ src, T * __restrict__ dst, const u32 (&pos)[PL_RPT], T * stage, const u8 * dig,
                                                 const u64 * gdelta, u32 tile_rows)
{
#pragma unroll
    for (u32 j0 = 0; j0 < PL_RPT; j0 += 8)
    {
        T v[8];
#pragma unroll
        for (u32 q = 0; q < 8; ++q)
            if (pos[j0 + q] != ~0u)
                v[q] = src[(j0 + q) * 64];
#pragma unroll
        for (u32 q = 0; q < 8; ++q)
            if (pos[j0 + q] != ~0u)
                stage[pos[j0 + q]] = v[q];
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads(); // (first column: also publishes dig[])
    for (u32 p = threadIdx.x; p < tile_rows; p += PT)
        dst[gdelta[dig[p]] + p] = stage[p];
    __syncthreads();
}

__global__ __launch_bounds__(PT) void k_part_scatter_lds(SelSrc sel, u64 n, u32 num_shards, u32 shard_bits, u64 n_tiles,
                                                         const u64 * __restrict__ offsets, PartCols cols)
{
    __shared__ __attribute__((aligned(16))) u64 stage[PL_TILE]; // one column of the tile in destination order (32 KiB)
    __shared__ u8 dig[PL_TILE];                                 // shard of the row at each destination position
    __shared__ u32 wcnt[PT / 64][MAX_SHARDS];                   // per wave: running count, then exclusive prefix over waves
    __shared__ u32 doff[MAX_SHARDS];                            // first position of each shard inside the sorted tile
    __shared__ u64 gdelta[MAX_SHARDS];                          // global row of (shard, tile) minus doff
    __shared__ u32 wave_sum[PT / 64];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
    {
        const u64 base = tile * PL_TILE;
        const u32 tile_rows = (u32)(n - base < PL_TILE ? n - base : PL_TILE);
        for (u32 s = threadIdx.x; s < (PT / 64) * MAX_SHARDS; s += PT)
            (&wcnt[0][0])[s] = 0;
        __syncthreads(); // also: the previous tile's write-out has finished with stage / dig / gdelta
        // A. ranks inside the wave's strip of PL_WAVE_ROWS rows, 64 rows per step
        u32 sr[PL_RPT]; // shard << 16 | rank in wave's strip   (rank < 2048)
        u32 sv[PL_RPT];
#pragma unroll
        for (u32 j = 0; j < PL_RPT; ++j)
        {
            const u32 r = wave * PL_WAVE_ROWS + j * 64 + lane;
            sv[j] = r < tile_rows ? sel_at(sel, base + r, num_shards) : ~0u;
        }
#pragma unroll
        for (u32 j = 0; j < PL_RPT; ++j)
        {
            const bool in = sv[j] != ~0u;
            const u32 s = in ? sv[j] : 0;
            u64 peers = __ballot(in);
            for (u32 b = 0; b < shard_bits; ++b)
            {
                const u64 bal = __ballot((s >> b) & 1);
                peers &= ((s >> b) & 1) ? bal : ~bal;
            }
            const u32 before = mbcnt(peers);
            u32 rank = 0;
            if (in)
                rank = wcnt[wave][s] + before; // LDS operations of one wave complete in order
            __builtin_amdgcn_wave_barrier();
            if (in && before == 0)
                wcnt[wave][s] += (u32)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
            sr[j] = in ? ((s << 16) | rank) : ~0u;
        }
        __syncthreads();
        // B. per shard: exclusive prefix over the waves, tile total; exclusive scan of the totals over the shards
        {
            const u32 d = threadIdx.x;
            u32 tot = 0;
            if (d < num_shards)
            {
#pragma unroll
                for (u32 w = 0; w < PT / 64; ++w)
                {
                    const u32 c = wcnt[w][d];
                    wcnt[w][d] = tot;
                    tot += c;
                }
            }
            u32 inc = tot;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1)
            {
                const u32 o = __shfl_up(inc, dlt, WAVE);
                if (lane >= (u32)dlt)
                    inc += o;
            }
            if (lane == 63)
                wave_sum[wave] = inc;
            __syncthreads();
            u32 ex = inc - tot;
            for (u32 w = 0; w < wave; ++w)
                ex += wave_sum[w];
            if (d < num_shards)
            {
                doff[d] = ex;
                gdelta[d] = offsets[(u64)d * n_tiles + tile] - ex;
            }
        }
        __syncthreads();
        // C. destination position of every row inside the sorted tile
        u32 pos[PL_RPT];
#pragma unroll
        for (u32 j = 0; j < PL_RPT; ++j)
        {
            pos[j] = ~0u;
            if (sr[j] != ~0u)
            {
                const u32 s = sr[j] >> 16;
                pos[j] = doff[s] + wcnt[wave][s] + (sr[j] & 0xFFFFu);
                dig[pos[j]] = (u8)s;
            }
        }
        // D. column by column: rows -> LDS in destination order -> coalesced runs
        for (u32 c = 0; c < cols.n_cols; ++c)
        {
            const u64 first = base + wave * PL_WAVE_ROWS + lane;
            switch (cols.elem_size[c])
            {
                case 8: part_move_column<u64>((const u64 *)cols.src[c] + first, (u64 *)cols.dst[c], pos, (u64 *)stage, dig, gdelta, tile_rows); break;
                case 4: part_move_column<u32>((const u32 *)cols.src[c] + first, (u32 *)cols.dst[c], pos, (u32 *)stage, dig, gdelta, tile_rows); break;
                case 2: part_move_column<u16>((const u16 *)cols.src[c] + first, (u16 *)cols.dst[c], pos, (u16 *)stage, dig, gdelta, tile_rows); break;
                default: part_move_column<u8>((const u8 *)cols.src[c] + first, (u8 *)cols.dst[c], pos, (u8 *)stage, dig, gdelta, tile_rows); break;
            }
        }
    }
}

__global__ void k_part_shard_starts(const u64 * __restrict__ offsets, u64 n_tiles, u32 num_shards, u64 * __restrict__ starts)
{
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < num_shards)
        starts[s] = offsets[(u64)s * n_tiles];
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
extern "C" int chgpu_weak_hash32(chgpu_ctx * ctx, const chgpu_col * col, chgpu_col * hash)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && col && hash, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(hash->type == CHGPU_U32, CHGPU_ERR_BAD_ARGUMENTS, "WeakHash32 data must be UInt32");
    CHGPU_REQUIRE(hash->rows == col->rows, CHGPU_ERR_SIZES_MISMATCH, "Size of WeakHash32 does not match size of column: column size is %llu, hash size is %llu",
                  (unsigned long long)col->rows, (unsigned long long)hash->rows);
    const u32 * lut = nullptr;
    CHGPU_TRY(chgpu_crc_lut(ctx, &lut));
    if (col->rows)
    {
        hipLaunchKernelGGL(k_weak_hash32, dim3(chgpu_grid_for(ctx, col->rows, PT, 8)), dim3(PT), 0, ctx->stream, (const void *)col->data, col->type, col->rows, lut, (u32 *)hash->data);
        ctx->counters[6] += 1;
    }
    CHGPU_HIP(hipGetLastError());
    return CHGPU_OK;
}

static int check_shards(u32 num_shards)
{
    CHGPU_REQUIRE(num_shards >= 1 && num_shards <= MAX_SHARDS && (num_shards & (num_shards - 1)) == 0, CHGPU_ERR_BAD_ARGUMENTS,
                  "num_shards must be a power of two <= %u (ConcurrentHashJoin.cpp:158)", MAX_SHARDS);
    return CHGPU_OK;
}

extern "C" int chgpu_hash_to_selector(chgpu_ctx * ctx, const chgpu_col * keys, uint32_t num_shards, chgpu_col ** selector)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && keys && selector, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_TRY(check_shards(num_shards));
    CHGPU_REQUIRE(!chgpu_type_is_float(keys->type), CHGPU_ERR_NOT_IMPLEMENTED, "Float64 shard keys: CPU path");
    const u32 * lut = nullptr;
    CHGPU_TRY(chgpu_crc_lut(ctx, &lut));
    chgpu_col * sel = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U32, keys->rows, &sel));
    if (keys->rows)
    {
        hipLaunchKernelGGL(k_selector, dim3(chgpu_grid_for(ctx, keys->rows, PT, 8)), dim3(PT), 0, ctx->stream, (const void *)keys->data, keys->type, keys->rows, lut,
                           num_shards - 1, (u32 *)sel->data);
        ctx->counters[6] += 1;
    }
    *selector = sel;
    return CHGPU_OK;
}

// Stable split of n_cols columns by shard into concatenated outputs; counts[num_shards] on the host (counts == nullptr: no
// read-back, no host synchronisation -- the radix sort's passes).
static int partition_core_src(chgpu_ctx * ctx, SelSrc sel, u64 n, u32 num_shards, u32 n_cols, const chgpu_col * const * cols,
                              chgpu_col ** outs, u64 * counts)
{
    CHGPU_REQUIRE(n_cols >= 1 && n_cols <= MAX_PART_COLS, CHGPU_ERR_NOT_IMPLEMENTED, "at most %u columns per partition call", MAX_PART_COLS);
    for (u32 c = 0; c < n_cols; ++c)
    {
        CHGPU_REQUIRE(cols[c], CHGPU_ERR_BAD_ARGUMENTS, "column %u is NULL", c);
        CHGPU_REQUIRE(cols[c]->rows == n, CHGPU_ERR_SIZES_MISMATCH, "Size of selector (%llu) doesn't match size of column (%llu)",
                      (unsigned long long)n, (unsigned long long)cols[c]->rows); // IColumn.cpp:249-251
        outs[c] = nullptr;
    }
    const u32 tile_rows = PL_TILE;
    const u64 n_tiles = (n + tile_rows - 1) / tile_rows;
    const u64 m = (u64)num_shards * (n_tiles ? n_tiles : 1);
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t cnt_b = al(m * 4), off_b = al(m * 8), st_b = al((size_t)num_shards * 8 + 8), tmp_b = chgpu_scan_tmp_bytes(m);
    void * scratch = nullptr;
    CHGPU_TRY(chgpu_scratch(ctx, cnt_b + off_b + st_b + tmp_b, &scratch));
    u32 * cnt = (u32 *)scratch;
    u64 * offs = (u64 *)((char *)scratch + cnt_b);
    u64 * starts = (u64 *)((char *)scratch + cnt_b + off_b); // [num_shards] + total
    void * tmp = (char *)scratch + cnt_b + off_b + st_b;

    for (u32 c = 0; c < n_cols; ++c)
    {
        int rc = chgpu_col_new(ctx, cols[c]->type, n, &outs[c]);
        if (rc != CHGPU_OK)
        {
            for (u32 k = 0; k < c; ++k)
                chgpu_col_free(outs[k]);
            return rc;
        }
    }
    if (n == 0)
    {
        if (counts)
            memset(counts, 0, sizeof(u64) * num_shards);
        return CHGPU_OK;
    }
    u32 shard_bits = 0;
    while ((1u << shard_bits) < num_shards)
        ++shard_bits;
    PartCols pc;
    pc.n_cols = n_cols;
    for (u32 c = 0; c < n_cols; ++c)
    {
        pc.elem_size[c] = (u32)chgpu_type_size(cols[c]->type);
        pc.src[c] = cols[c]->data;
        pc.dst[c] = outs[c]->data;
    }
    {
        constexpr u32 wg_per_cu = PL_RPT >= 32 ? 2 : PL_RPT >= 16 ? 3 : 5; // what LDS (9 B per tile row + 7 KiB) and the VGPR count allow
        const u32 grid = (u32)std::min<u64>(n_tiles, (u64)ctx->num_cus * wg_per_cu);
        hipLaunchKernelGGL(k_part_hist_lds, dim3((u32)std::min<u64>(n_tiles, (u64)ctx->num_cus * 8)), dim3(PT), 0, ctx->stream, sel, n, num_shards, n_tiles, cnt);
        CHGPU_TRY(chgpu_scan_exclusive_u32_u64(ctx, cnt, offs, m, starts + num_shards, tmp, tmp_b));
        hipLaunchKernelGGL(k_part_scatter_lds, dim3(grid), dim3(PT), 0, ctx->stream, sel, n, num_shards, shard_bits, n_tiles, (const u64 *)offs, pc);
    }
    ctx->counters[6] += 3;
    hipError_t e = hipGetLastError();
    int rc = CHGPU_OK;
    if (counts)
    {
        hipLaunchKernelGGL(k_part_shard_starts, dim3((num_shards + 63) / 64), dim3(64), 0, ctx->stream, (const u64 *)offs, n_tiles, num_shards, starts);
        u64 host_starts[MAX_SHARDS + 1];
        rc = chgpu_read_back(ctx, starts, host_starts, sizeof(u64) * (num_shards + 1));
        if (rc == CHGPU_OK)
            for (u32 s = 0; s < num_shards; ++s)
                counts[s] = (s + 1 < num_shards ? host_starts[s + 1] : host_starts[num_shards]) - host_starts[s];
    }
    if (rc != CHGPU_OK || e != hipSuccess)
    {
        for (u32 c = 0; c < n_cols; ++c)
            chgpu_col_free(outs[c]);
        return rc != CHGPU_OK ? rc : chgpu_set_error(CHGPU_ERR_DEVICE, "partition launch: %s", hipGetErrorString(e));
    }
    return CHGPU_OK;
}

int chgpu_partition_core(chgpu_ctx * ctx, const u32 * sel, u64 n, u32 num_shards, u32 n_cols, const chgpu_col * const * cols, chgpu_col ** outs, u64 * counts)
{
    return partition_core_src(ctx, SelSrc{sel, 0, 0}, n, num_shards, n_cols, cols, outs, counts);
}

// one radix pass: stable 256-way split of cols[] by byte `shift / 8` of the key column (no selector column, no host synchronisation)
int chgpu_partition_by_key_byte(chgpu_ctx * ctx, const chgpu_col * keys, u32 shift, u32 n_cols, const chgpu_col * const * cols, chgpu_col ** outs)
{
    return partition_core_src(ctx, SelSrc{keys->data, (u32)chgpu_type_size(keys->type), shift}, keys->rows, 256, n_cols, cols, outs, nullptr);
}

extern "C" int chgpu_partition_by_hash(chgpu_ctx * ctx, const chgpu_col * keys, uint32_t num_shards, uint32_t n_cols,
                                       const chgpu_col * const * cols, chgpu_col ** outs, uint64_t * counts)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && keys && cols && outs && counts, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_TRY(check_shards(num_shards));
    chgpu_col * sel = nullptr;
    CHGPU_TRY(chgpu_hash_to_selector(ctx, keys, num_shards, &sel));
    int rc = chgpu_partition_core(ctx, (const u32 *)sel->data, keys->rows, num_shards, n_cols, cols, outs, counts);
    chgpu_col_free(sel); // back to the pool; reuse is ordered behind the kernels above on this stream
    return rc;
}

extern "C" int chgpu_scatter(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * selector, uint32_t num_columns, chgpu_col ** outs)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && col && selector && outs, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(num_columns >= 1 && num_columns <= MAX_SHARDS, CHGPU_ERR_NOT_IMPLEMENTED, "scatter into more than %u columns: CPU path", MAX_SHARDS);
    CHGPU_REQUIRE(selector->type == CHGPU_U32, CHGPU_ERR_BAD_ARGUMENTS, "selector must be a UInt32 column");
    CHGPU_REQUIRE(selector->rows == col->rows, CHGPU_ERR_SIZES_MISMATCH, "Size of selector (%llu) doesn't match size of column (%llu)",
                  (unsigned long long)selector->rows, (unsigned long long)col->rows);
    // the split kernel addresses shards by bit pattern: round the shard count up to a power of two (extra shards stay empty)
    u32 shards_p2 = 1;
    while (shards_p2 < num_columns)
        shards_p2 <<= 1;
    chgpu_col * cat = nullptr;
    u64 counts[MAX_SHARDS];
    const chgpu_col * in[1] = {col};
    CHGPU_TRY(chgpu_partition_core(ctx, (const u32 *)selector->data, col->rows, shards_p2, 1, in, &cat, counts));
    // hand the concatenated buffer out as num_columns columns sharing one allocation
    int * refs = new int(0);
    u64 pos = 0;
    const size_t es = chgpu_type_size(col->type);
    for (u32 s = 0; s < num_columns; ++s)
    {
        chgpu_col * v = new chgpu_col();
        chgpu_ctx_retain(ctx);
        v->ctx = ctx;
        v->type = col->type;
        v->rows = counts[s];
        v->data = (char *)cat->data + pos * es;
        v->base = cat->base;
        v->owns = true;
        v->alloc_bytes = cat->alloc_bytes; // the shared allocation goes back to the pool with the last view
        v->shared_refs = refs;
        ++*refs;
        outs[s] = v;
        pos += counts[s];
    }
    // rows whose selector was >= num_columns (a caller bug) were folded into shard 0 by the kernels
    cat->owns = false;
    chgpu_col_free(cat);
    return CHGPU_OK;
}
