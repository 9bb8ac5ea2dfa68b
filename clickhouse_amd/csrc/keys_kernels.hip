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
