// chgpu_internal.h — shared host-side plumbing for libchgpu.so (context, columns, scratch, error handling) and
// device helpers (wave64 reductions, hashes).  gfx950 (MI355X / CDNA4) only: wavefront = 64 lanes, 256 CUs in 8 XCDs.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/chgpu.h"

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;
typedef int32_t i32;
typedef uint8_t u8;
typedef uint16_t u16;
typedef int16_t i16;
typedef int8_t i8;

// ---------------------------------------------------------------------------------------------
// error handling: thread-local last error, no exceptions across the C ABI
// ---------------------------------------------------------------------------------------------
int chgpu_set_error(int code, const char * fmt, ...);

#define CHGPU_HIP(expr)                                                                                       \
    do                                                                                                        \
    {                                                                                                         \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess)                                                                                 \
            return chgpu_set_error(_e == hipErrorOutOfMemory ? CHGPU_ERR_OOM : CHGPU_ERR_DEVICE, "%s: %s (%s:%d)", \
                                   #expr, hipGetErrorString(_e), __FILE__, __LINE__);                        \
    } while (0)

#define CHGPU_TRY(expr)      \
    do                       \
    {                        \
        int _rc = (expr);    \
        if (_rc != CHGPU_OK) \
            return _rc;      \
    } while (0)

#define CHGPU_REQUIRE(cond, code, ...)              \
    do                                              \
    {                                               \
        if (!(cond))                                \
            return chgpu_set_error(code, __VA_ARGS__); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// context & columns
// ---------------------------------------------------------------------------------------------
static constexpr size_t CHGPU_PAD = 64; // PaddedPODArray pad (src/Core/Defines.h:26)

struct chgpu_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int num_cus = 256;
    // growable device scratch (partials, tile counts, scan temporaries); valid until the next call on this ctx
    void * scratch = nullptr;
    size_t scratch_bytes = 0;
    // pinned host staging for small read-backs
    void * pinned = nullptr;
    size_t pinned_bytes = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    u64 counters[CHGPU_N_COUNTERS] = {0};
    u32 * crc_lut_dev = nullptr; // 8x256 slice tables + constant (CRC32-C, Hash.h:63-66)
    // Column memory pool: freed column buffers are kept (binned by size class) and handed to later allocations on the
    // SAME stream, which orders reuse after the last kernel that touched them.  hipMalloc/hipFree of multi-GB buffers
    // cost 10-400 ms and synchronise the device; an operator pipeline allocates one result column per call.
    std::multimap<size_t, void *> pool_free;
    size_t pool_cached_bytes = 0;
    size_t pool_limit_bytes = (size_t)96 << 30;
    // Objects made on this context (columns, aggregations, joins, communicators) hold a reference: chgpu_ctx_destroy with
    // children still alive only marks the context; the last child to go tears it down.  Single-threaded like the rest of a ctx.
    long refs = 0;
    bool zombie = false;
    // asynchronous uploads (chgpu_col_upload_async): their own stream, so a stripe's host-to-device copy overlaps the kernels of the
    // previous stripe on `stream`; completion events in a ring, addressed by ticket
    hipStream_t copy_stream = nullptr;
    static constexpr int UPLOAD_RING = 32;
    hipEvent_t upload_done[UPLOAD_RING] = {nullptr};
    hipEvent_t upload_gate = nullptr; // recorded on `stream` when the destination buffer is handed out; the copy waits for it
    u64 upload_next_ticket = 1;
    std::map<std::string, long long> options; // chgpu_ctx_set_option
};

// Developer options (plan-level A/B switches, launch geometry): set through chgpu_ctx_set_option, never read from the environment by an
// operator.  Looked up per context, then in the process-wide defaults (ctx == NULL), else `dflt`.
long long chgpu_opt(const chgpu_ctx * ctx, const char * name, long long dflt);

// Timing experiments that make a kernel SKIP work (wrong results) exist only in builds made with -DCHGPU_EXPERIMENTS; csrc/Makefile never
// sets it, so in the product every such switch is the constant 0 and the branches fold away.
#ifdef CHGPU_EXPERIMENTS
#define CHGPU_EXPERIMENT(ctx, name) ((int)chgpu_opt(ctx, name, 0))
#define CHGPU_EXPERIMENT_VALUE(x) (x)
#else
#define CHGPU_EXPERIMENT(ctx, name) 0
#define CHGPU_EXPERIMENT_VALUE(x) 0
#endif

void chgpu_ctx_retain(chgpu_ctx * ctx);
void chgpu_ctx_release(chgpu_ctx * ctx); // tears a zombie context down when the last reference goes

struct chgpu_col
{
    chgpu_ctx * ctx = nullptr;
    int type = 0;
    u64 rows = 0;
    void * data = nullptr; // first element
    void * base = nullptr; // allocation base when owning (data - CHGPU_PAD)
    bool owns = false;
    size_t alloc_bytes = 0; // size class of `base` when it came from the context's pool
    int * shared_refs = nullptr; // several owning columns carved out of one allocation (scatter outputs)
};

// Every entry point runs on its context's device whatever device the calling thread had current (pipeline threads migrate; a new
// thread starts on device 0): allocations, module loads and launches below must all land on ctx->device.  Restores on scope exit.
struct ChgpuDeviceGuard
{
    int prev = -1;
    bool switched = false;
    explicit ChgpuDeviceGuard(const chgpu_ctx * ctx)
    {
        if (ctx && hipGetDevice(&prev) == hipSuccess && prev != ctx->device)
            switched = hipSetDevice(ctx->device) == hipSuccess;
    }
    ~ChgpuDeviceGuard()
    {
        if (switched)
            (void)hipSetDevice(prev);
    }
    ChgpuDeviceGuard(const ChgpuDeviceGuard &) = delete;
    ChgpuDeviceGuard & operator=(const ChgpuDeviceGuard &) = delete;
};

static inline size_t chgpu_type_size(int type)
{
    switch (type)
    {
        case CHGPU_I64: case CHGPU_U64: case CHGPU_F64: return 8;
        case CHGPU_U32: case CHGPU_I32: case CHGPU_F32: return 4;
        case CHGPU_U16: case CHGPU_I16: return 2;
        case CHGPU_U8: case CHGPU_I8: return 1;
        default: return 0;
    }
}

static inline bool chgpu_type_is_float(int type) { return type == CHGPU_F64 || type == CHGPU_F32; }
static inline bool chgpu_type_is_int(int type) { return !chgpu_type_is_float(type) && chgpu_type_size(type) != 0; }
static inline bool chgpu_type_is_signed(int type) { return type == CHGPU_I64 || type == CHGPU_I32 || type == CHGPU_I16 || type == CHGPU_I8; }
// SumSimple result type (src/AggregateFunctions/AggregateFunctionSum.cpp:19-28)
static inline int chgpu_sum_result_type(int t)
{
    if (chgpu_type_is_float(t)) return CHGPU_F64;
    return chgpu_type_is_signed(t) ? CHGPU_I64 : CHGPU_U64;
}

int chgpu_scratch(chgpu_ctx * ctx, size_t bytes, void ** out);           // >= bytes, 256-B aligned
int chgpu_pool_alloc(chgpu_ctx * ctx, size_t bytes, void ** out, size_t * class_bytes);
void chgpu_pool_free(chgpu_ctx * ctx, void * p, size_t class_bytes);
int chgpu_pinned(chgpu_ctx * ctx, size_t bytes, void ** out);
int chgpu_col_new(chgpu_ctx * ctx, int type, u64 rows, chgpu_col ** out); // owning, padded
int chgpu_read_back(chgpu_ctx * ctx, const void * dev, void * host, size_t bytes); // async copy + stream sync
int chgpu_crc_lut(chgpu_ctx * ctx, const u32 ** lut_dev);                 // device LUT [8*256 + 1]

static inline u32 chgpu_grid_for(chgpu_ctx * ctx, u64 work_items, u32 block, u32 blocks_per_cu = 8)
{
    u64 want = (work_items + block - 1) / block;
    u64 cap = (u64)ctx->num_cus * blocks_per_cu;
    if (want < 1) want = 1;
    return (u32)(want < cap ? want : cap);
}

// device-wide scans (scan.hip).  tmp comes from ctx scratch *after* `scratch_offset` bytes already in use.
// exclusive: out[i] = sum(in[0..i)), *total_dev (u64) = sum(all).  in: u32, out: u64.
int chgpu_scan_exclusive_u32_u64(chgpu_ctx * ctx, const u32 * in, u64 * out, u64 n, u64 * total_dev, void * tmp, size_t tmp_bytes);
// inclusive: out[i] = sum(in[0..i]).
int chgpu_scan_inclusive_u32_u64(chgpu_ctx * ctx, const u32 * in, u64 * out, u64 n, u64 * total_dev, void * tmp, size_t tmp_bytes);
size_t chgpu_scan_tmp_bytes(u64 n);
// stable split of n_cols columns by sel[i] < num_shards (<= 256) into concatenated outputs (partition_kernels.hip); counts[num_shards] on the host
int chgpu_partition_by_key_byte(chgpu_ctx * ctx, const chgpu_col * keys, u32 shift, u32 n_cols, const chgpu_col * const * cols, chgpu_col ** outs);
int chgpu_partition_core(chgpu_ctx * ctx, const u32 * sel, u64 n, u32 num_shards, u32 n_cols, const chgpu_col * const * cols, chgpu_col ** outs, u64 * counts);

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__

static constexpr int WAVE = 64;

__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ u32 mbcnt(u64 mask)
{
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

__device__ __forceinline__ u64 shfl_down_u64(u64 v, int delta)
{
    u32 lo = __shfl_down((u32)v, delta, WAVE);
    u32 hi = __shfl_down((u32)(v >> 32), delta, WAVE);
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 wave_reduce_add_u64(u64 v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += shfl_down_u64(v, d);
    return v; // valid in lane 0
}

__device__ __forceinline__ double wave_reduce_add_f64(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __longlong_as_double((long long)shfl_down_u64((u64)__double_as_longlong(v), d));
    return v;
}

__device__ __forceinline__ u32 wave_reduce_add_u32(u32 v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_down(v, d, WAVE);
    return v;
}

// murmur finalizer == the reference's intHash64 (src/Common/HashTable/Hash.h:27-36): the device tables' placement hash.
__device__ __forceinline__ u64 dev_intHash64(u64 x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

// CRC32-C of one 64-bit key with seed, via 8 byte-sliced tables (lut[j*256+b]) and the affine constant for the seed:
// crc(seed, x) = crc(seed, 0) ^ XOR_j lut[j][byte_j(x)]; crc(seed,0) itself is linear in seed: precomputed for
// seed = -1 in lut[2048]; for other seeds use dev_crc32c_seeded.
__device__ __forceinline__ u32 dev_crc32c_tab(const u32 * __restrict__ lut, u64 x)
{
    u32 r = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        r ^= lut[j * 256 + ((x >> (8 * j)) & 0xFF)];
    return r;
}

// crc32c of 8 zero bytes starting from `seed` (bitwise; used only for non-default seeds)
__device__ __forceinline__ u32 dev_crc32c_zero8(u32 crc)
{
#pragma unroll 1
    for (int k = 0; k < 64; ++k)
        crc = (crc >> 1) ^ (0x82F63B78u & (0u - (crc & 1u)));
    return crc;
}

#endif // __HIPCC__
