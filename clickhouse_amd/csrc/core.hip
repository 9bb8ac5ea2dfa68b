// core.hip — context, device columns (PaddedPODArray pinned into HBM), scratch, error plumbing of libchgpu.so.
// Reference surfaces: PaddedPODArray (src/Common/PODArray.h:51-57,94-98), IColumn::cut (src/Columns/IColumn.h:118-121),
// ProfileEvents counters (src/Common/ProfileEvents.cpp:1034-1035,245-247), IProcessor::elapsed_ns (IProcessor.h:359-364).
#include "chgpu_internal.h"

#include <mutex>

static thread_local char g_last_error[512] = "";

int chgpu_set_error(int code, const char * fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char * chgpu_last_error(void) { return g_last_error; }
extern "C" int chgpu_abi_version(void) { return CHGPU_ABI_VERSION; }

extern "C" int chgpu_ctx_create(int device_id, void * hip_stream, chgpu_ctx ** out)
{
    CHGPU_REQUIRE(out, CHGPU_ERR_BAD_ARGUMENTS, "chgpu_ctx_create: out is NULL");
    int n_dev = 0;
    CHGPU_HIP(hipGetDeviceCount(&n_dev));
    CHGPU_REQUIRE(device_id >= 0 && device_id < n_dev, CHGPU_ERR_BAD_ARGUMENTS, "device %d out of range (%d devices)", device_id, n_dev);
    chgpu_ctx probe_dev;
    probe_dev.device = device_id;
    ChgpuDeviceGuard guard(&probe_dev); // the caller's current device is restored when this returns
    hipDeviceProp_t prop;
    CHGPU_HIP(hipGetDeviceProperties(&prop, device_id));
    chgpu_ctx * ctx = new chgpu_ctx();
    ctx->device = device_id;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hip_stream)
    {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->owns_stream = false;
    }
    else
    {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess)
        {
            delete ctx;
            return chgpu_set_error(CHGPU_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        ctx->owns_stream = true;
    }
    (void)hipEventCreate(&ctx->ev_start);
    (void)hipEventCreate(&ctx->ev_stop);
    *out = ctx;
    return CHGPU_OK;
}

static void ctx_teardown(chgpu_ctx * ctx)
{
    ChgpuDeviceGuard guard(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream)
    {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    for (auto & e : ctx->upload_done)
        if (e) (void)hipEventDestroy(e);
    if (ctx->upload_gate) (void)hipEventDestroy(ctx->upload_gate);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    for (auto & kv : ctx->pool_free)
        (void)hipFree(kv.second);
    ctx->pool_free.clear();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->crc_lut_dev) (void)hipFree(ctx->crc_lut_dev);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void chgpu_ctx_retain(chgpu_ctx * ctx)
{
    if (ctx)
        ++ctx->refs;
}

void chgpu_ctx_release(chgpu_ctx * ctx)
{
    if (ctx && --ctx->refs == 0 && ctx->zombie)
        ctx_teardown(ctx);
}

extern "C" int chgpu_ctx_destroy(chgpu_ctx * ctx)
{
    if (!ctx)
        return CHGPU_OK;
    if (ctx->refs > 0)
    {
        // columns / aggregations / joins made on this context are still alive and will touch it when they are freed
        // (pool_free, counters): keep it until the last of them is gone
        ctx->zombie = true;
        return CHGPU_OK;
    }
    ctx_teardown(ctx);
    return CHGPU_OK;
}

extern "C" int chgpu_ctx_synchronize(chgpu_ctx * ctx)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx, CHGPU_ERR_BAD_ARGUMENTS, "ctx is NULL");
    CHGPU_HIP(hipStreamSynchronize(ctx->stream));
    return CHGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// developer options
// ---------------------------------------------------------------------------------------------
static std::mutex g_opt_mu;
static std::map<std::string, long long> g_opt_defaults;
static const char * const CHGPU_OPTION_NAMES[] = {
    "agg_no_partition", "debug", "deterministic_float_sums", "experiment_gmajor", "experiment_join_lds", "experiment_tiles", "test_keydict_weak_tags", "tune_agg_lds_threads",
    "tune_agg_no_ranged", "tune_agg_ranged_s", "tune_cmp_wg", "tune_expr_wg", "tune_exprn_wg", "tune_fcount_wg", "tune_filter_no_multi",
    "tune_filter_no_staged", "tune_fs2_wg", "tune_fs_wg", "tune_fscatter_wg", "tune_gb_carry", "tune_gb_kib", "tune_gb_no_aos", "tune_gb_no_tiled", "tune_gb_no_word_passes",
    "tune_gb_no_two_level", "tune_gb_nocnt32", "tune_gb_noops", "tune_gb_nowide", "tune_gb_old_scatter", "tune_gb_s", "tune_gb_scatter_wgs",
    "tune_gb_tile", "tune_gb_unitdiv", "tune_jit_unroll", "tune_jit_wg_map", "tune_jit_wg_sum", "tune_join_cap_shift", "tune_join_eager_build",
    "tune_join_lds_filter_qpt", "tune_join_lds_min_rows", "tune_join_no_dense_prefilter", "tune_join_no_fused_payload", "tune_join_no_lds_filter",
    "tune_join_no_lds_filter_multi", "tune_join_no_lds_probe", "tune_join_no_prefilter", "tune_join_no_dense_map", "tune_join_no_radix", "tune_join_no_regions",
    "tune_join_no_slice_build", "tune_join_region_kib", "tune_join_region_min_rows", "tune_gb_no_tiled2", "tune_agg_no_det_f64",
};

long long chgpu_opt(const chgpu_ctx * ctx, const char * name, long long dflt)
{
    if (ctx && !ctx->options.empty())
    {
        auto it = ctx->options.find(name);
        if (it != ctx->options.end())
            return it->second;
    }
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (g_opt_defaults.empty())
        return dflt;
    auto it = g_opt_defaults.find(name);
    return it != g_opt_defaults.end() ? it->second : dflt;
}

extern "C" int chgpu_ctx_set_option(chgpu_ctx * ctx, const char * name, int64_t value)
{
    CHGPU_REQUIRE(name, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    bool known = false;
    for (const char * n : CHGPU_OPTION_NAMES)
        known = known || strcmp(n, name) == 0;
    CHGPU_REQUIRE(known, CHGPU_ERR_BAD_ARGUMENTS, "unknown option '%s'", name);
    if (ctx)
        ctx->options[name] = value;
    else
    {
        std::lock_guard<std::mutex> lk(g_opt_mu);
        g_opt_defaults[name] = value;
    }
    return CHGPU_OK;
}

extern "C" int chgpu_ctx_trim(chgpu_ctx * ctx)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx, CHGPU_ERR_BAD_ARGUMENTS, "ctx is NULL");
    CHGPU_HIP(hipStreamSynchronize(ctx->stream));
    for (auto & kv : ctx->pool_free)
        (void)hipFree(kv.second);
    ctx->pool_free.clear();
    ctx->pool_cached_bytes = 0;
    if (ctx->scratch)
        (void)hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    return CHGPU_OK;
}

extern "C" int chgpu_ctx_counters(chgpu_ctx * ctx, uint64_t out[CHGPU_N_COUNTERS])
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    memcpy(out, ctx->counters, sizeof(ctx->counters));
    return CHGPU_OK;
}

extern "C" int chgpu_timer_start(chgpu_ctx * ctx)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx, CHGPU_ERR_BAD_ARGUMENTS, "ctx is NULL");
    CHGPU_HIP(hipEventRecord(ctx->ev_start, ctx->stream));
    return CHGPU_OK;
}

extern "C" int chgpu_timer_stop_ms(chgpu_ctx * ctx, double * elapsed_ms)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && elapsed_ms, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_HIP(hipEventRecord(ctx->ev_stop, ctx->stream));
    CHGPU_HIP(hipEventSynchronize(ctx->ev_stop));
    float ms = 0;
    CHGPU_HIP(hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
    *elapsed_ms = ms;
    return CHGPU_OK;
}

int chgpu_scratch(chgpu_ctx * ctx, size_t bytes, void ** out)
{
    if (bytes > ctx->scratch_bytes)
    {
        ChgpuDeviceGuard guard(ctx);
        // growing frees the old buffer: earlier kernels on this stream may still read it -> drain first
        CHGPU_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch)
            CHGPU_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        size_t want = bytes + bytes / 4 + (1 << 20);
        CHGPU_HIP(hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return CHGPU_OK;
}

int chgpu_pinned(chgpu_ctx * ctx, size_t bytes, void ** out)
{
    if (bytes > ctx->pinned_bytes)
    {
        ChgpuDeviceGuard guard(ctx);
        CHGPU_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->pinned)
            CHGPU_HIP(hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
        size_t want = bytes < 4096 ? 4096 : bytes;
        CHGPU_HIP(hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return CHGPU_OK;
}

int chgpu_read_back(chgpu_ctx * ctx, const void * dev, void * host, size_t bytes)
{
    void * stage = nullptr;
    CHGPU_TRY(chgpu_pinned(ctx, bytes, &stage));
    CHGPU_HIP(hipMemcpyAsync(stage, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CHGPU_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(host, stage, bytes);
    return CHGPU_OK;
}

// size classes {1, 1.25, 1.5, 1.75} x 2^k: at most 25 % internal waste, few distinct classes
static size_t pool_class(size_t bytes)
{
    if (bytes < 4096)
        return 4096;
    size_t p = 4096;
    while (p * 2 <= bytes)
        p *= 2;
    for (int q = 4; q <= 8; ++q)
        if (p / 4 * q >= bytes)
            return p / 4 * q;
    return p * 2;
}

int chgpu_pool_alloc(chgpu_ctx * ctx, size_t bytes, void ** out, size_t * class_bytes)
{
    const size_t cls = pool_class(bytes);
    ChgpuDeviceGuard guard(ctx);
    auto it = ctx->pool_free.lower_bound(cls);
    if (it != ctx->pool_free.end() && it->first <= cls + cls / 2)
    {
        *out = it->second;
        *class_bytes = it->first;
        ctx->pool_cached_bytes -= it->first;
        ctx->pool_free.erase(it);
        return CHGPU_OK;
    }
    hipError_t e = hipMalloc(out, cls);
    if (e == hipErrorOutOfMemory && !ctx->pool_free.empty())
    {
        // give the cached blocks back and retry once
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);
        for (auto & kv : ctx->pool_free)
            (void)hipFree(kv.second);
        ctx->pool_free.clear();
        ctx->pool_cached_bytes = 0;
        e = hipMalloc(out, cls);
    }
    if (e != hipSuccess)
        return chgpu_set_error(e == hipErrorOutOfMemory ? CHGPU_ERR_OOM : CHGPU_ERR_DEVICE, "hipMalloc(%zu): %s", cls, hipGetErrorString(e));
    *class_bytes = cls;
    return CHGPU_OK;
}

void chgpu_pool_free(chgpu_ctx * ctx, void * p, size_t class_bytes)
{
    if (!p)
        return;
    if (ctx->pool_cached_bytes + class_bytes > ctx->pool_limit_bytes)
    {
        (void)hipFree(p); // synchronises
        return;
    }
    ctx->pool_free.emplace(class_bytes, p);
    ctx->pool_cached_bytes += class_bytes;
}

int chgpu_col_new(chgpu_ctx * ctx, int type, u64 rows, chgpu_col ** out)
{
    size_t es = chgpu_type_size(type);
    CHGPU_REQUIRE(es, CHGPU_ERR_BAD_ARGUMENTS, "unknown column type %d", type);
    ChgpuDeviceGuard guard(ctx);
    size_t bytes = rows * es + 2 * CHGPU_PAD;
    void * base = nullptr;
    size_t cls = 0;
    CHGPU_TRY(chgpu_pool_alloc(ctx, bytes, &base, &cls));
    chgpu_col * c = new chgpu_col();
    chgpu_ctx_retain(ctx);
    c->ctx = ctx;
    c->type = type;
    c->rows = rows;
    c->base = base;
    c->data = (char *)base + CHGPU_PAD;
    c->owns = true;
    c->alloc_bytes = cls;
    *out = c;
    return CHGPU_OK;
}

extern "C" int chgpu_col_alloc(chgpu_ctx * ctx, int type, uint64_t rows, chgpu_col ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    return chgpu_col_new(ctx, type, rows, out);
}

extern "C" int chgpu_col_upload(chgpu_ctx * ctx, int type, const void * host_ptr, uint64_t rows, chgpu_col ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && out && (host_ptr || rows == 0), CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    chgpu_col * c = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, type, rows, &c));
    if (rows)
    {
        hipError_t e = hipMemcpyAsync(c->data, host_ptr, rows * chgpu_type_size(type), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream); // the host buffer is the caller's: do not keep reading it
        if (e != hipSuccess)
        {
            chgpu_col_free(c);
            return chgpu_set_error(CHGPU_ERR_DEVICE, "upload: %s", hipGetErrorString(e));
        }
    }
    *out = c;
    return CHGPU_OK;
}

extern "C" int chgpu_host_alloc(size_t bytes, void ** out)
{
    CHGPU_REQUIRE(out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return CHGPU_OK;
}

extern "C" int chgpu_host_free(void * p)
{
    if (p)
        CHGPU_HIP(hipHostFree(p));
    return CHGPU_OK;
}

extern "C" int chgpu_col_upload_async(chgpu_ctx * ctx, int type, const void * pinned_host_ptr, uint64_t rows, chgpu_col ** out, uint64_t * ticket_out)
{
    CHGPU_REQUIRE(ctx && out && ticket_out && (pinned_host_ptr || rows == 0), CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    ChgpuDeviceGuard guard(ctx);
    if (!ctx->copy_stream)
    {
        CHGPU_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        CHGPU_HIP(hipEventCreateWithFlags(&ctx->upload_gate, hipEventDisableTiming));
        for (auto & e : ctx->upload_done)
            CHGPU_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    chgpu_col * c = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, type, rows, &c));
    const u64 ticket = ctx->upload_next_ticket++;
    hipEvent_t done = ctx->upload_done[ticket % chgpu_ctx::UPLOAD_RING];
    hipError_t e = hipSuccess;
    // the buffer may come from the pool: kernels queued on `stream` before now may still read its previous contents
    if ((e = hipEventRecord(ctx->upload_gate, ctx->stream)) == hipSuccess && (e = hipStreamWaitEvent(ctx->copy_stream, ctx->upload_gate, 0)) == hipSuccess)
    {
        if (rows)
            e = hipMemcpyAsync(c->data, pinned_host_ptr, rows * chgpu_type_size(type), hipMemcpyHostToDevice, ctx->copy_stream);
        if (e == hipSuccess)
            e = hipEventRecord(done, ctx->copy_stream);
        // whatever the caller launches on `stream` next sees the uploaded column
        if (e == hipSuccess)
            e = hipStreamWaitEvent(ctx->stream, done, 0);
    }
    if (e != hipSuccess)
    {
        chgpu_col_free(c);
        return chgpu_set_error(CHGPU_ERR_DEVICE, "upload_async: %s", hipGetErrorString(e));
    }
    *out = c;
    *ticket_out = ticket;
    return CHGPU_OK;
}

extern "C" int chgpu_upload_wait(chgpu_ctx * ctx, uint64_t ticket)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx, CHGPU_ERR_BAD_ARGUMENTS, "ctx is NULL");
    if (ticket == 0 || ticket >= ctx->upload_next_ticket)
        return ticket == 0 ? CHGPU_OK : chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "upload ticket %llu was never issued", (unsigned long long)ticket);
    ChgpuDeviceGuard guard(ctx);
    if (ticket + chgpu_ctx::UPLOAD_RING <= ctx->upload_next_ticket)
    {
        // its event slot has been reused by a later upload: copies on one stream complete in order, so waiting for the copy stream covers it
        CHGPU_HIP(hipStreamSynchronize(ctx->copy_stream));
        return CHGPU_OK;
    }
    CHGPU_HIP(hipEventSynchronize(ctx->upload_done[ticket % chgpu_ctx::UPLOAD_RING]));
    return CHGPU_OK;
}

extern "C" int chgpu_col_wrap(chgpu_ctx * ctx, int type, void * device_ptr, uint64_t rows, chgpu_col ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && out && (device_ptr || rows == 0), CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(chgpu_type_size(type), CHGPU_ERR_BAD_ARGUMENTS, "unknown column type %d", type);
    CHGPU_REQUIRE(((uintptr_t)device_ptr % chgpu_type_size(type)) == 0, CHGPU_ERR_BAD_ARGUMENTS, "device pointer not element-aligned");
    chgpu_col * c = new chgpu_col();
    chgpu_ctx_retain(ctx);
    c->ctx = ctx;
    c->type = type;
    c->rows = rows;
    c->data = device_ptr;
    c->owns = false;
    *out = c;
    return CHGPU_OK;
}

extern "C" int chgpu_col_slice(chgpu_ctx * ctx, const chgpu_col * col, uint64_t start, uint64_t rows, chgpu_col ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && col && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(start <= col->rows && rows <= col->rows - start, CHGPU_ERR_BAD_ARGUMENTS,
                  "cut(%llu,%llu) out of bounds of a column of %llu rows", (unsigned long long)start, (unsigned long long)rows, (unsigned long long)col->rows);
    return chgpu_col_wrap(ctx, col->type, (char *)col->data + start * chgpu_type_size(col->type), rows, out);
}

extern "C" int chgpu_col_concat(chgpu_ctx * ctx, uint32_t n, const chgpu_col * const * cols, chgpu_col ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && cols && out && n >= 1, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    u64 total = 0;
    for (u32 k = 0; k < n; ++k)
    {
        CHGPU_REQUIRE(cols[k], CHGPU_ERR_BAD_ARGUMENTS, "column %u is NULL", k);
        CHGPU_REQUIRE(cols[k]->type == cols[0]->type, CHGPU_ERR_BAD_ARGUMENTS, "columns of different types cannot be concatenated");
        total += cols[k]->rows;
    }
    chgpu_col * r = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, cols[0]->type, total, &r));
    const size_t es = chgpu_type_size(cols[0]->type);
    u64 pos = 0;
    for (u32 k = 0; k < n; ++k)
    {
        if (cols[k]->rows)
        {
            hipError_t e = hipMemcpyAsync((char *)r->data + pos * es, cols[k]->data, cols[k]->rows * es, hipMemcpyDeviceToDevice, ctx->stream);
            if (e != hipSuccess)
            {
                chgpu_col_free(r);
                return chgpu_set_error(CHGPU_ERR_DEVICE, "concat copy: %s", hipGetErrorString(e));
            }
        }
        pos += cols[k]->rows;
    }
    *out = r;
    return CHGPU_OK;
}

extern "C" int chgpu_col_download(chgpu_ctx * ctx, const chgpu_col * col, void * host_ptr, uint64_t rows)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && col && (host_ptr || rows == 0), CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(rows <= col->rows, CHGPU_ERR_SIZES_MISMATCH, "download of %llu rows from a column of %llu", (unsigned long long)rows, (unsigned long long)col->rows);
    if (rows)
    {
        CHGPU_HIP(hipMemcpyAsync(host_ptr, col->data, rows * chgpu_type_size(col->type), hipMemcpyDeviceToHost, ctx->stream));
        CHGPU_HIP(hipStreamSynchronize(ctx->stream));
    }
    return CHGPU_OK;
}

extern "C" int chgpu_col_download_many(chgpu_ctx * ctx, uint32_t n, const chgpu_col * const * cols, void * const * host_ptrs)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && (n == 0 || (cols && host_ptrs)), CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    bool any = false;
    for (u32 c = 0; c < n; ++c)
    {
        CHGPU_REQUIRE(cols[c] && (host_ptrs[c] || cols[c]->rows == 0), CHGPU_ERR_BAD_ARGUMENTS, "NULL column or host buffer %u", c);
        if (cols[c]->rows)
        {
            CHGPU_HIP(hipMemcpyAsync(host_ptrs[c], cols[c]->data, cols[c]->rows * chgpu_type_size(cols[c]->type), hipMemcpyDeviceToHost, ctx->stream));
            any = true;
        }
    }
    if (any)
        CHGPU_HIP(hipStreamSynchronize(ctx->stream)); // ONE wait for the whole Block (a result of a few rows is all latency)
    return CHGPU_OK;
}

extern "C" uint64_t chgpu_col_rows(const chgpu_col * col) { return col ? col->rows : 0; }
extern "C" int chgpu_col_type(const chgpu_col * col) { return col ? col->type : -1; }
extern "C" void * chgpu_col_device_ptr(const chgpu_col * col) { return col ? col->data : nullptr; }

extern "C" int chgpu_col_free(chgpu_col * col)
{
    if (!col)
        return CHGPU_OK;
    if (col->owns && col->base)
    {
        bool last = true;
        if (col->shared_refs)
        {
            last = --*col->shared_refs == 0;
            if (last)
                delete col->shared_refs;
        }
        // back to the context's pool: reuse is stream-ordered behind every kernel already enqueued on ctx->stream
        if (last)
        {
            if (col->alloc_bytes && col->ctx)
                chgpu_pool_free(col->ctx, col->base, col->alloc_bytes);
            else
                (void)hipFree(col->base);
        }
    }
    chgpu_ctx * ctx = col->ctx;
    delete col;
    chgpu_ctx_release(ctx);
    return CHGPU_OK;
}

// CRC32-C slice-by-8 tables for the bit-exact shard/bucket hash (src/Common/HashTable/Hash.h:63-66).
// lut[j*256+b] = crc(0, byte b at position j), lut[2048] = crc(-1, 0).
int chgpu_crc_lut(chgpu_ctx * ctx, const u32 ** lut_dev)
{
    if (!ctx->crc_lut_dev)
    {
        ChgpuDeviceGuard guard(ctx);
        std::vector<u32> lut(8 * 256 + 1);
        auto soft = [](u64 x, u32 crc) {
            for (int i = 0; i < 8; ++i)
            {
                crc ^= (u32)((x >> (8 * i)) & 0xFF);
                for (int k = 0; k < 8; ++k)
                    crc = (crc >> 1) ^ (0x82F63B78u & (0u - (crc & 1u)));
            }
            return crc;
        };
        for (int j = 0; j < 8; ++j)
            for (int b = 0; b < 256; ++b)
                lut[j * 256 + b] = soft((u64)b << (8 * j), 0);
        lut[2048] = soft(0, 0xFFFFFFFFu);
        CHGPU_HIP(hipMalloc((void **)&ctx->crc_lut_dev, lut.size() * sizeof(u32)));
        CHGPU_HIP(hipMemcpyAsync(ctx->crc_lut_dev, lut.data(), lut.size() * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
        CHGPU_HIP(hipStreamSynchronize(ctx->stream));
    }
    *lut_dev = ctx->crc_lut_dev;
    return CHGPU_OK;
}
