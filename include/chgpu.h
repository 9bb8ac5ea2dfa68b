/*
 * chgpu.h — C ABI of the MI355X-native block-processing hot path (filter -> aggregate -> hash join).
 *
 * This is the drop-in boundary.  The reference (filimonov/ClickHouse) has no C ABI for this path; its extension
 * points are C++ virtual interfaces.  Every entry point below names the reference interface it stands in for
 * (file:line relative to the reference checkout); INTEGRATION.md shows the C++ shim a maintainer adds on the
 * reference side (GpuFilterTransform : ISimpleTransform, GpuAggregator, GpuHashJoin : IJoin) that calls these.
 *
 * Conventions
 *   - plain C, opaque handles, no torch/HIP types in signatures (a hipStream_t travels as void*);
 *   - every call returns an int status: 0 = OK, negative = error mapped from the reference's error names;
 *     chgpu_last_error() gives the message of the calling thread's last failure; nothing throws across the ABI;
 *   - the caller owns host buffers; the library owns device buffers behind handles;
 *   - a chgpu_ctx binds one device + one HIP stream; N host threads drive N contexts concurrently (the
 *     threading contract of IProcessor::work(), src/Processors/IProcessor.h:176-193); one ctx is single-threaded, but MAY be used
 *     from different threads one after another: every entry point makes the context's device current for the duration of
 *     the call and restores the caller's (a fresh pipeline thread starts on device 0);
 *   - CHGPU_ERR_NOT_IMPLEMENTED means "fall back to the CPU path" (mirror of isCompilable() gating,
 *     src/AggregateFunctions/AggregateFunctionSum.h:590-604).
 */
#ifndef CHGPU_H
#define CHGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHGPU_ABI_VERSION 1

/* ---- status codes (names follow src/Common/ErrorCodes.cpp) ---- */
enum
{
    CHGPU_OK = 0,
    CHGPU_ERR_SIZES_MISMATCH = -1,  /* SIZES_OF_COLUMNS_DOESNT_MATCH (ColumnVector.cpp:685-686) */
    CHGPU_ERR_NOT_IMPLEMENTED = -2, /* NOT_IMPLEMENTED: caller falls back to CPU */
    CHGPU_ERR_OOM = -3,             /* CANNOT_ALLOCATE_MEMORY */
    CHGPU_ERR_LOGICAL = -4,         /* LOGICAL_ERROR */
    CHGPU_ERR_BAD_ARGUMENTS = -5,   /* BAD_ARGUMENTS */
    CHGPU_ERR_DEVICE = -6,          /* a HIP runtime call failed */
    CHGPU_ERR_TOO_MANY_ROWS = -7    /* TOO_MANY_ROWS (HashJoin.cpp:563-564: block >= 2^32 rows) */
};

/* ---- column element types (the TypeIndex subset of the hot path, src/Core/TypeId.h) ----
   Every entry point takes every type unless it says otherwise; arithmetic (chgpu_arith) and the fused expression kernel
   (chgpu_expr_filter_sum) are limited to the first six and answer CHGPU_ERR_NOT_IMPLEMENTED for UInt16 / Int16 / Int8 /
   Float32; keys (GROUP BY, join, sharding, packed) are integers. */
enum
{
    CHGPU_I64 = 0,
    CHGPU_U32 = 1,
    CHGPU_U64 = 2,
    CHGPU_F64 = 3,
    CHGPU_U8 = 4,
    CHGPU_I32 = 5,
    CHGPU_U16 = 6, /* also Date (days since epoch) */
    CHGPU_I16 = 7,
    CHGPU_I8 = 8,
    CHGPU_F32 = 9  /* sums and averages accumulate in Float64 (SumSimple: NearestFieldType<Float32>) */
};

/* ---- comparison functions (src/Functions/FunctionsComparison.h: equals..greaterOrEquals) ---- */
enum { CHGPU_EQ = 0, CHGPU_NE = 1, CHGPU_LT = 2, CHGPU_GT = 3, CHGPU_LE = 4, CHGPU_GE = 5 };

/* ---- aggregate functions with POD states the device can hold (src/AggregateFunctions/) ---- */
enum { CHGPU_AGG_COUNT = 0, CHGPU_AGG_SUM = 1, CHGPU_AGG_AVG = 2,
       /* min / max over a numeric argument, result in the argument's type (AggregateFunctionsMinMax.cpp, SingleValueDataFixed: SingleValueData.cpp:
          219-262); with a GROUP BY key (hash table states) or without (one reduction per block).  The 8-byte state word is an order key: merge /
          export / import as for the sums, but never through the wire serialisation of chgpu_agg_serialize_states. */
       CHGPU_AGG_MIN = 3, CHGPU_AGG_MAX = 4,
       /* any(x): the value of the group's FIRST row in the order the blocks were added (AggregateFunctionAny.cpp: setIfFirst; merges keep the
          state that already has a value -- changeFirstTime).  Two 8-byte state words {claim, value}: the claim names the earliest row, so the
          result does not depend on the order the hardware serves the rows in. */
       CHGPU_AGG_ANY = 5 };

/* ---- JoinKind / JoinStrictness subset (src/Core/Joins.h) ---- */
enum { CHGPU_JOIN_INNER = 0, CHGPU_JOIN_LEFT = 1, CHGPU_JOIN_RIGHT = 2, CHGPU_JOIN_FULL = 3 }; /* RIGHT / FULL: strictness ALL only */
enum { CHGPU_STRICT_ANY = 0, CHGPU_STRICT_ALL = 1, CHGPU_STRICT_SEMI = 2, CHGPU_STRICT_ANTI = 3 };

typedef struct chgpu_ctx chgpu_ctx;
typedef struct chgpu_col chgpu_col;
typedef struct chgpu_agg chgpu_agg;
typedef struct chgpu_join chgpu_join;

/* ================================================================================================
 * context
 * ============================================================================================== */
int chgpu_abi_version(void);
const char * chgpu_last_error(void);
/* hip_stream: an existing hipStream_t to launch on (e.g. the host framework's stream), or NULL for a private one */
int chgpu_ctx_create(int device_id, void * hip_stream, chgpu_ctx ** out);
/* Columns, aggregations, joins and communicators made on a context keep it alive: destroying a context that still has children only
   marks it, and the last child to be freed tears it down (no use-after-free whatever order a garbage collector picks). */
int chgpu_ctx_destroy(chgpu_ctx * ctx);
int chgpu_ctx_synchronize(chgpu_ctx * ctx);
/* give the context's cached device memory (column pool + scratch arena) back to the driver; synchronizes */
int chgpu_ctx_trim(chgpu_ctx * ctx);
/* Developer options: plan-level A/B switches and launch geometry (names: tools/README.md; e.g. "tune_join_no_radix", "tune_gb_tile").  No
   operator reads the environment: an option is set here, per context, or with ctx == NULL as the process-wide default.  Unknown names ->
   CHGPU_ERR_BAD_ARGUMENTS.  Results never depend on an option (only the plan taken does); the timing experiments that skip work exist only in
   builds made with -DCHGPU_EXPERIMENTS. */
int chgpu_ctx_set_option(chgpu_ctx * ctx, const char * name, int64_t value);
/* ProfileEvents-style counters of this context (src/Common/ProfileEvents.cpp:1034-1035, 245-247):
   [0] FilterTransformPassedRows [1] FilterTransformPassedBytes [2] JoinBuildTableRowCount
   [3] JoinProbeTableRowCount [4] JoinResultRowCount [5] AggregatedRows [6] kernel launches [7] table rehashes */
#define CHGPU_N_COUNTERS 8
int chgpu_ctx_counters(chgpu_ctx * ctx, uint64_t out[CHGPU_N_COUNTERS]);
/* HIP-event stopwatch on the context's stream (IProcessor::elapsed_ns analogue, IProcessor.h:359-364) */
int chgpu_timer_start(chgpu_ctx * ctx);
int chgpu_timer_stop_ms(chgpu_ctx * ctx, double * elapsed_ms); /* synchronizes on the stop event */

/* ================================================================================================
 * a1/a2 columns: PaddedPODArray<T> pinned into HBM (src/Common/PODArray.h:51-57; IColumn::Filter =
 * PaddedPODArray<UInt8>, src/Columns/FilterDescription.h:14).  Device buffers are over-allocated by 64 B on both
 * sides like the reference's pad so kernels may over-read a vector lane.
 * ============================================================================================== */
int chgpu_col_upload(chgpu_ctx * ctx, int type, const void * host_ptr, uint64_t rows, chgpu_col ** out);
int chgpu_col_alloc(chgpu_ctx * ctx, int type, uint64_t rows, chgpu_col ** out);
/* Pinned, double-buffered staging (the StripeBuilder of the host shim): chgpu_host_alloc gives page-locked host memory; an upload from it
   is queued on the context's COPY stream and returns at once -- the copy overlaps whatever kernels are already queued on the
   context's stream, and everything the caller launches on that stream afterwards sees the column (an event orders them).  The host
   buffer may be refilled once chgpu_upload_wait(ticket) has returned. */
int chgpu_host_alloc(size_t bytes, void ** out);
int chgpu_host_free(void * p);
int chgpu_col_upload_async(chgpu_ctx * ctx, int type, const void * pinned_host_ptr, uint64_t rows, chgpu_col ** out, uint64_t * ticket_out);
int chgpu_upload_wait(chgpu_ctx * ctx, uint64_t ticket);
/* non-owning view of caller-managed HBM (columns already resident on the device) */
int chgpu_col_wrap(chgpu_ctx * ctx, int type, void * device_ptr, uint64_t rows, chgpu_col ** out);
/* IColumn::cut(start, length) as a non-owning view (src/Columns/IColumn.h:118-121) */
int chgpu_col_slice(chgpu_ctx * ctx, const chgpu_col * col, uint64_t start, uint64_t rows, chgpu_col ** out);
/* concatenation of n columns of one type (the way right-side Blocks are glued into one payload column: ColumnVector::insertRangeFrom,
   src/Columns/ColumnVector.cpp:511-526) */
int chgpu_col_concat(chgpu_ctx * ctx, uint32_t n, const chgpu_col * const * cols, chgpu_col ** out);
int chgpu_col_download(chgpu_ctx * ctx, const chgpu_col * col, void * host_ptr, uint64_t rows);
/* every column of a (result) Block to host memory with ONE wait: host_ptrs[c] receives all rows of cols[c] */
int chgpu_col_download_many(chgpu_ctx * ctx, uint32_t n, const chgpu_col * const * cols, void * const * host_ptrs);
uint64_t chgpu_col_rows(const chgpu_col * col);
int chgpu_col_type(const chgpu_col * col);
void * chgpu_col_device_ptr(const chgpu_col * col);
int chgpu_col_free(chgpu_col * col);

/* ================================================================================================
 * §8(f) rank 3 — the feed side: compressed frames decoded in HBM.  A MergeTree column file / compressed Native stream is a
 * sequence of frames: 16-byte checksum, method byte, compressed size (incl. the 9-byte header), decompressed size, payload
 * (src/Compression/CompressedReadBufferBase.cpp:175-222, CompressionInfo.h:10-51).  The host walks the frame headers and verifies the
 * checksums (chgpu_compressed_walk_frames / chgpu_read_compressed_column below), uploads the compressed bytes once and names the
 * frames; every frame is decoded by one wavefront straight into the output buffer (LZ4 block format as in
 * LZ4_decompress_faster.cpp:470-684; method 0x02 NONE copies).  CODEC(Delta, LZ4) -- a Multiple frame (0x91,
 * CompressionCodecMultiple.cpp:68-130) whose methods are {0x92, 0x82} -- is two stages: the host names the inner LZ4 stage's payload
 * and output size (stage_sizes) and sets post_methods[f] = 0x92; the Delta stage (CompressionCodecDelta.cpp:84-175: running sums) checks
 * its own header on the device.  post_methods / stage_sizes may be NULL.  Other methods -> CHGPU_ERR_NOT_IMPLEMENTED; a malformed frame ->
 * CHGPU_ERR_BAD_ARGUMENTS (CANNOT_DECOMPRESS), never a fault.  chgpu_col_from_bytes turns a byte range of the result into a typed
 * column (SerializationNumber::deserializeBinaryBulk: plain little-endian arrays).
 * ============================================================================================== */
/* The frame walk itself, with the reference's integrity checks (CompressedReadBufferBase.cpp:49-127,130-222): every frame's checksum --
   CityHash128 (cityhash 1.0.2) over header + payload, stored in front of it as {low64, high64} -- is verified unless verify_checksums == 0
   (the reference's disable_checksum), compressed / decompressed sizes above 1 GiB (DBMS_MAX_COMPRESSED_SIZE) are refused, a Multiple{Delta,
   LZ4} frame is split into its stages.  `file` is host memory.  Fills up to `capacity` entries of the arrays chgpu_decompress_frames takes
   and returns the number of frames (call with capacity 0 to validate and count).  A mismatch answers CHGPU_ERR_BAD_ARGUMENTS with the
   reference's message ("Checksum doesn't match: corrupted data." ...). */
int chgpu_compressed_walk_frames(const uint8_t * file, uint64_t size, int verify_checksums, uint32_t capacity, uint32_t * n_frames, uint64_t * payload_offsets,
                                 uint32_t * payload_sizes, uint32_t * decompressed_sizes, uint8_t * methods, uint8_t * post_methods, uint32_t * stage_sizes);
/* CompressedReadBuffer + SerializationNumber::deserializeBinaryBulk in one call: a MergeTree `<column>.bin` / compressed Native stream of a
   numeric column in host memory -> a typed column in HBM (walk + verify on the host, the compressed bytes cross PCIe, decode on the device) */
int chgpu_read_compressed_column(chgpu_ctx * ctx, const uint8_t * file, uint64_t size, int type, int verify_checksums, chgpu_col ** out);
/* CityHash_v1_0_2::CityHash128 (the checksum of a frame a writer has to put in front of it): out = {low64, high64} */
int chgpu_city_hash128(const void * data, uint64_t size, uint64_t out_low_high[2]);
/* Native format (src/Formats/NativeReader.cpp:113-260): the header walk of ONE block in host memory.  [BlockInfo when server_revision > 0:
   (field, value)* 0 with field 1 = is_overflows, 2 = bucket_num (src/Core/BlockInfo.cpp:38-62)] columns, rows (VarUInt), then per column
   name, type name (strings), [a custom-serialization flag byte from revision 54454 on] and the values.  The walk describes where every
   column's parts lie; the caller moves them to HBM:
     numbers (kind NUMERIC): rows x sizeof(T) little-endian bytes at data_offset (chgpu_col_upload / chgpu_col_upload_async);
     FixedString(N) (FIXED_STRING): rows x N bytes at data_offset, a UInt8 column for chgpu_fixed_string_word;
     String (STRING): (VarUInt length, bytes) per value in [data_offset, +data_bytes), chars_bytes of them value bytes -> chgpu_native_read_strings;
     Nullable(T) of these: is_nullable = 1 and the null map's rows bytes at null_map_offset, in front of the nested values
       (SerializationNullable; chgpu_filter_description_nullable / NullableKeyAggregator take the map);
     LowCardinality(String) and LowCardinality(Nullable(String)) (LC_STRING): SerializationLowCardinality.cpp:84-164,:560-700 -- keys version
       (must be 1), index type + flags (a global dictionary is refused, additional keys are required: the reference's messages), lc_num_keys
       keys serialized as Strings in [lc_keys_offset, +lc_keys_bytes) (chgpu_native_read_strings; is_nullable: key 0 stands for NULL), then
       rows indexes of `type` (UInt8/16/32/64) at data_offset, each checked against lc_num_keys ("Index for LowCardinality is out of
       range", ColumnLowCardinality.cpp:240-252).  Indexes + dictionary are a ColumnLowCardinality for chgpu_lc_remap / GROUP BY / joins.
   Any other type answers CHGPU_ERR_NOT_IMPLEMENTED.  *bytes_consumed = where the next block starts.
   Parity pin: tests/golden/native_lc_blocks.json = the bytes and messages of tests/queries/0_stateless/02010_lc_native.{python,reference}. */
enum
{
    CHGPU_NATIVE_NUMERIC = 0,
    CHGPU_NATIVE_STRING = 1,
    CHGPU_NATIVE_FIXED_STRING = 2,
    CHGPU_NATIVE_LC_STRING = 3
};
typedef struct chgpu_native_column
{
    char name[64];
    char type_name[64];
    int32_t type;         /* CHGPU_* element type: the values' (NUMERIC), UInt8 (STRING / FIXED_STRING), the indexes' (LC_STRING) */
    int32_t kind;         /* CHGPU_NATIVE_* */
    uint32_t fixed_n;     /* FixedString(N) */
    int32_t is_nullable;
    uint64_t data_offset; /* byte offset of the column's values inside `data` */
    uint64_t data_bytes;
    uint64_t null_map_offset;
    uint64_t chars_bytes; /* STRING: sum of the value lengths */
    uint64_t lc_num_keys, lc_keys_offset, lc_keys_bytes, lc_keys_chars_bytes;
} chgpu_native_column;
int chgpu_native_walk_block(const uint8_t * data, uint64_t size, uint64_t server_revision, uint32_t capacity, chgpu_native_column * columns, uint32_t * n_columns,
                            uint64_t * n_rows, int32_t * bucket_num, int * is_overflows, uint64_t * bytes_consumed);
/* `rows` serialized String values (SerializationString::deserializeBinaryBulk) in host memory -> a ColumnString in HBM */
int chgpu_native_read_strings(chgpu_ctx * ctx, const uint8_t * serialized, uint64_t bytes, uint64_t rows, chgpu_col ** offsets_u64, chgpu_col ** chars_u8);
int chgpu_decompress_frames(chgpu_ctx * ctx, const chgpu_col * compressed_u8, uint32_t n_frames, const uint64_t * payload_offsets,
                            const uint32_t * payload_sizes, const uint32_t * decompressed_sizes, const uint8_t * methods,
                            const uint8_t * post_methods, const uint32_t * stage_sizes, chgpu_col ** out_u8);
int chgpu_col_from_bytes(chgpu_ctx * ctx, const chgpu_col * bytes_u8, uint64_t byte_offset, int type, uint64_t rows, chgpu_col ** out);

/* ================================================================================================
 * a3 comparison  —  IFunction::executeImpl for less/greater/equals... with a constant right argument:
 * NumComparisonImpl<A,B,Op>::vectorConstant (src/Functions/FunctionsComparison.h:204-245), semantics
 * accurate::lessOp/equalsOp (src/Core/AccurateComparison.h:20-130).  mask[i] = Op(col[i], scalar) ? 1 : 0.
 * Supported: integer column x integer scalar of any signedness, F64 x F64, integer column x F64 scalar and F64 column x
 * integer scalar -- all compared mathematically (a constant no value of the column type can equal compares accordingly;
 * NaN is unequal to everything and unordered).
 * ============================================================================================== */
int chgpu_cmp_const(chgpu_ctx * ctx, const chgpu_col * col, int op, int scalar_type, const void * scalar,
                    chgpu_col ** mask_u8);

/* ================================================================================================
 * a4/a5 filter  —  IColumn::filter(const Filter &, ssize_t result_size_hint) (src/Columns/IColumn.h:313-314;
 * ColumnVector<T>::filter, src/Columns/ColumnVector.cpp:682-724).  Order-preserving; any non-zero mask byte keeps
 * the row; size mismatch -> CHGPU_ERR_SIZES_MISMATCH.  result_size_hint: <0 reserve rows, >0 reserve hint, 0 exact.
 * ============================================================================================== */
int chgpu_count_bytes_in_filter(chgpu_ctx * ctx, const chgpu_col * mask_u8, uint64_t * count); /* ColumnsCommon.cpp:31-58 */
int chgpu_filter(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * mask_u8, int64_t result_size_hint,
                 chgpu_col ** out, uint64_t * out_rows);
/* The same for every column of a Block at once -- FilterTransform::doTransform's loop over the chunk's columns
   (src/Processors/Transforms/FilterTransform.cpp:238-252) and joinBlock's filtering of the left columns
   (src/Interpreters/HashJoin/HashJoinMethodsImpl.h:122-123): the mask is counted and scanned once, one host synchronisation
   for the whole Block.  outs[k] receives column k filtered; all have out_rows rows. */
int chgpu_filter_columns(chgpu_ctx * ctx, uint32_t n_cols, const chgpu_col * const * cols, const chgpu_col * filter_u8,
                         int64_t result_size_hint, chgpu_col ** outs, uint64_t * out_rows);
/* a6 FilterDescription for Nullable(UInt8): res = data && !null (src/Columns/FilterDescription.cpp:86-92) */
int chgpu_filter_description_nullable(chgpu_ctx * ctx, const chgpu_col * data_u8, const chgpu_col * null_u8, chgpu_col ** out);

/* ================================================================================================
 * a8/a9 aggregate functions without key  —  IAggregateFunction::addBatchSinglePlace
 * (src/AggregateFunctions/IAggregateFunction.h:254-260): AggregateFunctionSumData::addMany /
 * addManyConditional (AggregateFunctionSum.h:62-236).  state8 is the 8-byte host-side state (Int64 for signed,
 * UInt64 for unsigned, Float64 for floats: SumSimple, AggregateFunctionSum.cpp:19-28); the batch sum is ADDED to it.
 * ============================================================================================== */
int chgpu_sum_add_many(chgpu_ctx * ctx, const chgpu_col * col, uint64_t row_begin, uint64_t row_end, void * state8);
int chgpu_sum_add_many_conditional(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * cond_u8,
                                   uint64_t row_begin, uint64_t row_end, void * state8);

/* Fused a3+a5+a8/a9 for `SELECT sum(val), count() WHERE pred <op> scalar` (FilterTransform::doTransform,
 * src/Processors/Transforms/FilterTransform.cpp:136-256 -> Aggregator::executeWithoutKeyImpl,
 * src/Interpreters/Aggregator.cpp:1276-1321): one pass over HBM, no mask, no filtered column.
 * val may equal pred.  sum_out: 8 bytes typed like SumSimple(val type). */
int chgpu_filter_sum(chgpu_ctx * ctx, const chgpu_col * pred, int op, int scalar_type, const void * scalar,
                     const chgpu_col * val, void * sum_out, uint64_t * count_out);
/* same, no host synchronisation: result_u64x2 is a 2-row CHGPU_U64 device column receiving {sum bits, count} */
int chgpu_filter_sum_async(chgpu_ctx * ctx, const chgpu_col * pred, int op, int scalar_type, const void * scalar,
                           const chgpu_col * val, chgpu_col * result_u64x2);

/* ================================================================================================
 * SURVEY §8(f) rank 1 — pieces of the expression DAG an SSB Q1.1-style query needs, and their fusion.
 * `and`: FunctionAnd over UInt8 columns (AndImpl::apply = a & b, src/Functions/FunctionsLogical.h:82-96).
 * multiply / plus / minus: FunctionBinaryArithmetic (src/Functions/multiply.cpp:10-28) with NumberTraits result types
 * (src/DataTypes/NumberTraits.h:73-87): integer operands of <= 8 bytes give a 64-bit result, unsigned only for
 * multiply/plus of two unsigned operands; wrap-around arithmetic.  Float operands -> CHGPU_ERR_NOT_IMPLEMENTED.
 * ============================================================================================== */
enum { CHGPU_VAL_COL = 0, CHGPU_VAL_MUL = 1, CHGPU_VAL_PLUS = 2, CHGPU_VAL_MINUS = 3 };
int chgpu_and(chgpu_ctx * ctx, const chgpu_col * a_u8, const chgpu_col * b_u8, chgpu_col ** out_u8);
int chgpu_arith(chgpu_ctx * ctx, int value_op, const chgpu_col * a, const chgpu_col * b, chgpu_col ** out);
/* Fused `SELECT sum(<value>), count() WHERE p_0 AND p_1 ...` in ONE pass over HBM (ExpressionActions::execute +
 * FilterTransform + AggregatingTransform without key).  cols[n_cols] are the distinct columns touched (<= 4, all of one
 * element width); predicate k is `cols[pred_col[k]] pred_op[k] constant` (constant = 8 raw bytes in pred_scalar_bits[k], typed
 * pred_scalar_type[k]; <= 8 predicates, and-ed); value = cols[val_a] (CHGPU_VAL_COL) or cols[val_a] <op> cols[val_b].
 * sum_out: 8 bytes, Int64 or UInt64 per the result type rules above (*result_type_out tells which). */
int chgpu_expr_filter_sum(chgpu_ctx * ctx, uint32_t n_cols, const chgpu_col * const * cols, uint32_t n_preds,
                          const uint32_t * pred_col, const int * pred_op, const int * pred_scalar_type,
                          const uint64_t * pred_scalar_bits, int value_op, uint32_t val_a, uint32_t val_b,
                          int * result_type_out, void * sum_out, uint64_t * count_out);

/* ------------------------------------------------------------------------------------------------
 * The general form: an expression DAG compiled at run time into ONE kernel (hiprtc, gfx950) -- ExpressionActions::execute
 * (src/Interpreters/ExpressionActions.cpp:595-747) without materialised intermediates; the counterpart of the reference's
 * compile_expressions JIT (src/Interpreters/JIT/CHJIT.cpp, compileFunction.cpp).  A program is an array of nodes in
 * topological order (operands precede their users); result types follow src/DataTypes/NumberTraits.h, comparisons
 * src/Core/AccurateComparison.h (mixed-sign and integer-vs-float operands compared mathematically; NaN as there), logical
 * functions see static_cast<bool>(x) (FunctionsLogical.cpp:81,424), date functions take a Date (CHGPU_U16 day number).
 * Type combinations the reference answers with a type this path does not carry (128-bit, Float -> integer casts, if() over
 * UInt64 and a signed type) give CHGPU_ERR_NOT_IMPLEMENTED at compile time: the caller keeps its CPU actions.
 * ---------------------------------------------------------------------------------------------- */
enum { CHGPU_EX_INPUT = 0, CHGPU_EX_CONST = 1, CHGPU_EX_FUNC = 2 };
enum
{
    CHGPU_FN_EQUALS = 0, CHGPU_FN_NOT_EQUALS = 1, CHGPU_FN_LESS = 2, CHGPU_FN_GREATER = 3, CHGPU_FN_LESS_OR_EQUALS = 4,
    CHGPU_FN_GREATER_OR_EQUALS = 5,                                   /* = CHGPU_EQ .. CHGPU_GE; result UInt8 */
    CHGPU_FN_PLUS = 10, CHGPU_FN_MINUS = 11, CHGPU_FN_MULTIPLY = 12,  /* ResultOfAdditionMultiplication / ResultOfSubtraction */
    CHGPU_FN_DIVIDE = 13,                                             /* Float64 (ResultOfFloatingPointDivision) */
    CHGPU_FN_NEGATE = 14,                                             /* ResultOfNegate */
    CHGPU_FN_INT_DIV = 15, CHGPU_FN_MODULO = 16,                      /* integers, constant divisor that cannot raise ILLEGAL_DIVISION */
    CHGPU_FN_AND = 20, CHGPU_FN_OR = 21, CHGPU_FN_XOR = 22, CHGPU_FN_NOT = 23, /* UInt8 */
    CHGPU_FN_IF = 30,                                                 /* if(cond, then, else): ResultOfIf */
    CHGPU_FN_BIT_AND = 40, CHGPU_FN_BIT_OR = 41, CHGPU_FN_BIT_XOR = 42, /* integers: ResultOfBit */
    CHGPU_FN_TO_YEAR = 50, CHGPU_FN_TO_MONTH = 51, CHGPU_FN_TO_DAY_OF_MONTH = 52, CHGPU_FN_TO_YYYYMM = 53, /* Date -> UInt16/UInt8/UInt8/UInt32 */
    CHGPU_FN_TO_YYYYMMDD = 54, CHGPU_FN_TO_DAY_OF_WEEK = 55, CHGPU_FN_TO_QUARTER = 56, CHGPU_FN_TO_START_OF_MONTH = 57, /* -> UInt32/UInt8/UInt8/Date */
    CHGPU_FN_CAST = 64                                                /* CHGPU_FN_CAST + CHGPU_<type>: toInt64(x) ... (static_cast) */
};
typedef struct chgpu_expr_node
{
    int32_t kind;    /* CHGPU_EX_* */
    int32_t code;    /* INPUT: index into cols[] (< 8); FUNC: CHGPU_FN_*; CONST: unused */
    int32_t type;    /* INPUT, CONST: element type; FUNC: ignored (inferred, see chgpu_expr_node_type) */
    int32_t args[3]; /* FUNC: operand node indices, unused = -1 */
    uint64_t bits;   /* CONST: the value's raw little-endian bytes */
} chgpu_expr_node;
typedef struct chgpu_expr chgpu_expr;
/* type-checks the DAG and generates its row function; needs no device */
int chgpu_expr_compile(uint32_t n_nodes, const chgpu_expr_node * nodes, chgpu_expr ** out);
int chgpu_expr_node_type(const chgpu_expr * expr, uint32_t node, int * type_out);
/* runs the run-time compiler only (no device): n_outputs > 0 -> the materialising kernel for out_nodes (with filter_node >= 0: the
   WHERE + projection pair of chgpu_expr_filter_execute), n_outputs == 0 -> the fused filter + sum kernel for (filter_node, value_node) */
int chgpu_expr_precompile(const chgpu_expr * expr, uint32_t n_outputs, const uint32_t * out_nodes, int filter_node, int value_node,
                          uint64_t * code_bytes_out);
/* materialise out_nodes[n_outputs] (<= 8) as new columns over cols[n_cols] (the INPUT nodes' columns, all of one length) */
int chgpu_expr_execute(chgpu_ctx * ctx, const chgpu_expr * expr, uint32_t n_cols, const chgpu_col * const * cols,
                       uint32_t n_outputs, const uint32_t * out_nodes, chgpu_col ** outs);
/* SELECT sum(value_node), count() WHERE filter_node in one pass, nothing materialised (filter_node < 0: no WHERE;
   value_node < 0: count only).  sum_out: 8 bytes of the SumSimple type of the value node (*result_type_out). */
int chgpu_expr_filter_sum_node(chgpu_ctx * ctx, const chgpu_expr * expr, uint32_t n_cols, const chgpu_col * const * cols,
                               int filter_node, int value_node, int * result_type_out, void * sum_out, uint64_t * count_out);
/* WHERE filter_node + projection of out_nodes (<= 7; INPUT nodes pass columns through) in one step: only the rows whose filter value
   is non-zero are written, in order -- FilterTransform (FilterTransform.cpp:136-256) fused behind the ExpressionTransform, no mask
   and no unfiltered intermediate in HBM.  *rows_out = surviving rows (0: the chunk is dropped; the outputs are empty columns). */
int chgpu_expr_filter_execute(chgpu_ctx * ctx, const chgpu_expr * expr, uint32_t n_cols, const chgpu_col * const * cols,
                              uint32_t filter_node, uint32_t n_outputs, const uint32_t * out_nodes, chgpu_col ** outs,
                              uint64_t * rows_out);
/* SELECT min(value_node), max(value_node), count() WHERE filter_node in one pass (AggregateFunctionMin / Max without key,
   src/AggregateFunctions/AggregateFunctionMinMax.h) for INTEGER value nodes; min_out / max_out receive the value in the node's own
   width (*value_type_out), 0 when no row passes.  Float values -> CHGPU_ERR_NOT_IMPLEMENTED (the reference keeps a NaN that arrives
   first: an order-dependent result). */
int chgpu_expr_filter_minmax_node(chgpu_ctx * ctx, const chgpu_expr * expr, uint32_t n_cols, const chgpu_col * const * cols,
                                  int filter_node, uint32_t value_node, int * value_type_out, void * min_out, void * max_out,
                                  uint64_t * count_out);
int chgpu_expr_free(chgpu_expr * expr);

/* ================================================================================================
 * ASOF joins (JoinStrictness::Asof, INNER and LEFT: joinDispatch.h:66-67).  The reference keeps, per join key, a SortedLookupVector of
 * (asof value, row) and answers a left row with the closest right row under the join's inequality (src/Interpreters/RowRefs.cpp:40-215:
 * insert / findAsof -> boundSearch; HashJoinMethodsImpl.h:462-478).  chgpu_asof is that map for one fixed-width integer key and one numeric
 * asof column: chgpu_asof_add_block = addBlockToJoin (rows with a NULL key, a zero ON mask or a NaN asof value are not inserted), the
 * first probe sorts (the vectors are immutable from then on, RowRefs.cpp:174-178).  chgpu_asof_probe = joinBlock: at most one right row
 * per left row.  INNER: *filter_u8 marks the left rows that found one (need_filter), right_rowid_u64 holds their (block << 32 | row) ids in
 * order, *n_out their number.  LEFT: filter_u8 may be NULL; right_rowid_u64 has one entry per left row, all-ones = the default row
 * (add_missing).  inequality as ASOFJoinInequality (src/Core/Joins.h:78-85) with the LEFT value on the left: GREATER_OR_EQUALS (the
 * default, `a.t >= b.t`) takes the greatest b.t <= a.t.  Right rows with equal (key, asof): the reference's choice is unspecified; here the
 * last inserted for >= / >, the first inserted for <= / <.
 * ============================================================================================== */
enum { CHGPU_ASOF_LESS = 1, CHGPU_ASOF_GREATER = 2, CHGPU_ASOF_LESS_OR_EQUALS = 3, CHGPU_ASOF_GREATER_OR_EQUALS = 4 };
typedef struct chgpu_asof chgpu_asof;
int chgpu_asof_create(chgpu_ctx * ctx, int key_type, int asof_type, int kind, int inequality, chgpu_asof ** out);
int chgpu_asof_add_block(chgpu_asof * join, const chgpu_col * key_col, const chgpu_col * asof_col, const chgpu_col * null_map_u8, const chgpu_col * join_mask_u8,
                         uint64_t * block_index);
int chgpu_asof_total_rows(chgpu_asof * join, uint64_t * rows);
int chgpu_asof_probe(chgpu_asof * join, const chgpu_col * key_col, const chgpu_col * asof_col, const chgpu_col * null_map_u8, chgpu_col ** filter_u8,
                     chgpu_col ** right_rowid_u64, uint64_t * n_out);
int chgpu_asof_free(chgpu_asof * join);

/* ================================================================================================
 * a22 data movement  —  IColumn::index / replicate (src/Columns/ColumnVector.cpp:1121-1143, 879-907)
 * ============================================================================================== */
/* out[i] = col[indexes[i]], i < limit (limit 0 = all); indexes: CHGPU_U64 or CHGPU_U32.
   default_for_missing != 0: index == all-ones (-1) yields the type default 0 (join LEFT misses, insertDefault) */
int chgpu_index(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * indexes, uint64_t limit,
                int default_for_missing, chgpu_col ** out);
/* offsets: CHGPU_U64 cumulative (IColumn::Offsets) */
int chgpu_replicate(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * offsets_u64, chgpu_col ** out);
/* every column of a Block replicated by one offsets_to_replicate (joinBlock's loop over the left columns,
   src/Interpreters/HashJoin/HashJoinMethodsImpl.h:186-194): one host synchronisation for all of them, columns of one element width share
   a kernel (the offsets are read once).  outs[n_cols] receives the new columns. */
int chgpu_replicate_columns(chgpu_ctx * ctx, uint32_t n_cols, const chgpu_col * const * cols, const chgpu_col * offsets_u64, chgpu_col ** outs);

/* §8(f) rank 4 — ordered output: IColumn::getPermutation(direction, Stable, limit = 0, nan_direction_hint, res) for
   ColumnVector<T> (src/Columns/ColumnVector.cpp:245-330), the step under sortBlock / MergeSortingTransform.  LSD radix sort on
   the device with the stable semantics of less_stable / greater_stable (:120-166): equal values (-0.0 == 0.0, NaN with NaN) keep
   their original order in both directions; NaN is greater than every number when nan_direction_hint > 0, smaller otherwise.
   perm_in_u64 (may be NULL): sort the rows col[perm_in[i]] instead and return the composed permutation -- ORDER BY a, b is
   sort by b, then by a with b's permutation as perm_in.  Apply the result with chgpu_index (IColumn::permute), cut it for LIMIT. */
int chgpu_sort_permutation(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * perm_in_u64, int descending,
                           int nan_direction_hint, chgpu_col ** perm_out_u64);

/* getPermutation with a limit (ColumnVector.cpp:254-281: ORDER BY ... LIMIT n): the first `limit` entries of the stable sorted
   permutation.  A threshold from a sorted sample names the candidate rows (one comparison pass + filterToIndices); only they are
   sorted.  Exact; falls back to the full sort when the sample misleads, a value repeats massively or NaNs must come first. */
int chgpu_sort_permutation_limit(chgpu_ctx * ctx, const chgpu_col * col, int descending, int nan_direction_hint, uint64_t limit,
                                 chgpu_col ** perm_out_u64);
/* filterToIndices (src/Columns/ColumnsCommon.cpp:384-470): row numbers whose filter byte is non-zero, ascending */
int chgpu_filter_to_indices(chgpu_ctx * ctx, const chgpu_col * filter_u8, chgpu_col ** indexes_u64, uint64_t * rows_out);

/* ================================================================================================
 * a11/a21 hashing & sharding  —  ColumnVector::getWeakHash32 (ColumnVector.cpp:78-95), ConcurrentHashJoin
 * hashToSelector (src/Interpreters/ConcurrentHashJoin.cpp:426-440), IColumn::scatter (src/Columns/IColumn.cpp:245-269).
 * Bit-exact CRC32-C (Hash.h:63-66): shard ids are externally visible.
 * ============================================================================================== */
/* hash_u32[i] = (u32) hashCRC32(col[i], hash_u32[i])  (in place; initialise to 0xFFFFFFFF like WeakHash32) */
int chgpu_weak_hash32(chgpu_ctx * ctx, const chgpu_col * col, chgpu_col * hash_u32);
/* selector[i] = ((crc32c(key) >> 24) & 0xFF) & (num_shards-1); num_shards power of two <= 256; out: CHGPU_U32 */
int chgpu_hash_to_selector(chgpu_ctx * ctx, const chgpu_col * keys, uint32_t num_shards, chgpu_col ** selector_u32);
/* stable split of col by selector into num_columns new columns (outs[num_columns]) */
int chgpu_scatter(chgpu_ctx * ctx, const chgpu_col * col, const chgpu_col * selector_u32, uint32_t num_columns,
                  chgpu_col ** outs);
/* one-call hash partition of n_cols columns by keys (cols[0] must be the key column or any payload): computes the
   selector once, returns per-shard row counts (host, counts[num_shards]) and for each input column ONE output column
   holding the shards back to back (shard s occupies [sum(counts[:s]), +counts[s])) — the send buffer of an
   all-to-all.  Stable within a shard. */
int chgpu_partition_by_hash(chgpu_ctx * ctx, const chgpu_col * keys, uint32_t num_shards, uint32_t n_cols,
                            const chgpu_col * const * cols, chgpu_col ** outs, uint64_t * counts);

/* ================================================================================================
 * a12-a17 GROUP BY  —  Aggregator::executeOnBlock / mergeOnBlock / convertToBlocks
 * (src/Interpreters/Aggregator.h:179-190,227,253; Aggregator.cpp:1506-1627, 2468-2725, 2037-2117) with
 * AggregatedDataVariants key32/key64 = HashMap<UInt64, AggregateDataPtr> (AggregatedData.h:38).
 * One chgpu_agg == one AggregatedDataVariants (single writer).  key_type < 0: without_key.
 * Device table: open addressing, linear probing, power-of-two capacity, max fill 1/2, growth x4 until 2^23 then x2,
 * empty <=> key == 0 with the zero key kept out of line — the geometry of HashTable.h:217-330,358-391.
 * ============================================================================================== */
/* a15 multi-column fixed-width keys: packFixed<UInt64> (src/Interpreters/AggregationCommon.h:91-158) — the keys16/32/64
   variants of chooseAggregationMethod (Aggregator.cpp:773-778).  Columns are laid out consecutively (little endian) in one
   UInt64 per row; more than 8 key bytes -> CHGPU_ERR_NOT_IMPLEMENTED here: use a chgpu_keydict (keys128 / keys256, below).  unpack is the
   inverse used when the key column(s) of the result block are produced (insertKeyIntoColumns). */
int chgpu_pack_fixed_keys(chgpu_ctx * ctx, uint32_t n_cols, const chgpu_col * const * cols, chgpu_col ** packed_u64);
int chgpu_unpack_fixed_key(chgpu_ctx * ctx, const chgpu_col * packed_u64, uint32_t byte_offset, int type, chgpu_col ** out);
/* FixedString(N) keys (ColumnFixedString: rows x N raw bytes, src/Columns/ColumnFixedString.h; AggregatedDataVariants::key_fixed_string,
   AggregatedDataVariants.h:65-66,91-92; HashJoin key_fixed_string).  A value is N bytes, padding zeros included: a fixed-width key.  Word w =
   bytes [8w, 8w + 8) of every value as one UInt64 (little endian, zero padded past N): N <= 8 -> the ordinary UInt64 key (key64), N <= 32 ->
   keys128 / keys256 (chgpu_keydict_*).  chgpu_fixed_string_from_words is insertKeyIntoColumns: words -> chars.  N > 32 -> NOT_IMPLEMENTED. */
int chgpu_fixed_string_word(chgpu_ctx * ctx, const chgpu_col * chars_u8, uint32_t n, uint32_t word_index, chgpu_col ** out_u64);
int chgpu_fixed_string_from_words(chgpu_ctx * ctx, uint32_t n_words, const chgpu_col * const * words_u64, uint32_t n, chgpu_col ** chars_u8);
/* a15 keys128 / keys256: several fixed-width key columns that pack into more than 8 bytes (AggregatedDataVariants.h:70-71,83-84:
   HashMap<UInt128 / UInt256, ...>; HashMethodKeysFixed, src/Common/ColumnsHashing/HashMethod.h:236-410; packFixed,
   AggregationCommon.h:91-158; the same maps under HashJoin, HashJoin.h:267-358).  A chgpu_keydict is a device-resident, exact dictionary
   packed key -> dense UInt32 id (numbered as keys are first claimed; stable for the dictionary's lifetime, across blocks and growth).
   chgpu_keydict_encode packs the columns of rows [row_begin, row_end) (packFixed order: consecutively, little endian, zero padded to
   key_bytes = 16 or 32) and returns their ids: insert != 0 is emplaceKey (GROUP BY, join build), insert == 0 is findKey (join probe):
   a key the dictionary does not hold gets 0xFFFFFFFF.  The ids are an ordinary UInt32 key column for chgpu_agg_* / chgpu_join_* /
   chgpu_partition_*; chgpu_keydict_key_column gives back one original key column for a column of ids (insertKeyIntoColumns; bytes
   [byte_offset, +sizeof(type)) of the packed key; 0xFFFFFFFF -> 0); chgpu_keydict_selector the shard of every id by the reference's own
   hash of the packed key (UInt128HashCRC32 / UInt256HashCRC32, Hash.h:346-355,412-423 -> two-level bucket & (num_shards - 1)). */
typedef struct chgpu_keydict chgpu_keydict;
int chgpu_keydict_create(chgpu_ctx * ctx, uint32_t key_bytes, uint64_t size_hint, chgpu_keydict ** out);
int chgpu_keydict_encode(chgpu_keydict * dict, uint32_t n_cols, const chgpu_col * const * cols, uint64_t row_begin, uint64_t row_end, int insert,
                         chgpu_col ** ids_u32);
int chgpu_keydict_size(chgpu_keydict * dict, uint64_t * n_keys);
int chgpu_keydict_key_column(chgpu_keydict * dict, const chgpu_col * ids_u32, uint32_t byte_offset, int type, chgpu_col ** out);
int chgpu_keydict_selector(chgpu_keydict * dict, const chgpu_col * ids_u32, uint32_t num_shards, chgpu_col ** selector_u32);
int chgpu_keydict_free(chgpu_keydict * dict);
/* §8(f) rank 2 — LowCardinality keys (src/Columns/ColumnLowCardinality.h:27-69; low_cardinality_key* variants,
   AggregatedDataVariants.h:119-127; HashMethodSingleLowCardinalityColumn's per-position cache, ColumnsHashing.h:82-260).
   Every Block brings its own dictionary; the host resolves it against the query-wide dictionary into remap_u32[local position]
   = global id, and the rows are translated on the device: out_u32[i] = remap_u32[indexes[i]] (indexes: UInt8/16/32/64).
   The result is an ordinary UInt32 key column for chgpu_agg_* / chgpu_join_* / chgpu_hash_to_selector. */
int chgpu_lc_remap(chgpu_ctx * ctx, const chgpu_col * indexes, const chgpu_col * remap_u32, chgpu_col ** out_u32);
/* §8(f) rank 2 — String keys.  ColumnString (src/Columns/ColumnString.h:40-49) = chars_u8 (every value followed by a zero byte) +
   offsets_u64 (offsets[i] = end of value i including that zero).  Every row gets the dense id of its value, ids numbered by
   first appearance (ColumnUnique::uniqueInsertRangeFrom, src/Columns/ColumnUnique.h:520-620); first_rows_u64[id] = the row where
   the value first appears (the caller reads the dictionary's strings from its own Block there).  Exact: values are compared
   byte by byte; a 64-bit tag shared by two different values answers CHGPU_ERR_NOT_IMPLEMENTED (CPU path).  ids + dictionary
   are a ColumnLowCardinality: chgpu_lc_remap / GROUP BY / join as above.  The kernels read values 8 bytes at a time: a wrapped chars
   buffer (chgpu_col_wrap) must keep 8 readable bytes after its end, as PaddedPODArray's right pad does (columns made by this library do). */
int chgpu_string_dictionary_encode(chgpu_ctx * ctx, const chgpu_col * offsets_u64, const chgpu_col * chars_u8, chgpu_col ** ids_u32,
                                   chgpu_col ** first_rows_u64, uint64_t * n_distinct);
/* ColumnString::filter (src/Columns/ColumnString.cpp:270-290 -> filterArraysImpl, src/Columns/ColumnsCommon.cpp:191-286): the values
   whose filter byte is non-zero, in order, as a new ColumnString (offsets rebuilt, bytes moved together). */
int chgpu_string_filter(chgpu_ctx * ctx, const chgpu_col * offsets_u64, const chgpu_col * chars_u8, const chgpu_col * filter_u8,
                        chgpu_col ** out_offsets_u64, chgpu_col ** out_chars_u8, uint64_t * rows_out);
int chgpu_agg_create(chgpu_ctx * ctx, int key_type, uint32_t n_aggs, const int * agg_kinds, const int * arg_types,
                     uint64_t size_hint, chgpu_agg ** out);
/* executeOnBlock over rows [row_begin,row_end) of the key column and the argument columns (arg_cols[j] may be NULL
   for count()) */
int chgpu_agg_add_block(chgpu_agg * agg, const chgpu_col * key_col, const chgpu_col * const * arg_cols,
                        uint64_t row_begin, uint64_t row_end);
/* The same over the rows of [row_begin,row_end) whose filter byte is non-zero -- a FilterTransform (FilterTransform.cpp:136-256)
   directly in front of the AggregatingTransform, fused: rows that fail the WHERE clause neither create groups nor update
   states.  Low-cardinality aggregations read the mask inside the aggregation kernel (no filtered copy of the columns is made);
   the other strategies filter the block's columns first (chgpu_filter_columns).  filter_u8 == NULL: plain add_block. */
int chgpu_agg_add_block_filtered(chgpu_agg * agg, const chgpu_col * key_col, const chgpu_col * const * arg_cols,
                                 uint64_t row_begin, uint64_t row_end, const chgpu_col * filter_u8);
/* mergeDataImpl: fold src's states into dst (src stays valid but should be freed) */
int chgpu_agg_merge(chgpu_agg * dst, const chgpu_agg * src);
/* merge partial states that arrive as columns (the ColumnAggregateFunction blocks of a distributed GROUP BY,
   Aggregator::mergeOnBlock :227): keys + one state column per aggregate (avg: two columns numerator, denominator
   passed consecutively in state_cols) */
int chgpu_agg_merge_states(chgpu_agg * dst, const chgpu_col * key_col, const chgpu_col * const * state_cols, uint64_t rows);
int chgpu_agg_size(chgpu_agg * agg, uint64_t * groups);
/* convertToBlockImplFinal: key column + one result column per aggregate (count -> U64, sum -> SumSimple type,
   avg -> F64).  Row order is table order (unspecified, as in the reference); keys_out may be NULL for without_key. */
int chgpu_agg_finalize(chgpu_agg * agg, chgpu_col ** keys_out, chgpu_col ** res_cols, uint64_t * groups);
/* convertToBlockImplNotFinal: raw states (sum/count 8 B; avg -> numerator, denominator as two columns) */
int chgpu_agg_export_states(chgpu_agg * agg, chgpu_col ** keys_out, chgpu_col ** state_cols, uint64_t * groups);
/* convertToBlockImplNotFinal for a two-level result (Aggregator::prepareBlocksAndFillTwoLevelImpl, Aggregator.cpp:2790-2860): the
   same states ordered by bucket = (crc32c(key) >> 24) & 0xFF (TwoLevelHashTable.h:53 getBucketFromHash over HashCRC32<Key>,
   Hash.h:280-288), bucket_counts[256] rows each -- bucket b is rows [sum(counts[:b]), +counts[b]): the 256 blocks with
   BlockInfo::bucket_num (src/Core/BlockInfo.h:21-29) a CPU initiator's MergingAggregatedMemoryEfficientTransform expects. */
int chgpu_agg_export_states_two_level(chgpu_agg * agg, chgpu_col ** keys_out, chgpu_col ** state_cols, uint64_t * groups,
                                      uint64_t * bucket_counts);
/* §8(f) rank 4 — partial states on the wire: the bytes IAggregateFunction::serialize writes for every row of a ColumnAggregateFunction
   (SerializationAggregateFunction: the rows' states one after the other, no lengths): sum = the 8-byte sum, little endian
   (AggregateFunctionSum.h:288-291); count = VarUInt (AggregateFunctionCount.h:126-129); avg = the 8-byte numerator + VarUInt denominator
   (AggregateFunctionAvg.h, AvgFraction).  word0 / word1 are the state columns chgpu_agg_export_states(_two_level) returns (word1: avg's
   denominator).  offsets_u64 (may be NULL) receives rows + 1 byte offsets, so the 256 bucket blocks of a two-level export can be cut
   out of one buffer.  deserialize is the inverse for mergeOnBlock: `n_streams` independent runs of states (stream s: stream_rows[s] states
   from byte stream_byte_begin[s]; a single stream may pass NULL) -> word columns for chgpu_agg_merge_states.  A state's length is only
   known after its VarUInt, so one lane walks one stream; bucket-wise exchanges give 256 streams per peer. */
int chgpu_agg_serialize_states(chgpu_ctx * ctx, int kind, const chgpu_col * word0, const chgpu_col * word1, chgpu_col ** bytes_u8, chgpu_col ** offsets_u64);
int chgpu_agg_deserialize_states(chgpu_ctx * ctx, int kind, const chgpu_col * bytes_u8, uint32_t n_streams, const uint64_t * stream_byte_begin,
                                 const uint64_t * stream_rows, chgpu_col ** word0, chgpu_col ** word1);
int chgpu_agg_free(chgpu_agg * agg);

/* ================================================================================================
 * a19/a20 hash join  —  IJoin::addBlockToJoin / onBuildPhaseFinish / joinBlock (src/Interpreters/IJoin.h:80-93,142)
 * for HashJoin key64 (one numeric key up to 8 bytes; src/Interpreters/HashJoin/HashJoin.cpp:254-356,556-768,1036-1101;
 * HashJoinMethodsImpl.h:220-281,402-549).  RowRef{block*,row} becomes a global build row id
 * (block_index << 32 | row_in_block); RowRefList's linked 7-slot batches become one CSR array.
 * ============================================================================================== */
int chgpu_join_create(chgpu_ctx * ctx, int key_type, int kind, int strictness, int any_take_last_row,
                      uint64_t size_hint, chgpu_join ** out);
/* null_map_u8 / join_mask_u8 may be NULL (rows with null key or failed ON mask are not inserted, Impl.h:261-272).
   block_index_out receives the index the block got (0,1,2...).  rows >= 2^32 -> CHGPU_ERR_TOO_MANY_ROWS. */
int chgpu_join_add_block(chgpu_join * j, const chgpu_col * key_col, const chgpu_col * null_map_u8,
                         const chgpu_col * join_mask_u8, uint32_t * block_index_out);
/* IJoin::onBuildPhaseFinish: no more right blocks.  The hash table is built lazily, by the first call that needs it (chgpu_join_probe, the key
   count of chgpu_join_total_rows, chgpu_join_non_joined_rows); chgpu_join_probe_agg over a large unique-key build side joins without one. */
int chgpu_join_finish_build(chgpu_join * j);
int chgpu_join_total_rows(chgpu_join * j, uint64_t * rows, uint64_t * keys);
/* joinRightColumns over the left key column.  Outputs (all device columns, caller frees; unused ones are NULL):
     filter_u8        [n_left_consumed]  need_filter variants (INNER ANY, LEFT SEMI, LEFT ANTI)
     offsets_u64      [n_left_consumed]  need_replication variants (ALL): cumulative offsets_to_replicate
     right_rowid_u64  [n_out]            one entry per appended right row: (block<<32|row), all-ones = default row;
                                         may be NULL for LEFT SEMI / LEFT ANTI when the right side contributes no columns
                                         (an empty AddedColumns, AddedColumns.h): only the filter and n_out are produced
   max_joined_block_rows: 0 = unlimited; else processing stops BEFORE the first left row at which the running
   output count is already >= max (HashJoinMethodsImpl.h:436-444) and n_left_consumed < rows tells the caller to
   resubmit the tail (JoiningTransform.cpp:220-260). */
int chgpu_join_probe(chgpu_join * j, const chgpu_col * key_col, const chgpu_col * null_map_u8,
                     uint64_t max_joined_block_rows, chgpu_col ** filter_u8, chgpu_col ** offsets_u64,
                     chgpu_col ** right_rowid_u64, uint64_t * n_out, uint64_t * n_left_consumed);
/* joinBlock with the aggregation that consumes its output fused behind it: `SELECT count(), sum(right.payload) FROM left JOIN right`
   (HashJoinMethodsImpl.h:402-549 -> AddedColumns' lazy gather, AddedColumns.cpp:39-131 -> Aggregator::executeWithoutKeyImpl,
   Aggregator.cpp:1276-1321).  One pass over the left keys; nothing per left row is written (no offsets_to_replicate, no row ids, no
   gathered column).  *count_out = rows the join would emit; sum_out = 8 bytes typed like SumSimple(payload type): the sum of the
   payload over those rows (default rows of LEFT joins add 0; LEFT ANTI emits no right row).  right_payload is the right Blocks'
   column glued in insertion order (chgpu_col_concat), may be NULL for count only.  Variants: INNER ALL, LEFT ALL, LEFT ANY,
   LEFT SEMI, LEFT ANTI; INNER ANY / RIGHT / FULL -> CHGPU_ERR_NOT_IMPLEMENTED (stateful across calls: use chgpu_join_probe).
   Float sums are reduced in a fixed order: run-to-run reproducible. */
int chgpu_join_probe_agg(chgpu_join * j, const chgpu_col * key_col, const chgpu_col * null_map_u8, const chgpu_col * right_payload,
                         uint64_t * count_out, void * sum_out);
/* A CHAIN of JoiningTransforms answered in one sweep over the left key columns (late materialisation).  When every join of the chain is of
   the filter form -- LEFT SEMI, LEFT ANTI, or ALL over a build side without duplicate keys: HashJoinMethodsImpl.h:68-202 then computes a
   filter and calls `block.filter(filter)` (:122-123) on EVERY left column, once per join -- the left rows that survive the whole chain are
   the AND of the joins' filters and no join changes a row's multiplicity.  This entry computes that AND before any left column is copied
   (key_cols[s] is probed against joins[s]; null_maps / null_maps[s] may be NULL): dense dimension key sets are staged in LDS, the other
   steps are probed only for the rows still alive.  Outputs, all sized by the survivors (*n_kept) and in ascending left-row order; every
   output pointer may be NULL when not wanted:
     indexes_u64          the surviving left row numbers (filterToIndices, src/Columns/FilterDescription.cpp:113-116)
     right_rowid_u64[s]   for steps with want_right_rows[s] != 0: the matched build row (block << 32 | row) of joins[s], all-ones where the
                          join adds a default row (LEFT joins without a match, ANTI) -- what chgpu_join_probe would return over the survivors
     carry_out[c]         carry_cols[c] gathered at the survivors (IColumn::index; the chain's `block.filter` over the left columns)
     filter_u8 [rows]     the chain's IColumn::Filter itself, 1 = the row survives every join
   LEFT ANY / LEFT ALL steps keep every left row (they only contribute right_rowid).  INNER ANY, RIGHT, FULL (stateful across calls) and ALL
   over duplicate build keys (replicates rows) -> CHGPU_ERR_NOT_IMPLEMENTED: run those joins one by one with chgpu_join_probe. */
int chgpu_join_probe_chain(uint32_t n_steps, chgpu_join * const * joins, const chgpu_col * const * key_cols, const chgpu_col * const * null_maps,
                           const int * want_right_rows, uint32_t n_carry, const chgpu_col * const * carry_cols, chgpu_col ** indexes_u64,
                           chgpu_col ** right_rowid_u64, chgpu_col ** carry_out, chgpu_col ** filter_u8, uint64_t * n_kept);
/* The same with ONE right column per payload step gathered inside the call (AddedColumns' lazy columns, AddedColumns.cpp:39-131): right_cols[s]
   != NULL (a step with want_right_rows[s], over a one-block build side) makes right_out[s] that column's values at the matched rows -- a miss
   gives the type's default -- instead of the row ids; steps with right_cols[s] == NULL still return row ids in right_out[s]. */
int chgpu_join_probe_chain_columns(uint32_t n_steps, chgpu_join * const * joins, const chgpu_col * const * key_cols, const chgpu_col * const * null_maps,
                                   const int * want_right_rows, const chgpu_col * const * right_cols, uint32_t n_carry, const chgpu_col * const * carry_cols,
                                   chgpu_col ** indexes_u64, chgpu_col ** right_out, chgpu_col ** carry_out, chgpu_col ** filter_u8, uint64_t * n_kept);
/* (block_index << 32 | row) ids -> running ordinal of the row over all right blocks in insertion order (all-ones stays
   all-ones): the index into payload columns concatenated with chgpu_col_concat, i.e. fillFromBlocksAndRowNumbers
   (src/Columns/IColumn.cpp:515-526) for many right Blocks */
int chgpu_join_flatten_rowids(chgpu_join * j, const chgpu_col * right_rowid_u64, chgpu_col ** flat_u64);
/* IJoin::getNonJoinedBlocks (src/Interpreters/IJoin.h:133-134; NotJoinedHash, HashJoin.cpp:1280-1420) for RIGHT / FULL joins (strictness
   ALL): every right row a joinBlock emitted is flagged (JoinUsedFlags); after the last joinBlock this returns the build rows no left row
   matched -- rows with a NULL key or a zero ON mask included -- as (block << 32 | row) ids in insertion order.  RIGHT probes like INNER,
   FULL like LEFT. */
int chgpu_join_non_joined_rows(chgpu_join * j, chgpu_col ** right_rowid_u64, uint64_t * rows_out);
int chgpu_join_free(chgpu_join * j);

/* ================================================================================================
 * (e) multi-GPU  —  the exchange step of the sharded operators over RCCL (xGMI inside a node); one process per GPU.
 * ConcurrentHashJoin::dispatchBlock hands the sub-blocks of a scattered Block to the per-slot HashJoins
 * (src/Interpreters/ConcurrentHashJoin.cpp:538-565), a parallel merge hands two-level buckets to their owners
 * (ConcurrentHashJoin.cpp:600-653; src/Processors/Transforms/AggregatingTransform.cpp:120-136): threads of one address space there,
 * ranks exchanging hash partitions here.  Shard of a key = ((crc32c(key) >> 24) & 0xFF) & (world - 1) (chgpu_hash_to_selector),
 * so world is a power of two <= 256 (ConcurrentHashJoin.cpp:158).  A communicator binds one rank to one chgpu_ctx; all its
 * traffic runs on that context's stream.  librccl is loaded on first use; without it these calls fail with CHGPU_ERR_DEVICE.
 * Sequence: rank 0 calls chgpu_comm_unique_id and hands the 128 bytes to every rank out of band (the host pipeline's own control
 * plane), every rank calls chgpu_comm_init (collective).  Then per exchange: chgpu_partition_by_hash -> chgpu_all_to_all_counts
 * -> chgpu_all_to_all per column.
 * ============================================================================================== */
#define CHGPU_UNIQUE_ID_BYTES 128
typedef struct chgpu_comm chgpu_comm;
int chgpu_comm_unique_id(uint8_t id_out[CHGPU_UNIQUE_ID_BYTES]);
int chgpu_comm_init(chgpu_ctx * ctx, int rank, int world, const uint8_t unique_id[CHGPU_UNIQUE_ID_BYTES], chgpu_comm ** out);
int chgpu_comm_destroy(chgpu_comm * comm);
int chgpu_comm_rank(const chgpu_comm * comm);
int chgpu_comm_world(const chgpu_comm * comm);
/* [0] bytes sent to other ranks [1] bytes received from other ranks [2] collectives issued */
int chgpu_comm_stats(const chgpu_comm * comm, uint64_t out[3]);
/* recv_counts[p] = the send_counts[my rank] of rank p: sizes the receive side of the all-to-all that follows (host arrays of `world`) */
int chgpu_all_to_all_counts(chgpu_comm * comm, const uint64_t * send_counts, uint64_t * recv_counts);
/* send: shards back to back as chgpu_partition_by_hash returns them (shard p = send_counts[p] rows for rank p); *recv_out: a new column of
   sum(recv_counts) rows, what ranks 0..world-1 sent here in rank order.  One grouped send/recv: every xGMI link carries its peer's
   partition at once; the own partition is a device copy.  Asynchronous on the context's stream. */
int chgpu_all_to_all(chgpu_comm * comm, const chgpu_col * send, const uint64_t * send_counts, const uint64_t * recv_counts, chgpu_col ** recv_out);
/* The exchange of a whole Block: n_cols columns that share their partition boundaries (chgpu_partition_by_hash's outputs: the key column
   and every state / payload column) leave in ONE grouped send / recv over all columns and peers, after ONE exchange of the row counts
   (recv_counts [world] is an output: the rows each rank sent here).  What dispatchBlock + the shard's input port do for one Block
   (src/Interpreters/ConcurrentHashJoin.cpp:538-565).  chgpu_all_to_all_counts + chgpu_all_to_all remain for a single column. */
int chgpu_all_to_all_multi(chgpu_comm * comm, uint32_t n_cols, const chgpu_col * const * send, const uint64_t * send_counts, uint64_t * recv_counts,
                           chgpu_col ** recv_out);
/* mergeWithoutKeyDataImpl across ranks (src/Interpreters/Aggregator.cpp:2584-2628): element-wise wrap-around sum of a UInt64 / Int64
   device column over all ranks, in place (asynchronous); _host: the same for <= 64 host values (synchronises) */
int chgpu_all_reduce_u64(chgpu_comm * comm, chgpu_col * inout_u64);
int chgpu_all_reduce_u64_host(chgpu_comm * comm, uint64_t * values, uint32_t n);
/* returns once every rank's queued work on its stream has finished */
int chgpu_comm_barrier(chgpu_comm * comm);

#ifdef __cplusplus
}
#endif
#endif /* CHGPU_H */
