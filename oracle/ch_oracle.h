/*
 * ch_oracle.h — CPU restatement ("oracle") of the ClickHouse block-processing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product path (clickhouse_amd/, libchgpu.so)
 * never links, imports or calls anything in oracle/.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout, filimonov/ClickHouse @ 2025-07-11).  Parity pinning:
 *   - hashes: checked against the reference's own Hash.h compiled into oracle/_ref/
 *     (oracle/ref_hash_wrapper.cpp) and against the KATs in tests/golden/hash_kat.json;
 *   - filter: the gtest property of src/Columns/tests/gtest_column_vector.cpp:41-104;
 *   - hash table: scenarios of src/Common/tests/gtest_hash_table.cpp:50-140;
 *   - join / group-by semantics: the .reference outputs of tests/queries/0_stateless
 *     00042-00055, 00120 (tests/golden/ JSON fixtures);
 *   - Float64 sum association order is pinned by no reference test (tolerance 1e-6 rel).
 */
#ifndef CH_ORACLE_H
#define CH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* type tags shared with include/chgpu.h (same numeric values) */
enum { CHO_I64 = 0, CHO_U32 = 1, CHO_U64 = 2, CHO_F64 = 3, CHO_U8 = 4, CHO_I32 = 5, CHO_U16 = 6, CHO_I16 = 7, CHO_I8 = 8, CHO_F32 = 9 };
/* comparison ops (FunctionsComparison.h: EqualsOp..GreaterOrEqualsOp) */
enum { CHO_EQ = 0, CHO_NE = 1, CHO_LT = 2, CHO_GT = 3, CHO_LE = 4, CHO_GE = 5 };
/* aggregate kinds */
enum { CHO_AGG_COUNT = 0, CHO_AGG_SUM = 1, CHO_AGG_AVG = 2, CHO_AGG_MIN = 3, CHO_AGG_MAX = 4, CHO_AGG_ANY = 5 };
/* join kind / strictness (src/Core/Joins.h) */
enum { CHO_JOIN_INNER = 0, CHO_JOIN_LEFT = 1 };
enum { CHO_STRICT_ANY = 0, CHO_STRICT_ALL = 1, CHO_STRICT_SEMI = 2, CHO_STRICT_ANTI = 3 };

#define CHO_DEFAULT_BLOCK_SIZE 65409 /* src/Core/Defines.h:31-32 */

/* ---- a11 hashes: src/Common/HashTable/Hash.h ---- */
uint64_t cho_intHash64(uint64_t x);                      /* Hash.h:27-36 */
uint64_t cho_intHashCRC32(uint64_t x);                   /* Hash.h:63-78, seed -1 */
uint64_t cho_intHashCRC32_seed(uint64_t x, uint64_t updated_value); /* Hash.h:79-93 */
uint64_t cho_intHashCRC32_soft(uint64_t x, uint64_t updated_value); /* bitwise CRC32-C, no SSE4.2 */
uint32_t cho_intHash32(uint64_t key, uint64_t salt);     /* Hash.h:498-511 */
uint64_t cho_sql_intHash64(uint64_t x);                  /* FunctionsHashing.h:184-192 */
uint32_t cho_sql_intHash32(uint64_t x);                  /* FunctionsHashing.h:173-182 */
/* hashCRC32<T>(key, seed): zero-extends <=8-byte keys to 64 bit (Hash.h:276-288) */
void cho_hash_crc32_batch(int type, const void * keys, size_t n, uint64_t * out);
/* ColumnVector<T>::getWeakHash32 (ColumnVector.cpp:78-95): hash[i] = (u32)hashCRC32(data[i], hash[i]) */
void cho_weak_hash32(int type, const void * data, size_t n, uint32_t * hash_inout);
/* TwoLevelHashTable::getBucketFromHash (TwoLevelHashTable.h:53) */
uint32_t cho_two_level_bucket(uint64_t hash_value);
/* crc32c slice-by-8 tables T[j][b] such that crc32c(-1, x) = XOR_j T[j][byte_j(x)] ^ T_const; for GPU LUT parity */
void cho_crc32c_tables(uint32_t tables[8][256], uint32_t * constant);

/* ---- a3 comparison: FunctionsComparison.h:165-259 + AccurateComparison.h:20-130 ---- */
/* c[i] = Op(a[i], scalar) ? 1 : 0 ; a_type/scalar_type are CHO_* tags; scalar passed by pointer */
int cho_cmp_const(int a_type, const void * a, size_t n, int op, int scalar_type, const void * scalar, uint8_t * c);

/* ---- a4/a5 filter ---- */
uint64_t cho_bytes64MaskToBits64Mask(const uint8_t * bytes64);            /* ColumnsCommon.h:27-72 */
size_t cho_countBytesInFilter(const uint8_t * filt, size_t start, size_t end); /* ColumnsCommon.cpp:31-58 */
/* ColumnVector<T>::filter (ColumnVector.cpp:682-724), Default doFilterAligned (:559-594).
   returns number of rows written to out, or -1 when filt_n != n (SIZES_OF_COLUMNS_DOESNT_MATCH). */
int64_t cho_filter(int elem_size, const void * data, size_t n, const uint8_t * filt, size_t filt_n, void * out);
/* a6 FilterDescription with Nullable(UInt8): res[i] = data[i] && !null[i] (FilterDescription.cpp:86-92) */
void cho_filter_description_nullable(const uint8_t * data, const uint8_t * null_map, size_t n, uint8_t * res);

/* ---- a22 index / replicate / scatter ---- */
void cho_index(int elem_size, const void * data, const uint64_t * indexes, size_t limit, void * out); /* ColumnVector.cpp:1121-1143 */
void cho_replicate(int elem_size, const void * data, size_t n, const uint64_t * offsets, void * out); /* ColumnVector.cpp:879-907 */
/* IColumn::scatter (IColumn.cpp:245-269): stable split by selector; out_concat holds the num_columns outputs
   back to back, out_sizes[k] their sizes. */
void cho_scatter(int elem_size, const void * data, size_t n, const uint64_t * selector, size_t num_columns,
                 void * out_concat, uint64_t * out_sizes);

/* ---- a8/a9 aggregate functions without key ---- */
/* AggregateFunctionSumData<T>::addMany (AggregateFunctionSum.h:62-103). state is 8 bytes (Int64/UInt64/Float64). */
void cho_sum_add_many(int type, void * state, const void * ptr, size_t start, size_t end);
/* addManyConditional (AggregateFunctionSum.h:138-236), add_if_zero=false */
void cho_sum_add_many_conditional(int type, void * state, const void * ptr, const uint8_t * cond, size_t start, size_t end);
/* AvgFraction::divide (AggregateFunctionAvg.h:61-67) */
double cho_avg_divide(int numerator_type, const void * numerator, uint64_t denominator);

/* The whole C1/C2 pipeline restated per Block: FilterTransform::doTransform (FilterTransform.cpp:136-256)
   -> Aggregator::executeWithoutKeyImpl (Aggregator.cpp:1276-1321) -> sum/count states; blocks of block_rows.
   pred column == value column when val == NULL.  threads>1 mirrors one AggregatingTransform per stream +
   mergeWithoutKeyDataImpl (Aggregator.cpp:2584-2628).  sum_out is 8 bytes typed like SumSimple result. */
int cho_filter_sum_pipeline(int type, const void * pred, const void * val, size_t n, int op, const void * scalar,
                            size_t block_rows, int threads, void * sum_out, uint64_t * count_out,
                            uint64_t * chunks_dropped, uint64_t * chunks_passthrough);

/* ---- §8(f) rank 1: expression DAG pieces for SSB Q1.1-style queries ---- */
/* value expressions: FunctionBinaryArithmetic with NumberTraits result types (src/DataTypes/NumberTraits.h:73-87):
   multiply/plus -> next size up, signed if either side is; minus -> next size up, always signed; 8-byte inputs stay 8 bytes
   and wrap (MultiplyImpl::apply, src/Functions/multiply.cpp:10-28: static_cast<Result>(a) * b). */
enum { CHO_VAL_COL = 0, CHO_VAL_MUL = 1, CHO_VAL_PLUS = 2, CHO_VAL_MINUS = 3 };
/* and(a, b) over UInt8 0/1 columns: AndImpl::apply = a & b (src/Functions/FunctionsLogical.h:82-96) */
void cho_and_u8(const uint8_t * a, const uint8_t * b, size_t n, uint8_t * out);
/* out type = cho_arith_result_type(op, a_type, b_type) (CHO_I64 or CHO_U64 here); integer inputs only */
int cho_arith_result_type(int value_op, int a_type, int b_type);
int cho_arith_sum_type(int value_op, int a_type, int b_type);
int cho_arith(int value_op, int a_type, const void * a, int b_type, const void * b, size_t n, void * out);
/* `SELECT sum(<value>), count() WHERE p1 AND p2 ...` through the per-Block pipeline: every predicate is a comparison of a
   column with a constant (a3), the masks are and-ed, FilterTransform filters the columns the projection needs, the value
   expression runs on the filtered Block (ExpressionActions order), sum/count states accumulate (a8/a9).
   cols[n_cols] typed cols_type[]; predicate k tests cols[pred_col[k]] pred_op[k] scalar (8 raw bytes, typed pred_stype[k]). */
int cho_expr_filter_sum_pipeline(size_t n_cols, const int * cols_type, const void * const * cols, size_t n,
                                 size_t n_preds, const uint32_t * pred_col, const int * pred_op, const int * pred_stype,
                                 const uint64_t * pred_scalar_bits, int value_op, uint32_t val_a, uint32_t val_b,
                                 size_t block_rows, int threads, void * sum_out, uint64_t * count_out);

/* ---- a12/a13 hash tables (exposed for the gtest_hash_table scenarios) ---- */
typedef struct cho_hashmap cho_hashmap; /* HashMap<UInt64, UInt64, HashCRC32<UInt64>> */
cho_hashmap * cho_hashmap_create(void);
void cho_hashmap_free(cho_hashmap *);
/* emplace: returns 1 if inserted; *mapped_out points at the mapped value */
int cho_hashmap_emplace(cho_hashmap *, uint64_t key, uint64_t ** mapped_out);
uint64_t * cho_hashmap_find(cho_hashmap *, uint64_t key);
void cho_hashmap_reserve(cho_hashmap *, size_t num_elements);
size_t cho_hashmap_size(const cho_hashmap *);
size_t cho_hashmap_buf_size(const cho_hashmap *);
int cho_hashmap_has_zero(const cho_hashmap *);
/* iteration order: zero key first then buffer order (HashTable.h:620-660); returns count */
size_t cho_hashmap_dump(const cho_hashmap *, uint64_t * keys, uint64_t * values);

/* ---- a14-a17 Aggregator (key32/key64 -> HashMap<UInt64, AggregateDataPtr, HashCRC32<UInt64>>) ---- */
typedef struct cho_agg cho_agg;
/* key_type: CHO_U32/CHO_U64/CHO_I64, or -1 for without_key.  group_by_two_level_threshold: 100000 default
   (Settings.cpp:957); 0 disables conversion. */
cho_agg * cho_agg_create(int key_type, int n_aggs, const int * kinds, const int * arg_types,
                         uint64_t group_by_two_level_threshold);
void cho_agg_free(cho_agg *);
/* Aggregator::executeOnBlock (Aggregator.cpp:1506-1627) over rows [row_begin,row_end) */
int cho_agg_execute_on_block(cho_agg *, const void * keys, const void * const * args, size_t row_begin, size_t row_end);
/* mergeDataImpl / mergeSingleLevelDataImpl / mergeBucketImpl (Aggregator.cpp:2468-2725); src is consumed */
int cho_agg_merge(cho_agg * dst, cho_agg * src);
size_t cho_agg_size(const cho_agg *);
int cho_agg_is_two_level(const cho_agg *);
/* convertToBlockImplFinal (Aggregator.cpp:2037-2117): keys in table iteration order (zero key first,
   two-level: bucket 0..255); results[j] typed: count->u64, sum->SumSimple(i64/u64/f64), avg->f64.
   keys_out typed like key_type. returns rows. */
size_t cho_agg_convert_to_block(const cho_agg *, void * keys_out, void * const * results_out);

/* ---- a19/a20 HashJoin (one 8-byte numeric key: key64, HashMap<UInt64, Mapped, HashCRC32<UInt64>>) ---- */
typedef struct cho_join cho_join;
cho_join * cho_join_create(int kind, int strictness, int any_take_last_row);
void cho_join_free(cho_join *);
/* HashJoin::addBlockToJoin (HashJoin.cpp:556-768) -> insertFromBlockImplTypeCase (HashJoinMethodsImpl.h:220-281).
   null_map/join_mask may be NULL. returns the block id (0,1,2..) */
int64_t cho_join_add_block(cho_join *, const uint64_t * keys, size_t rows, const uint8_t * null_map, const uint8_t * join_mask);
size_t cho_join_total_rows(const cho_join *);
size_t cho_join_keys(const cho_join *);
/* joinRightColumns (HashJoinMethodsImpl.h:402-549).  Outputs:
     filter[rows]            (need_filter variants; else untouched)
     offsets[rows]           (need_replication variants: cumulative; else untouched)
     added_block/added_row[] one entry per appended right row: (block id,row) or (-1,-1) for a default row
   returns number of left rows consumed (< rows when max_joined_block_rows hit), *n_added = entries written. */
size_t cho_join_probe(cho_join *, const uint64_t * keys, size_t rows, const uint8_t * null_map,
                      size_t max_joined_block_rows, uint8_t * filter, uint64_t * offsets,
                      int64_t * added_block, int64_t * added_row, size_t added_cap, size_t * n_added);
int cho_join_need_filter(const cho_join *);
int cho_join_need_replication(const cho_join *);

/* ---- a21 ConcurrentHashJoin sharding (ConcurrentHashJoin.cpp:426-440) ---- */
/* selector[i] = getBucketFromHash(hashCRC32(key)) & (num_shards-1) ; num_shards power of two <= 256 */
void cho_hash_to_selector(int type, const void * keys, size_t n, size_t num_shards, uint64_t * selector);

/* ---- a15 keys128 / keys256: packFixed (AggregationCommon.h:91-158) + HashMap<UInt128 / UInt256> with UInt128HashCRC32 / UInt256HashCRC32 ---- */
void cho_pack_fixed(size_t n_cols, const uint32_t * sizes, const void * const * cols, size_t n, size_t key_bytes, uint8_t * out);
uint64_t cho_hash_keys_fixed(const uint64_t * words, size_t n_words); /* Hash.h:346-355, 412-423 */
typedef struct cho_widemap cho_widemap;
cho_widemap * cho_widemap_create(size_t key_bytes); /* 16 or 32 */
void cho_widemap_free(cho_widemap *);
size_t cho_widemap_size(const cho_widemap *);
/* emplaceKey (insert != 0) / findKey: ids_out[i] = number of the key by first appearance, ~0 when absent */
void cho_widemap_batch(cho_widemap *, const uint8_t * packed, size_t n, int insert, uint64_t * ids_out);
void cho_widemap_keys(const cho_widemap *, uint8_t * out);

/* ---- CPU-baseline drivers (bench.py cpu_baseline leg; also the parity check on the sample): N pipeline streams over Blocks ---- */
/* GROUP BY: one Aggregator per stream with the reference's hash-cell prefetch (Aggregator.cpp:1025-1054), two-level conversion,
   bucket-parallel merge (AggregatingTransform.cpp:120-136).  Returns the merged aggregator; seconds_out[2] = {consume, merge}. */
cho_agg * cho_groupby_pipeline(int key_type, int n_aggs, const int * kinds, const int * arg_types, const void * keys, const void * const * args,
                               size_t n, size_t block_rows, int threads, uint64_t two_level_threshold, double * seconds_out);
/* SELECT count(), sum(bv) FROM probe INNER JOIN build ON pk = bk (ALL): build by one stream, probe by N; seconds_out[2] = {build, probe} */
int cho_join_count_sum_pipeline(const uint64_t * bk, const int64_t * bv, size_t nb, const uint64_t * pk, size_t np, size_t block_rows, int threads,
                                uint64_t * count_out, uint64_t * sum_out, double * seconds_out);

#ifdef __cplusplus
}
#endif
#endif
