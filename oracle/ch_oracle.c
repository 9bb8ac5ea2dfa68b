/*
 * ch_oracle.c — CPU restatement of the ClickHouse block-processing hot path (filter -> aggregate -> hash join).
 *
 * TEST INFRASTRUCTURE ONLY (see ch_oracle.h).  Written from scratch following the reference's algorithms;
 * each function cites the reference file:line it restates.  Never linked into the product library.
 */
#define _GNU_SOURCE
#include "ch_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#if defined(__SSE4_2__)
#include <nmmintrin.h>
#endif

#if defined(__GNUC__) && !defined(__clang__) && defined(__x86_64__)
#define CHO_MULTITARGET __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define CHO_MULTITARGET
#endif

/* ------------------------------------------------------------------------------------------------
 * a11  hashes  (src/Common/HashTable/Hash.h)
 * ---------------------------------------------------------------------------------------------- */

uint64_t cho_intHash64(uint64_t x) /* Hash.h:27-36 (murmur finalizer) */
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

/* Bitwise CRC32-C (Castagnoli, reflected 0x82F63B78), 8 message bytes little-endian, no final xor:
   the semantics of _mm_crc32_u64(seed, x) that Hash.h:63-93 relies on. */
uint64_t cho_intHashCRC32_soft(uint64_t x, uint64_t updated_value)
{
    uint32_t crc = (uint32_t)updated_value;
    for (int i = 0; i < 8; ++i)
    {
        crc ^= (uint32_t)((x >> (8 * i)) & 0xFF);
        for (int k = 0; k < 8; ++k)
            crc = (crc >> 1) ^ (0x82F63B78u & (0u - (crc & 1u)));
    }
    return crc;
}

uint64_t cho_intHashCRC32_seed(uint64_t x, uint64_t updated_value) /* Hash.h:79-93 */
{
#if defined(__SSE4_2__)
    return _mm_crc32_u64(updated_value, x);
#else
    return cho_intHashCRC32_soft(x, updated_value);
#endif
}

uint64_t cho_intHashCRC32(uint64_t x) /* Hash.h:63-78 */
{
    return cho_intHashCRC32_seed(x, (uint64_t)-1);
}

uint32_t cho_intHash32(uint64_t key, uint64_t salt) /* Hash.h:498-511 */
{
    key ^= salt;
    key = (~key) + (key << 18);
    key = key ^ ((key >> 31) | (key << 33));
    key = key * 21;
    key = key ^ ((key >> 11) | (key << 53));
    key = key + (key << 6);
    key = key ^ ((key >> 22) | (key << 42));
    return (uint32_t)key;
}

uint64_t cho_sql_intHash64(uint64_t x) { return cho_intHash64(x ^ 0x4CF2D2BAAE6DA887ULL); }   /* FunctionsHashing.h:184-192 */
uint32_t cho_sql_intHash32(uint64_t x) { return cho_intHash32(x, 0x75D9543DE018BF45ULL); }    /* FunctionsHashing.h:173-182 */

static size_t type_size(int type)
{
    switch (type)
    {
        case CHO_I64: case CHO_U64: case CHO_F64: return 8;
        case CHO_U32: case CHO_I32: case CHO_F32: return 4;
        case CHO_U16: case CHO_I16: return 2;
        case CHO_U8: case CHO_I8: return 1;
        default: return 0;
    }
}

/* hashCRC32<T> (Hash.h:276-288): memcpy the key into a zeroed UInt64 (zero extension, little endian) */
static inline uint64_t load_key_zext(int type, const void * keys, size_t i)
{
    uint64_t out = 0;
    size_t sz = type_size(type);
    memcpy(&out, (const char *)keys + i * sz, sz);
    return out;
}

void cho_hash_crc32_batch(int type, const void * keys, size_t n, uint64_t * out)
{
    for (size_t i = 0; i < n; ++i)
        out[i] = cho_intHashCRC32(load_key_zext(type, keys, i));
}

void cho_weak_hash32(int type, const void * data, size_t n, uint32_t * hash) /* ColumnVector.cpp:78-95 */
{
    for (size_t i = 0; i < n; ++i)
        hash[i] = (uint32_t)cho_intHashCRC32_seed(load_key_zext(type, data, i), hash[i]);
}

uint32_t cho_two_level_bucket(uint64_t hash_value) /* TwoLevelHashTable.h:53, BITS_FOR_BUCKET = 8 */
{
    return (uint32_t)((hash_value >> (32 - 8)) & 0xFF);
}

void cho_crc32c_tables(uint32_t tables[8][256], uint32_t * constant)
{
    /* CRC is GF(2)-affine in the message: crc(seed, x) = crc(seed, 0) ^ XOR_j crc(0, byte j only). */
    for (int j = 0; j < 8; ++j)
        for (int b = 0; b < 256; ++b)
            tables[j][b] = (uint32_t)cho_intHashCRC32_soft((uint64_t)b << (8 * j), 0);
    *constant = (uint32_t)cho_intHashCRC32_soft(0, (uint64_t)-1);
}

/* ------------------------------------------------------------------------------------------------
 * a3  comparison  (FunctionsComparison.h:165-259 ; AccurateComparison.h:20-130,206-245)
 * ---------------------------------------------------------------------------------------------- */

/* A value of any supported type, carried exactly: integers as __int128, floats as long double
   (x86-64 long double has a 64-bit mantissa: Int64/UInt64/Float64 all convert exactly, so the
   comparison below is the mathematical comparison accurate::lessOp/equalsOp define). */
typedef struct
{
    int is_float;
    __int128 i;
    double f;
} num_t;

static inline num_t load_num(int type, const void * p, size_t idx)
{
    num_t r;
    r.is_float = 0;
    r.i = 0;
    r.f = 0;
    switch (type)
    {
        case CHO_I64: r.i = ((const int64_t *)p)[idx]; break;
        case CHO_U64: r.i = ((const uint64_t *)p)[idx]; break;
        case CHO_U32: r.i = ((const uint32_t *)p)[idx]; break;
        case CHO_I32: r.i = ((const int32_t *)p)[idx]; break;
        case CHO_U8: r.i = ((const uint8_t *)p)[idx]; break;
        case CHO_U16: r.i = ((const uint16_t *)p)[idx]; break;
        case CHO_I16: r.i = ((const int16_t *)p)[idx]; break;
        case CHO_I8: r.i = ((const int8_t *)p)[idx]; break;
        case CHO_F64: r.is_float = 1; r.f = ((const double *)p)[idx]; break;
        case CHO_F32: r.is_float = 1; r.f = (double)((const float *)p)[idx]; break; /* exact */
        default: break;
    }
    return r;
}

static inline int num_less(num_t a, num_t b) /* accurate::lessOp, AccurateComparison.h:20-72 */
{
    if (a.is_float && b.is_float)
        return a.f < b.f;
    if ((a.is_float && isnan(a.f)) || (b.is_float && isnan(b.f)))
        return 0;
    if (!a.is_float && !b.is_float)
        return a.i < b.i;
    long double x = a.is_float ? (long double)a.f : (long double)a.i;
    long double y = b.is_float ? (long double)b.f : (long double)b.i;
    return x < y;
}

static inline int num_equals(num_t a, num_t b) /* accurate::equalsOp, AccurateComparison.h:96-130 */
{
    if (a.is_float && b.is_float)
        return a.f == b.f;
    if ((a.is_float && isnan(a.f)) || (b.is_float && isnan(b.f)))
        return 0;
    if (!a.is_float && !b.is_float)
        return a.i == b.i;
    long double x = a.is_float ? (long double)a.f : (long double)a.i;
    long double y = b.is_float ? (long double)b.f : (long double)b.i;
    return x == y;
}

static inline int num_isnan(num_t a) { return a.is_float && isnan(a.f); }

static inline int num_cmp(int op, num_t a, num_t b)
{
    switch (op)
    {
        case CHO_EQ: return num_equals(a, b);
        case CHO_NE: return !num_equals(a, b);                                   /* notEqualsOp = !equalsOp */
        case CHO_LT: return num_less(a, b);
        case CHO_GT: return num_less(b, a);                                      /* greaterOp(a,b) = lessOp(b,a) */
        case CHO_LE: return (num_isnan(a) || num_isnan(b)) ? 0 : !num_less(b, a); /* AccurateComparison.h:85-91 */
        case CHO_GE: return (num_isnan(a) || num_isnan(b)) ? 0 : !num_less(a, b); /* :76-82 */
        default: return 0;
    }
}

/* Fast same-type Int64 loops (what NumComparisonImpl<Int64,Int64,Op>::vectorConstant compiles to). */
#define CMP_LOOP(T, OPSYM)                                 \
    {                                                      \
        const T * a_pos = (const T *)a;                    \
        const T b = *(const T *)scalar;                    \
        for (size_t i = 0; i < n; ++i)                     \
            c[i] = a_pos[i] OPSYM b;                       \
        return 0;                                          \
    }

CHO_MULTITARGET
static int cmp_const_same_i64(const void * a, size_t n, int op, const void * scalar, uint8_t * c)
{
    switch (op)
    {
        case CHO_EQ: CMP_LOOP(int64_t, ==)
        case CHO_NE: CMP_LOOP(int64_t, !=)
        case CHO_LT: CMP_LOOP(int64_t, <)
        case CHO_GT: CMP_LOOP(int64_t, >)
        case CHO_LE: CMP_LOOP(int64_t, <=)
        case CHO_GE: CMP_LOOP(int64_t, >=)
        default: return -1;
    }
}

int cho_cmp_const(int a_type, const void * a, size_t n, int op, int scalar_type, const void * scalar, uint8_t * c)
{
    if (op < CHO_EQ || op > CHO_GE || !type_size(a_type) || !type_size(scalar_type))
        return -1;
    if (a_type == CHO_I64 && scalar_type == CHO_I64)
        return cmp_const_same_i64(a, n, op, scalar, c);
    num_t b = load_num(scalar_type, scalar, 0);
    for (size_t i = 0; i < n; ++i) /* vectorConstantImpl: *c_pos = Op::apply(*a_pos, b) (FunctionsComparison.h:204-218) */
        c[i] = (uint8_t)num_cmp(op, load_num(a_type, a, i), b);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a4/a5  filter  (ColumnsCommon.h:27-72, ColumnsCommon.cpp:31-58, ColumnVector.cpp:528-724)
 * ---------------------------------------------------------------------------------------------- */

uint64_t cho_bytes64MaskToBits64Mask(const uint8_t * bytes64) /* portable branch, ColumnsCommon.h:66-71 */
{
    uint64_t res = 0;
    for (size_t i = 0; i < 64; ++i)
        res |= (uint64_t)(0 == bytes64[i]) << i;
    return ~res;
}

size_t cho_countBytesInFilter(const uint8_t * filt, size_t start, size_t end) /* ColumnsCommon.cpp:31-58 */
{
    size_t count = 0;
    const int8_t * pos = (const int8_t *)filt + start;
    const int8_t * end_pos = pos + (end - start);
    const int8_t * end_pos64 = pos + (end - start) / 64 * 64;
    for (; pos < end_pos64; pos += 64)
        count += (size_t)__builtin_popcountll(cho_bytes64MaskToBits64Mask((const uint8_t *)pos));
    for (; pos < end_pos; ++pos)
        count += *pos != 0;
    return count;
}

static uint8_t prefixToCopy(uint64_t mask) /* ColumnVector.cpp:539-551 */
{
    if (mask == 0)
        return 0;
    if (mask == (uint64_t)-1)
        return 64;
    const uint64_t leading_zeroes = (uint64_t)__builtin_clzll(mask);
    if (mask == ((((uint64_t)-1) << leading_zeroes) >> leading_zeroes))
        return (uint8_t)(64 - leading_zeroes);
    return 0xFF;
}

static uint8_t suffixToCopy(uint64_t mask) /* ColumnVector.cpp:553-557 */
{
    const uint8_t prefix_to_copy = prefixToCopy(~mask);
    return prefix_to_copy >= 64 ? prefix_to_copy : (uint8_t)(64 - prefix_to_copy);
}

int64_t cho_filter(int elem_size, const void * data, size_t n, const uint8_t * filt, size_t filt_n, void * out)
{
    if (n != filt_n)
        return -1; /* SIZES_OF_COLUMNS_DOESNT_MATCH, ColumnVector.cpp:685-686 */

    const size_t es = (size_t)elem_size;
    const uint8_t * filt_pos = filt;
    const uint8_t * filt_end = filt_pos + n;
    const char * data_pos = (const char *)data;
    char * res = (char *)out;
    const size_t SIMD_ELEMENTS = 64;
    const uint8_t * filt_end_aligned = filt_pos + n / SIMD_ELEMENTS * SIMD_ELEMENTS;

    /* Default doFilterAligned, ColumnVector.cpp:559-594 */
    while (filt_pos < filt_end_aligned)
    {
        uint64_t mask = cho_bytes64MaskToBits64Mask(filt_pos);
        const uint8_t prefix_to_copy = prefixToCopy(mask);
        if (0xFF != prefix_to_copy)
        {
            memcpy(res, data_pos, prefix_to_copy * es);
            res += prefix_to_copy * es;
        }
        else
        {
            const uint8_t suffix_to_copy = suffixToCopy(mask);
            if (0xFF != suffix_to_copy)
            {
                memcpy(res, data_pos + (SIMD_ELEMENTS - suffix_to_copy) * es, suffix_to_copy * es);
                res += suffix_to_copy * es;
            }
            else
            {
                while (mask)
                {
                    size_t index = (size_t)__builtin_ctzll(mask);
                    memcpy(res, data_pos + index * es, es);
                    res += es;
                    mask = mask & (mask - 1); /* blsr */
                }
            }
        }
        filt_pos += SIMD_ELEMENTS;
        data_pos += SIMD_ELEMENTS * es;
    }

    while (filt_pos < filt_end) /* tail, ColumnVector.cpp:714-722 */
    {
        if (*filt_pos)
        {
            memcpy(res, data_pos, es);
            res += es;
        }
        ++filt_pos;
        data_pos += es;
    }
    return (int64_t)((size_t)(res - (char *)out) / es);
}

void cho_filter_description_nullable(const uint8_t * data, const uint8_t * null_map, size_t n, uint8_t * res)
{
    for (size_t i = 0; i < n; ++i) /* FilterDescription.cpp:86-92 */
        res[i] = data[i] && !null_map[i];
}

/* ------------------------------------------------------------------------------------------------
 * a22  index / replicate / scatter
 * ---------------------------------------------------------------------------------------------- */

void cho_index(int elem_size, const void * data, const uint64_t * indexes, size_t limit, void * out)
{
    for (size_t i = 0; i < limit; ++i) /* ColumnVector.cpp:1121-1143: res_data[i] = data[indexes[i]] */
        memcpy((char *)out + i * (size_t)elem_size, (const char *)data + indexes[i] * (size_t)elem_size, (size_t)elem_size);
}

void cho_replicate(int elem_size, const void * data, size_t n, const uint64_t * offsets, void * out)
{
    /* ColumnVector.cpp:879-907: row i repeated offsets[i]-offsets[i-1] times */
    uint64_t prev = 0;
    char * o = (char *)out;
    for (size_t i = 0; i < n; ++i)
    {
        uint64_t cnt = offsets[i] - prev;
        prev = offsets[i];
        for (uint64_t k = 0; k < cnt; ++k)
        {
            memcpy(o, (const char *)data + i * (size_t)elem_size, (size_t)elem_size);
            o += elem_size;
        }
    }
}

void cho_scatter(int elem_size, const void * data, size_t n, const uint64_t * selector, size_t num_columns,
                 void * out_concat, uint64_t * out_sizes)
{
    /* IColumn::scatterImpl (IColumn.cpp:245-269): columns[selector[i]]->insertFrom(*this, i), i ascending */
    memset(out_sizes, 0, num_columns * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i)
        ++out_sizes[selector[i]];
    uint64_t * cursor = (uint64_t *)malloc(num_columns * sizeof(uint64_t));
    uint64_t acc = 0;
    for (size_t k = 0; k < num_columns; ++k)
    {
        cursor[k] = acc;
        acc += out_sizes[k];
    }
    for (size_t i = 0; i < n; ++i)
        memcpy((char *)out_concat + cursor[selector[i]]++ * (size_t)elem_size,
               (const char *)data + i * (size_t)elem_size, (size_t)elem_size);
    free(cursor);
}

/* ------------------------------------------------------------------------------------------------
 * a8/a9  sum / count / avg  (AggregateFunctionSum.h:33-303, Count.h:26-135, Avg.h:37-285)
 * ---------------------------------------------------------------------------------------------- */

/* Integer sums wrap modulo 2^64 (NO_SANITIZE_UNDEFINED, AggregateFunctionSum.h:36-39) -> unsigned adds. */
#define SUM_INT_LOOP(VT)                                              \
    {                                                                 \
        const VT * p = (const VT *)ptr + start;                       \
        const VT * end_ptr = p + (end - start);                       \
        uint64_t local_sum = 0;                                       \
        while (p < end_ptr)                                           \
        {                                                             \
            local_sum += (uint64_t)(*p);                              \
            ++p;                                                      \
        }                                                             \
        *(uint64_t *)state += local_sum;                              \
    }

CHO_MULTITARGET
static void sum_add_many_f64(double * sum, const double * ptr, size_t start, size_t end)
{
    /* AggregateFunctionSum.h:72-101: unroll_count = 128/sizeof(T) = 16 independent lanes,
       folded into sum in lane order, then a scalar tail in local_sum. */
    ptr += start;
    size_t count = end - start;
    const double * end_ptr = ptr + count;
    enum { unroll_count = 16 };
    double partial_sums[unroll_count];
    for (int i = 0; i < unroll_count; ++i)
        partial_sums[i] = 0;
    const double * unrolled_end = ptr + (count / unroll_count * unroll_count);
    while (ptr < unrolled_end)
    {
        for (int i = 0; i < unroll_count; ++i)
            partial_sums[i] += ptr[i];
        ptr += unroll_count;
    }
    for (int i = 0; i < unroll_count; ++i)
        *sum += partial_sums[i];
    double local_sum = 0;
    while (ptr < end_ptr)
    {
        local_sum += *ptr;
        ++ptr;
    }
    *sum += local_sum;
}

/* sum(Float32): the accumulator type is Float64 (SumSimple: NearestFieldType<Float32>), so the loop is the Float64 one with
   T(ptr[i]) converting every value (AggregateFunctionSum.h:72-101: unroll_count = 128 / sizeof(Float64) = 16) */
static void sum_add_many_f32(double * sum, const float * ptr, size_t start, size_t end)
{
    ptr += start;
    size_t count = end - start;
    const float * end_ptr = ptr + count;
    enum { unroll_count = 16 };
    double partial_sums[unroll_count];
    for (int i = 0; i < unroll_count; ++i)
        partial_sums[i] = 0;
    const float * unrolled_end = ptr + (count / unroll_count * unroll_count);
    while (ptr < unrolled_end)
    {
        for (int i = 0; i < unroll_count; ++i)
            partial_sums[i] += (double)ptr[i];
        ptr += unroll_count;
    }
    for (int i = 0; i < unroll_count; ++i)
        *sum += partial_sums[i];
    double local_sum = 0;
    while (ptr < end_ptr)
    {
        local_sum += (double)*ptr;
        ++ptr;
    }
    *sum += local_sum;
}

CHO_MULTITARGET
static void sum_add_many_i64(void * state, const void * ptr, size_t start, size_t end) SUM_INT_LOOP(int64_t)

void cho_sum_add_many(int type, void * state, const void * ptr, size_t start, size_t end)
{
    switch (type)
    {
        case CHO_I64: sum_add_many_i64(state, ptr, start, end); break;
        case CHO_U64: SUM_INT_LOOP(uint64_t) break;
        case CHO_U32: SUM_INT_LOOP(uint32_t) break;
        case CHO_I32: SUM_INT_LOOP(int32_t) break;
        case CHO_U8: SUM_INT_LOOP(uint8_t) break;
        case CHO_U16: SUM_INT_LOOP(uint16_t) break;
        case CHO_I16: SUM_INT_LOOP(int16_t) break;
        case CHO_I8: SUM_INT_LOOP(int8_t) break;
        case CHO_F64: sum_add_many_f64((double *)state, (const double *)ptr, start, end); break;
        case CHO_F32: sum_add_many_f32((double *)state, (const float *)ptr, start, end); break;
        default: break;
    }
}

#define SUM_INT_COND_LOOP(VT)                                                             \
    {                                                                                     \
        /* AggregateFunctionSum.h:149-162: multiply by 0/1 */                             \
        const VT * p = (const VT *)ptr + start;                                           \
        const uint8_t * cm = cond + start;                                                \
        const VT * end_ptr = p + (end - start);                                           \
        uint64_t local_sum = 0;                                                           \
        while (p < end_ptr)                                                               \
        {                                                                                 \
            uint64_t multiplier = !*cm == 0; /* add_if_zero = false */                    \
            local_sum += (uint64_t)(*p) * multiplier;                                     \
            ++p;                                                                          \
            ++cm;                                                                         \
        }                                                                                 \
        *(uint64_t *)state += local_sum;                                                  \
    }

void cho_sum_add_many_conditional(int type, void * state, const void * ptr, const uint8_t * cond, size_t start, size_t end)
{
    switch (type)
    {
        case CHO_I64: SUM_INT_COND_LOOP(int64_t) break;
        case CHO_U64: SUM_INT_COND_LOOP(uint64_t) break;
        case CHO_U32: SUM_INT_COND_LOOP(uint32_t) break;
        case CHO_I32: SUM_INT_COND_LOOP(int32_t) break;
        case CHO_U8: SUM_INT_COND_LOOP(uint8_t) break;
        case CHO_U16: SUM_INT_COND_LOOP(uint16_t) break;
        case CHO_I16: SUM_INT_COND_LOOP(int16_t) break;
        case CHO_I8: SUM_INT_COND_LOOP(int8_t) break;
        case CHO_F64:
        {
            /* AggregateFunctionSum.h:196-235: mask trick over 16 lanes, then branchy tail */
            const double * p = (const double *)ptr + start;
            const uint8_t * cm = cond + start;
            size_t count = end - start;
            const double * end_ptr = p + count;
            enum { unroll_count = 16 };
            double partial_sums[unroll_count];
            for (int i = 0; i < unroll_count; ++i)
                partial_sums[i] = 0;
            const double * unrolled_end = p + (count / unroll_count * unroll_count);
            while (p < unrolled_end)
            {
                for (int i = 0; i < unroll_count; ++i)
                {
                    uint64_t value;
                    memcpy(&value, &p[i], 8);
                    value &= (uint64_t)((!cm[i] != 0) - 1); /* (!condition_map[i] != add_if_zero) - 1 */
                    double d;
                    memcpy(&d, &value, 8);
                    partial_sums[i] += d;
                }
                p += unroll_count;
                cm += unroll_count;
            }
            double * sum = (double *)state;
            for (int i = 0; i < unroll_count; ++i)
                *sum += partial_sums[i];
            double local_sum = 0;
            while (p < end_ptr)
            {
                if (!*cm == 0)
                    local_sum += *p;
                ++p;
                ++cm;
            }
            *sum += local_sum;
            break;
        }
        case CHO_F32:
        {
            /* same structure as the Float64 case on converted values */
            const float * p = (const float *)ptr + start;
            const uint8_t * cm = cond + start;
            size_t count = end - start;
            double partial_sums[16];
            for (int i = 0; i < 16; ++i)
                partial_sums[i] = 0;
            size_t k = 0;
            for (; k + 16 <= count; k += 16)
                for (int i = 0; i < 16; ++i)
                    partial_sums[i] += cm[k + i] ? (double)p[k + i] : 0.0;
            double * sum = (double *)state;
            for (int i = 0; i < 16; ++i)
                *sum += partial_sums[i];
            double local_sum = 0;
            for (; k < count; ++k)
                if (cm[k])
                    local_sum += (double)p[k];
            *sum += local_sum;
            break;
        }
        default: break;
    }
}

double cho_avg_divide(int numerator_type, const void * numerator, uint64_t denominator)
{
    /* AvgFraction::divide (AggregateFunctionAvg.h:61-67): static_cast<Float64>(numerator) / denominator */
    switch (numerator_type)
    {
        case CHO_I64: case CHO_I32: case CHO_I16: case CHO_I8: return (double)(*(const int64_t *)numerator) / (double)denominator;
        case CHO_U64: case CHO_U32: case CHO_U16: case CHO_U8: return (double)(*(const uint64_t *)numerator) / (double)denominator;
        case CHO_F64: case CHO_F32: return *(const double *)numerator / (double)denominator;
        default: return NAN;
    }
}

/* SumSimple result type (AggregateFunctionSum.cpp:19-28) */
static int sum_result_type(int arg_type)
{
    switch (arg_type)
    {
        case CHO_I64: case CHO_I32: case CHO_I16: case CHO_I8: return CHO_I64;
        case CHO_U64: case CHO_U32: case CHO_U16: case CHO_U8: return CHO_U64;
        default: return CHO_F64;
    }
}

/* ---- the C1/C2 pipeline per Block ---- */

typedef struct
{
    int type;
    const char * pred;
    const char * val;
    size_t begin, end; /* row range of this stream */
    int op;
    const void * scalar;
    size_t block_rows;
    uint64_t sum_state; /* 8-byte state, reinterpret as double for F64 */
    uint64_t count;
    uint64_t dropped, passthrough;
} fs_stream;

static void * filter_sum_stream(void * arg)
{
    fs_stream * s = (fs_stream *)arg;
    const size_t es = type_size(s->type);
    uint8_t * mask = (uint8_t *)malloc(s->block_rows + 64);
    char * filtered = (char *)malloc((s->block_rows + 64) * es);
    s->sum_state = 0;
    s->count = 0;
    s->dropped = s->passthrough = 0;
    for (size_t b = s->begin; b < s->end; b += s->block_rows)
    {
        size_t rows = s->end - b < s->block_rows ? s->end - b : s->block_rows;
        const char * pcol = s->pred + b * es;
        const char * vcol = s->val + b * es;
        /* FilterTransform::doTransform (FilterTransform.cpp:136-256): run the expression -> UInt8 column */
        cho_cmp_const(s->type, pcol, rows, s->op, s->type, s->scalar, mask);
        /* count first via the narrowest column == the value column here (:192-216) */
        size_t num_filtered_rows = cho_countBytesInFilter(mask, 0, rows);
        if (num_filtered_rows == 0)
        {
            ++s->dropped; /* chunk dropped, FilterTransform.cpp:221-226 */
            continue;
        }
        const char * agg_input = vcol;
        if (num_filtered_rows == rows)
            ++s->passthrough; /* all rows pass: columns untouched, :229-235 */
        else
        {
            cho_filter((int)es, vcol, rows, mask, rows, filtered); /* IColumn::filter, :238-252 */
            agg_input = filtered;
        }
        /* AggregatingTransform::consume -> Aggregator::executeOnBlock -> executeWithoutKeyImpl
           (Aggregator.cpp:1276-1321) -> addBatchSinglePlace for sum(a), count() */
        cho_sum_add_many(s->type, &s->sum_state, agg_input, 0, num_filtered_rows);
        s->count += num_filtered_rows; /* AggregateFunctionCount::addBatchSinglePlace, Count.h:54-70 */
    }
    free(mask);
    free(filtered);
    return NULL;
}

int cho_filter_sum_pipeline(int type, const void * pred, const void * val, size_t n, int op, const void * scalar,
                            size_t block_rows, int threads, void * sum_out, uint64_t * count_out,
                            uint64_t * chunks_dropped, uint64_t * chunks_passthrough)
{
    if (!type_size(type) || type == CHO_U8)
        return -1;
    if (threads < 1)
        threads = 1;
    if (!val)
        val = pred;
    if (!block_rows)
        block_rows = CHO_DEFAULT_BLOCK_SIZE;
    fs_stream * st = (fs_stream *)calloc((size_t)threads, sizeof(fs_stream));
    pthread_t * th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    /* pipeline.resize(max_threads): contiguous block ranges per stream (AggregatingStep.cpp:498-517) */
    size_t n_blocks = (n + block_rows - 1) / block_rows;
    for (int t = 0; t < threads; ++t)
    {
        size_t b0 = n_blocks * (size_t)t / (size_t)threads;
        size_t b1 = n_blocks * (size_t)(t + 1) / (size_t)threads;
        st[t].type = type;
        st[t].pred = (const char *)pred;
        st[t].val = (const char *)val;
        st[t].begin = b0 * block_rows;
        st[t].end = b1 * block_rows < n ? b1 * block_rows : n;
        if (st[t].begin > n)
            st[t].begin = n;
        st[t].op = op;
        st[t].scalar = scalar;
        st[t].block_rows = block_rows;
    }
    if (threads == 1)
        filter_sum_stream(&st[0]);
    else
    {
        for (int t = 0; t < threads; ++t)
            pthread_create(&th[t], NULL, filter_sum_stream, &st[t]);
        for (int t = 0; t < threads; ++t)
            pthread_join(th[t], NULL);
    }
    /* mergeWithoutKeyDataImpl (Aggregator.cpp:2584-2628): states folded into the first in stream order */
    uint64_t cnt = 0, dropped = 0, pass = 0;
    if (sum_result_type(type) == CHO_F64)
    {
        double s = 0;
        for (int t = 0; t < threads; ++t)
        {
            double d;
            memcpy(&d, &st[t].sum_state, 8);
            s += d;
        }
        memcpy(sum_out, &s, 8);
    }
    else
    {
        uint64_t s = 0;
        for (int t = 0; t < threads; ++t)
            s += st[t].sum_state;
        memcpy(sum_out, &s, 8);
    }
    for (int t = 0; t < threads; ++t)
    {
        cnt += st[t].count;
        dropped += st[t].dropped;
        pass += st[t].passthrough;
    }
    *count_out = cnt;
    if (chunks_dropped)
        *chunks_dropped = dropped;
    if (chunks_passthrough)
        *chunks_passthrough = pass;
    free(st);
    free(th);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * §8(f)-1  and / multiply / plus / minus and the Q1.1-style pipeline
 * ---------------------------------------------------------------------------------------------- */

void cho_and_u8(const uint8_t * a, const uint8_t * b, size_t n, uint8_t * out)
{
    for (size_t i = 0; i < n; ++i) /* AndImpl::apply (FunctionsLogical.h:94) */
        out[i] = a[i] & b[i];
}

/* Type of sum(a OP b) over integer columns: the operation's result type (NumberTraits.h:73-87: the next size above the wider
   operand, signed if either is or for minus) summed (AggregateFunctionSum: Int64 for signed, UInt64 for unsigned). */
int cho_arith_sum_type(int value_op, int a_type, int b_type)
{
    if (a_type == CHO_F64 || b_type == CHO_F64 || a_type == CHO_F32 || b_type == CHO_F32 || !type_size(a_type) || !type_size(b_type))
        return -1;
    if (a_type > CHO_I32 || b_type > CHO_I32)
        return -1; /* UInt16 / Int16 / Int8 operands: arithmetic is not restated for them */
    const int sgn_a = a_type == CHO_I64 || a_type == CHO_I32, sgn_b = b_type == CHO_I64 || b_type == CHO_I32;
    if (value_op == CHO_VAL_MINUS)
        return CHO_I64;
    if (value_op == CHO_VAL_MUL || value_op == CHO_VAL_PLUS)
        return (sgn_a || sgn_b) ? CHO_I64 : CHO_U64;
    return -1;
}

/* Result type of a OP b as a column.  Operands of 4 or 8 bytes promote to 8 bytes; two 1-byte operands promote to a 2-byte
   type (UInt16 / Int16), which this path does not carry: -1. */
int cho_arith_result_type(int value_op, int a_type, int b_type)
{
    if (type_size(a_type) == 1 && type_size(b_type) == 1)
        return -1;
    return cho_arith_sum_type(value_op, a_type, b_type);
}

static inline uint64_t load_int_as_u64(int type, const void * p, size_t i)
{
    switch (type)
    {
        case CHO_I64: return (uint64_t)((const int64_t *)p)[i];
        case CHO_U64: return ((const uint64_t *)p)[i];
        case CHO_U32: return ((const uint32_t *)p)[i];
        case CHO_I32: return (uint64_t)(int64_t)((const int32_t *)p)[i];
        case CHO_U8: return ((const uint8_t *)p)[i];
        default: return 0;
    }
}

/* values of a OP b in 64-bit two's complement: exact for every result type up to 8 bytes (a 2-byte result cannot overflow) */
static void arith_u64(int value_op, int a_type, const void * a, int b_type, const void * b, size_t n, uint64_t * o);

int cho_arith(int value_op, int a_type, const void * a, int b_type, const void * b, size_t n, void * out)
{
    if (cho_arith_result_type(value_op, a_type, b_type) < 0)
        return -1;
    arith_u64(value_op, a_type, a, b_type, b, n, (uint64_t *)out);
    return 0;
}

static void arith_u64(int value_op, int a_type, const void * a, int b_type, const void * b, size_t n, uint64_t * o)
{
    for (size_t i = 0; i < n; ++i)
    {
        /* static_cast<Result>(a) OP b in the 64-bit result type; two's complement wrap (NO_SANITIZE_UNDEFINED) */
        const uint64_t x = load_int_as_u64(a_type, a, i), y = load_int_as_u64(b_type, b, i);
        o[i] = value_op == CHO_VAL_MUL ? x * y : value_op == CHO_VAL_PLUS ? x + y : x - y;
    }
}

typedef struct
{
    size_t n_cols;
    const int * cols_type;
    const void * const * cols;
    size_t begin, end;
    size_t n_preds;
    const uint32_t * pred_col;
    const int * pred_op;
    const int * pred_stype;
    const uint64_t * pred_scalar_bits;
    int value_op;
    uint32_t val_a, val_b;
    size_t block_rows;
    uint64_t sum_state, count;
} ex_stream;

static void * expr_stream(void * arg)
{
    ex_stream * s = (ex_stream *)arg;
    const size_t br = s->block_rows;
    uint8_t * mask = (uint8_t *)malloc(br + 64);
    uint8_t * tmp = (uint8_t *)malloc(br + 64);
    char * fa = (char *)malloc((br + 64) * 8);
    char * fb = (char *)malloc((br + 64) * 8);
    uint64_t * val = (uint64_t *)malloc((br + 64) * 8);
    s->sum_state = 0;
    s->count = 0;
    for (size_t b = s->begin; b < s->end; b += br)
    {
        const size_t rows = s->end - b < br ? s->end - b : br;
        /* WHERE: one comparison function per predicate, combined by `and` (ExpressionActions::execute) */
        for (size_t k = 0; k < s->n_preds; ++k)
        {
            const int t = s->cols_type[s->pred_col[k]];
            const char * col = (const char *)s->cols[s->pred_col[k]] + b * type_size(t);
            uint8_t * dst = k == 0 ? mask : tmp;
            cho_cmp_const(t, col, rows, s->pred_op[k], s->pred_stype[k], &s->pred_scalar_bits[k], dst);
            if (k > 0)
                cho_and_u8(mask, tmp, rows, mask);
        }
        if (s->n_preds == 0)
            memset(mask, 1, rows);
        const size_t kept = cho_countBytesInFilter(mask, 0, rows);
        if (kept == 0)
            continue; /* FilterTransform drops the chunk */
        /* FilterTransform filters the columns the projection needs; then the value expression runs on the filtered Block */
        const int ta = s->cols_type[s->val_a];
        const char * ca = (const char *)s->cols[s->val_a] + b * type_size(ta);
        const void * pa = ca;
        if (kept != rows)
        {
            cho_filter((int)type_size(ta), ca, rows, mask, rows, fa);
            pa = fa;
        }
        if (s->value_op == CHO_VAL_COL)
            cho_sum_add_many(ta, &s->sum_state, pa, 0, kept);
        else
        {
            const int tb = s->cols_type[s->val_b];
            const char * cb = (const char *)s->cols[s->val_b] + b * type_size(tb);
            const void * pb = cb;
            if (kept != rows)
            {
                cho_filter((int)type_size(tb), cb, rows, mask, rows, fb);
                pb = fb;
            }
            arith_u64(s->value_op, ta, pa, tb, pb, kept, (uint64_t *)val);
            cho_sum_add_many(cho_arith_sum_type(s->value_op, ta, tb), &s->sum_state, val, 0, kept);
        }
        s->count += kept;
    }
    free(mask);
    free(tmp);
    free(fa);
    free(fb);
    free(val);
    return NULL;
}

int cho_expr_filter_sum_pipeline(size_t n_cols, const int * cols_type, const void * const * cols, size_t n,
                                 size_t n_preds, const uint32_t * pred_col, const int * pred_op, const int * pred_stype,
                                 const uint64_t * pred_scalar_bits, int value_op, uint32_t val_a, uint32_t val_b,
                                 size_t block_rows, int threads, void * sum_out, uint64_t * count_out)
{
    if (val_a >= n_cols || (value_op != CHO_VAL_COL && val_b >= n_cols))
        return -1;
    for (size_t k = 0; k < n_preds; ++k)
        if (pred_col[k] >= n_cols)
            return -1;
    for (size_t c = 0; c < n_cols; ++c)
        if (cols_type[c] == CHO_F64 || !type_size(cols_type[c]))
            return -1;
    if (threads < 1)
        threads = 1;
    if (!block_rows)
        block_rows = CHO_DEFAULT_BLOCK_SIZE;
    ex_stream * st = (ex_stream *)calloc((size_t)threads, sizeof(ex_stream));
    pthread_t * th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    const size_t n_blocks = (n + block_rows - 1) / block_rows;
    for (int t = 0; t < threads; ++t)
    {
        const size_t b0 = n_blocks * (size_t)t / (size_t)threads, b1 = n_blocks * (size_t)(t + 1) / (size_t)threads;
        st[t].n_cols = n_cols;
        st[t].cols_type = cols_type;
        st[t].cols = cols;
        st[t].begin = b0 * block_rows > n ? n : b0 * block_rows;
        st[t].end = b1 * block_rows > n ? n : b1 * block_rows;
        st[t].n_preds = n_preds;
        st[t].pred_col = pred_col;
        st[t].pred_op = pred_op;
        st[t].pred_stype = pred_stype;
        st[t].pred_scalar_bits = pred_scalar_bits;
        st[t].value_op = value_op;
        st[t].val_a = val_a;
        st[t].val_b = val_b;
        st[t].block_rows = block_rows;
    }
    if (threads == 1)
        expr_stream(&st[0]);
    else
    {
        for (int t = 0; t < threads; ++t)
            pthread_create(&th[t], NULL, expr_stream, &st[t]);
        for (int t = 0; t < threads; ++t)
            pthread_join(th[t], NULL);
    }
    uint64_t sum = 0, cnt = 0;
    for (int t = 0; t < threads; ++t)
    {
        sum += st[t].sum_state; /* integer states: wrap-around merge */
        cnt += st[t].count;
    }
    memcpy(sum_out, &sum, 8);
    *count_out = cnt;
    free(st);
    free(th);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a12/a13  hash tables
 * ---------------------------------------------------------------------------------------------- */

#define HT_NAME u64map
#define HT_MAPPED uint64_t
#include "ch_hashtable.inc"

struct cho_hashmap
{
    u64map_t t;
};

cho_hashmap * cho_hashmap_create(void)
{
    cho_hashmap * m = (cho_hashmap *)malloc(sizeof(*m));
    u64map_init(&m->t, 0);
    return m;
}
void cho_hashmap_free(cho_hashmap * m)
{
    if (!m)
        return;
    u64map_destroy(&m->t);
    free(m);
}
int cho_hashmap_emplace(cho_hashmap * m, uint64_t key, uint64_t ** mapped_out)
{
    int inserted;
    u64map_cell * c = u64map_emplace(&m->t, key, &inserted);
    if (mapped_out)
        *mapped_out = &c->mapped;
    return inserted;
}
uint64_t * cho_hashmap_find(cho_hashmap * m, uint64_t key)
{
    u64map_cell * c = u64map_find(&m->t, key);
    return c ? &c->mapped : NULL;
}
void cho_hashmap_reserve(cho_hashmap * m, size_t num_elements) { u64map_reserve(&m->t, num_elements); } /* HashTable.h:962-965 */
size_t cho_hashmap_size(const cho_hashmap * m) { return m->t.m_size; }
size_t cho_hashmap_buf_size(const cho_hashmap * m) { return u64map_buf_size(&m->t); }
int cho_hashmap_has_zero(const cho_hashmap * m) { return m->t.has_zero; }
size_t cho_hashmap_dump(const cho_hashmap * m, uint64_t * keys, uint64_t * values)
{
    size_t n = 0;
    u64map_t * t = (u64map_t *)&m->t;
    for (u64map_cell * c = u64map_first(t); c; c = u64map_next_cell(t, c))
    {
        keys[n] = c->key;
        values[n] = c->mapped;
        ++n;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * Arena (src/Common/Arena.h:192 alignedAlloc) — bump allocator, chunks never move.
 * ---------------------------------------------------------------------------------------------- */

typedef struct arena_chunk
{
    struct arena_chunk * prev;
    size_t cap, used;
    char data[];
} arena_chunk;

typedef struct
{
    arena_chunk * head;
    size_t total;
} arena_t;

static void * arena_aligned_alloc(arena_t * a, size_t size, size_t align)
{
    for (;;)
    {
        if (a->head)
        {
            size_t off = (a->head->used + align - 1) & ~(align - 1);
            if (off + size <= a->head->cap)
            {
                a->head->used = off + size;
                return a->head->data + off;
            }
        }
        size_t cap = a->head ? a->head->cap * 2 : 4096;
        while (cap < size + align)
            cap *= 2;
        arena_chunk * c = (arena_chunk *)malloc(sizeof(arena_chunk) + cap);
        c->prev = a->head;
        c->cap = cap;
        c->used = 0;
        a->head = c;
        a->total += cap;
    }
}

static void arena_free(arena_t * a)
{
    while (a->head)
    {
        arena_chunk * p = a->head->prev;
        free(a->head);
        a->head = p;
    }
    a->total = 0;
}

/* ------------------------------------------------------------------------------------------------
 * a14-a17  Aggregator
 * ---------------------------------------------------------------------------------------------- */

#define HT_NAME aggmap
#define HT_MAPPED char *
#include "ch_hashtable.inc"

#define CHO_MAX_AGGS 8
#define CHO_NUM_BUCKETS 256

struct cho_agg
{
    int key_type; /* -1 = without_key */
    int n_aggs;
    int kinds[CHO_MAX_AGGS];
    int arg_types[CHO_MAX_AGGS];
    size_t offsets[CHO_MAX_AGGS]; /* offsets_of_aggregate_states (Aggregator.cpp:467-494) */
    size_t total_size_of_aggregate_states;
    uint64_t two_level_threshold;
    int is_two_level;
    aggmap_t single;              /* AggregatedDataWithUInt64Key (AggregatedData.h:38) */
    aggmap_t * impls;             /* AggregatedDataWithUInt64KeyTwoLevel: 256 sub-tables */
    char * without_key;           /* AggregatedDataWithoutKey */
    arena_t arenas[512];           /* aggregates_pools: own + adopted on merge */
    int n_arenas;
};

/* min / max: SingleValueDataFixed<T> {has_value, value} (src/AggregateFunctions/SingleValueData.h) -- here 8 bytes of flag + the value widened
   to 8 bytes (integers sign- / zero-extended, Float32 widened to Float64: both order-preserving and exactly reversible) */
static size_t state_size(int kind) { return (kind == CHO_AGG_AVG || kind == CHO_AGG_MIN || kind == CHO_AGG_MAX || kind == CHO_AGG_ANY) ? 16 : 8; }

static int is_signed_type(int t) { return t == CHO_I64 || t == CHO_I32 || t == CHO_I16 || t == CHO_I8; }
static int is_float_type(int t) { return t == CHO_F64 || t == CHO_F32; }
static uint64_t load_widened(int t, const void * arg, size_t i)
{
    switch (t)
    {
        case CHO_I64: return (uint64_t)((const int64_t *)arg)[i];
        case CHO_U64: return ((const uint64_t *)arg)[i];
        case CHO_U32: return ((const uint32_t *)arg)[i];
        case CHO_I32: return (uint64_t)(int64_t)((const int32_t *)arg)[i];
        case CHO_U16: return ((const uint16_t *)arg)[i];
        case CHO_I16: return (uint64_t)(int64_t)((const int16_t *)arg)[i];
        case CHO_U8: return ((const uint8_t *)arg)[i];
        case CHO_I8: return (uint64_t)(int64_t)((const int8_t *)arg)[i];
        case CHO_F64: return ((const uint64_t *)arg)[i];
        case CHO_F32: { double d = (double)((const float *)arg)[i]; uint64_t b; memcpy(&b, &d, 8); return b; }
        default: return 0;
    }
}
/* `to < value` / `to > value` in the argument's own type (SingleValueDataFixed<T>::setIfSmaller / setIfGreater, SingleValueData.cpp:219-240) */
static int widened_less(int t, uint64_t x, uint64_t y)
{
    if (is_float_type(t)) { double a, b; memcpy(&a, &x, 8); memcpy(&b, &y, 8); return a < b; }
    if (is_signed_type(t)) return (int64_t)x < (int64_t)y;
    return x < y;
}
static void extremum_update(int kind, int t, char * st, uint64_t v)
{
    uint64_t * has = (uint64_t *)st, * val = (uint64_t *)(st + 8);
    /* any: SingleValueDataFixed<T>::setIfFirst / changeFirstTime (AggregateFunctionAny.cpp: add -> data().setIfFirst; SingleValueData.cpp) --
       the first value the state is offered stays */
    if (kind == CHO_AGG_ANY ? !*has : (!*has || (kind == CHO_AGG_MIN ? widened_less(t, v, *val) : widened_less(t, *val, v))))
    {
        *has = 1;
        *val = v;
    }
}

cho_agg * cho_agg_create(int key_type, int n_aggs, const int * kinds, const int * arg_types, uint64_t two_level_threshold)
{
    if (n_aggs < 0 || n_aggs > CHO_MAX_AGGS)
        return NULL;
    cho_agg * a = (cho_agg *)calloc(1, sizeof(*a));
    a->key_type = key_type;
    a->n_aggs = n_aggs;
    size_t off = 0;
    for (int j = 0; j < n_aggs; ++j)
    {
        a->kinds[j] = kinds[j];
        a->arg_types[j] = arg_types ? arg_types[j] : CHO_I64;
        a->offsets[j] = off; /* all states are 8-byte aligned multiples of 8 */
        off += state_size(kinds[j]);
    }
    a->total_size_of_aggregate_states = off ? off : 8;
    a->two_level_threshold = two_level_threshold;
    a->n_arenas = 1;
    aggmap_init(&a->single, 0);
    if (key_type < 0)
    {
        /* without_key state is created up-front (Aggregator.cpp:1519-1526) */
        a->without_key = (char *)arena_aligned_alloc(&a->arenas[0], a->total_size_of_aggregate_states, 8);
        memset(a->without_key, 0, a->total_size_of_aggregate_states);
    }
    return a;
}

void cho_agg_free(cho_agg * a)
{
    if (!a)
        return;
    aggmap_destroy(&a->single);
    if (a->impls)
    {
        for (int b = 0; b < CHO_NUM_BUCKETS; ++b)
            aggmap_destroy(&a->impls[b]);
        free(a->impls);
    }
    for (int i = 0; i < a->n_arenas; ++i)
        arena_free(&a->arenas[i]);
    free(a);
}

/* IAggregateFunction::add for row i (Sum.h:470-480, Count.h:49-52, Avg.h:237-262) */
static inline void agg_add_row(const cho_agg * a, int j, char * place, const void * arg, size_t i)
{
    char * st = place + a->offsets[j];
    switch (a->kinds[j])
    {
        case CHO_AGG_COUNT:
            ++*(uint64_t *)st;
            break;
        case CHO_AGG_MIN:
        case CHO_AGG_MAX:
        case CHO_AGG_ANY:
            extremum_update(a->kinds[j], a->arg_types[j], st, load_widened(a->arg_types[j], arg, i));
            break;
        case CHO_AGG_AVG:
            ++*(uint64_t *)(st + 8); /* denominator */
            /* numerator accumulates like sum */
            __attribute__((fallthrough));
        case CHO_AGG_SUM:
            switch (a->arg_types[j])
            {
                case CHO_I64: *(uint64_t *)st += (uint64_t)((const int64_t *)arg)[i]; break;
                case CHO_U64: *(uint64_t *)st += ((const uint64_t *)arg)[i]; break;
                case CHO_U32: *(uint64_t *)st += ((const uint32_t *)arg)[i]; break;
                case CHO_I32: *(uint64_t *)st += (uint64_t)(int64_t)((const int32_t *)arg)[i]; break;
                case CHO_U8: *(uint64_t *)st += ((const uint8_t *)arg)[i]; break;
                case CHO_U16: *(uint64_t *)st += ((const uint16_t *)arg)[i]; break;
                case CHO_I16: *(uint64_t *)st += (uint64_t)(int64_t)((const int16_t *)arg)[i]; break;
                case CHO_I8: *(uint64_t *)st += (uint64_t)(int64_t)((const int8_t *)arg)[i]; break;
                case CHO_F64: *(double *)st += ((const double *)arg)[i]; break;
                case CHO_F32: *(double *)st += (double)((const float *)arg)[i]; break; /* Impl::add(sum, T(value)), T = Float64 */
                default: break;
            }
            break;
        default: break;
    }
}

/* IAggregateFunction::merge (Sum.h:283-286, Count.h:110-113, Avg.h:134-138) */
static void agg_merge_states(const cho_agg * a, char * dst, const char * src)
{
    for (int j = 0; j < a->n_aggs; ++j)
    {
        char * d = dst + a->offsets[j];
        const char * s = src + a->offsets[j];
        if (a->kinds[j] == CHO_AGG_MIN || a->kinds[j] == CHO_AGG_MAX || a->kinds[j] == CHO_AGG_ANY)
        {
            /* setIfSmaller(const SingleValueDataFixed &) / setIfGreater (SingleValueData.cpp:243-262): `to.has() && (!has() || to.value < value)` */
            if (*(const uint64_t *)s)
                extremum_update(a->kinds[j], a->arg_types[j], d, *(const uint64_t *)(s + 8));
            continue;
        }
        int is_f = (a->kinds[j] != CHO_AGG_COUNT) && sum_result_type(a->arg_types[j]) == CHO_F64;
        if (is_f)
            *(double *)d += *(const double *)s;
        else
            *(uint64_t *)d += *(const uint64_t *)s;
        if (a->kinds[j] == CHO_AGG_AVG)
            *(uint64_t *)(d + 8) += *(const uint64_t *)(s + 8);
    }
}

static aggmap_t * agg_table_for_hash(cho_agg * a, size_t hash_value)
{
    if (!a->is_two_level)
        return &a->single;
    return &a->impls[cho_two_level_bucket(hash_value)];
}

/* AggregatedDataVariants::convertToTwoLevel (AggregatedDataVariants.cpp:158-179) ->
   TwoLevelHashTable(const Source &) (TwoLevelHashTable.h:100-120) */
static void agg_convert_to_two_level(cho_agg * a)
{
    if (a->is_two_level || a->key_type < 0)
        return;
    a->impls = (aggmap_t *)malloc(sizeof(aggmap_t) * CHO_NUM_BUCKETS);
    for (int b = 0; b < CHO_NUM_BUCKETS; ++b)
        aggmap_init(&a->impls[b], 1);
    aggmap_cell * c = aggmap_first(&a->single);
    if (c && c == &a->single.zero_value)
    {
        /* zero key first: insert(it->getValue()) through the two-level emplace */
        size_t h = cho_intHashCRC32(0);
        int inserted;
        aggmap_cell * d = aggmap_emplace_hashed(&a->impls[cho_two_level_bucket(h)], 0, h, &inserted);
        d->mapped = c->mapped;
        c = aggmap_next_cell(&a->single, c);
    }
    for (; c; c = aggmap_next_cell(&a->single, c))
    {
        size_t h = cho_intHashCRC32(c->key);
        aggmap_insert_unique_non_zero(&a->impls[cho_two_level_bucket(h)], c, h);
    }
    aggmap_destroy(&a->single);
    aggmap_init(&a->single, 0);
    a->is_two_level = 1;
}

int cho_agg_execute_on_block(cho_agg * a, const void * keys, const void * const * args, size_t row_begin, size_t row_end)
{
    if (row_end < row_begin)
        return -1;
    if (a->key_type < 0)
    {
        /* executeWithoutKeyImpl (Aggregator.cpp:1276-1321): addBatchSinglePlace per function */
        for (int j = 0; j < a->n_aggs; ++j)
        {
            char * st = a->without_key + a->offsets[j];
            switch (a->kinds[j])
            {
                case CHO_AGG_COUNT: *(uint64_t *)st += row_end - row_begin; break; /* Count.h:54-70 */
                case CHO_AGG_AVG: *(uint64_t *)(st + 8) += row_end - row_begin;   /* Avg.h:264-285 */
                    __attribute__((fallthrough));
                case CHO_AGG_SUM: cho_sum_add_many(a->arg_types[j], st, args[j], row_begin, row_end); break;
                case CHO_AGG_MIN:
                case CHO_AGG_MAX:
                case CHO_AGG_ANY: /* the default addBatchSinglePlace: add() row by row (IAggregateFunction.h:224-260) */
                    for (size_t i = row_begin; i < row_end; ++i)
                        agg_add_row(a, j, a->without_key, args[j], i);
                    break;
                default: return -1;
            }
        }
        return 0;
    }

    /* executeImplBatch (Aggregator.cpp:1010-1206): resolve places[] then one pass per function */
    char ** places = (char **)malloc(sizeof(char *) * (row_end ? row_end : 1));
    for (size_t i = row_begin; i < row_end; ++i)
    {
        /* HashMethodOneNumber::getKeyHolder (HashMethod.h:91): unalignedLoad<FieldType>, zero-extended
           into the UInt64 table key (AggregatedDataVariants.h:63-64) */
        uint64_t key = load_key_zext(a->key_type, keys, i);
        size_t h = cho_intHashCRC32(key);
        int inserted;
        aggmap_cell * c = aggmap_emplace_hashed(agg_table_for_hash(a, h), key, h, &inserted);
        if (inserted)
        {
            c->mapped = NULL; /* exception-safety step, Aggregator.cpp:1150 */
            char * place = (char *)arena_aligned_alloc(&a->arenas[0], a->total_size_of_aggregate_states, 8);
            memset(place, 0, a->total_size_of_aggregate_states); /* createAggregateStates (:799-830): POD zero states */
            c->mapped = place;
        }
        places[i] = c->mapped;
    }
    /* executeAggregateInstructions (:1208-1273) -> addBatch (IAggregateFunction.h:428-452) */
    for (int j = 0; j < a->n_aggs; ++j)
        for (size_t i = row_begin; i < row_end; ++i)
            if (places[i])
                agg_add_row(a, j, places[i], args ? args[j] : NULL, i);
    free(places);

    /* Aggregator.cpp:1596-1609: convert to two-level when worth it (rows threshold; the bytes threshold
       only changes the moment of conversion, never results) */
    if (!a->is_two_level && a->two_level_threshold && a->single.m_size >= a->two_level_threshold)
        agg_convert_to_two_level(a);
    return 0;
}

/* mergeDataImpl (Aggregator.cpp:2468-2521) via HashMap::mergeToViaEmplace (HashMap.h:203-233) */
static void agg_merge_table(cho_agg * dst_agg, aggmap_t * dst, aggmap_t * src)
{
    for (aggmap_cell * c = aggmap_first(src); c; c = aggmap_next_cell(src, c))
    {
        int inserted;
        aggmap_cell * d = aggmap_emplace_hashed(dst, c->key, cho_intHashCRC32(c->key), &inserted);
        if (inserted)
            d->mapped = c->mapped; /* dst = src (pointer adopted; arena adopted below) */
        else
            agg_merge_states(dst_agg, d->mapped, c->mapped); /* merge + destroy */
        c->mapped = NULL;
    }
}

int cho_agg_merge(cho_agg * dst, cho_agg * src)
{
    if (dst->key_type != src->key_type || dst->n_aggs != src->n_aggs)
        return -1;
    if (dst->key_type < 0)
    {
        agg_merge_states(dst, dst->without_key, src->without_key); /* mergeWithoutKeyDataImpl (:2584-2628) */
        return 0;
    }
    /* prepareVariantsToMerge (:2727-2788): if any is two-level, all become two-level */
    if (dst->is_two_level || src->is_two_level)
    {
        agg_convert_to_two_level(dst);
        agg_convert_to_two_level(src);
        for (int b = 0; b < CHO_NUM_BUCKETS; ++b) /* mergeBucketImpl (:2691-2725) */
            agg_merge_table(dst, &dst->impls[b], &src->impls[b]);
    }
    else
        agg_merge_table(dst, &dst->single, &src->single); /* mergeSingleLevelDataImpl (:2631-2683) */
    /* the destination keeps the source arenas alive (aggregates_pools adoption, :2500-2520) */
    for (int i = 0; i < src->n_arenas && dst->n_arenas < 512; ++i)
    {
        dst->arenas[dst->n_arenas++] = src->arenas[i];
        src->arenas[i].head = NULL;
        src->arenas[i].total = 0;
    }
    return 0;
}

size_t cho_agg_size(const cho_agg * a)
{
    if (a->key_type < 0)
        return 1;
    if (!a->is_two_level)
        return a->single.m_size;
    size_t n = 0;
    for (int b = 0; b < CHO_NUM_BUCKETS; ++b)
        n += a->impls[b].m_size;
    return n;
}

int cho_agg_is_two_level(const cho_agg * a) { return a->is_two_level; }

static void agg_emit_row(const cho_agg * a, uint64_t key, const char * place, size_t row, void * keys_out, void * const * results_out)
{
    if (a->key_type >= 0 && keys_out)
        memcpy((char *)keys_out + row * type_size(a->key_type), &key, type_size(a->key_type)); /* insertKeyIntoColumns: cast back */
    for (int j = 0; j < a->n_aggs; ++j)
    {
        const char * st = place + a->offsets[j];
        char * o = (char *)results_out[j] + row * 8;
        if (a->kinds[j] == CHO_AGG_AVG)
        {
            double r = cho_avg_divide(sum_result_type(a->arg_types[j]), st, *(const uint64_t *)(st + 8));
            memcpy(o, &r, 8);
        }
        else if (a->kinds[j] == CHO_AGG_MIN || a->kinds[j] == CHO_AGG_MAX || a->kinds[j] == CHO_AGG_ANY)
            memcpy(o, st + 8, 8); /* the widened value (a state without value inserts the default 0) */
        else
            memcpy(o, st, 8); /* insertResultInto: sum / count raw 8 bytes */
    }
}

size_t cho_agg_convert_to_block(const cho_agg * ca, void * keys_out, void * const * results_out)
{
    cho_agg * a = (cho_agg *)ca;
    if (a->key_type < 0)
    {
        agg_emit_row(a, 0, a->without_key, 0, NULL, results_out); /* prepareBlockAndFillWithoutKey */
        return 1;
    }
    size_t row = 0;
    /* convertToBlockImplFinal (Aggregator.cpp:2037-2117): data.forEachValue in iteration order */
    if (!a->is_two_level)
    {
        for (aggmap_cell * c = aggmap_first(&a->single); c; c = aggmap_next_cell(&a->single, c))
            agg_emit_row(a, c->key, c->mapped, row++, keys_out, results_out);
    }
    else
    {
        for (int b = 0; b < CHO_NUM_BUCKETS; ++b)
            for (aggmap_cell * c = aggmap_first(&a->impls[b]); c; c = aggmap_next_cell(&a->impls[b], c))
                agg_emit_row(a, c->key, c->mapped, row++, keys_out, results_out);
    }
    return row;
}

/* ------------------------------------------------------------------------------------------------
 * a19/a20  HashJoin
 * ---------------------------------------------------------------------------------------------- */

/* RowRef / RowRefList / Batch (RowRefs.h:16-141) with block pointers replaced by block ids */
typedef struct
{
    int64_t block;
    uint32_t row_num;
} row_ref;

typedef struct rr_batch
{
    uint32_t size;
    struct rr_batch * next;
    row_ref row_refs[7]; /* MAX_SIZE = 7 */
} rr_batch;

typedef struct
{
    int64_t block;     /* RowRef part: first inserted row */
    uint32_t row_num;
    uint32_t rows;     /* RowRefList::rows */
    rr_batch * next;
    uint8_t used;      /* JoinUsedFlags for this cell (flagged maps: INNER ANY) */
} join_mapped;

#define HT_NAME joinmap
#define HT_MAPPED join_mapped
#include "ch_hashtable.inc"

struct cho_join
{
    int kind, strictness, any_take_last_row;
    int maps_all;        /* MapGetter (joinDispatch.h:30-68) */
    int flagged;
    joinmap_t map;
    arena_t pool;
    int64_t n_blocks;
    size_t total_rows;
};

static int jf_need_replication(const cho_join * j) { return j->strictness == CHO_STRICT_ALL; } /* JoinFeatures.h:29 (left/inner only) */
static int jf_need_filter(const cho_join * j)
{
    /* JoinFeatures.h:32 */
    return !jf_need_replication(j)
        && (j->kind == CHO_JOIN_INNER || (j->strictness == CHO_STRICT_SEMI && j->kind == CHO_JOIN_LEFT)
            || (j->strictness == CHO_STRICT_ANTI && j->kind == CHO_JOIN_LEFT));
}
static int jf_add_missing(const cho_join * j) { return j->kind == CHO_JOIN_LEFT && j->strictness != CHO_STRICT_SEMI; } /* :35 */

int cho_join_need_filter(const cho_join * j) { return jf_need_filter(j); }
int cho_join_need_replication(const cho_join * j) { return jf_need_replication(j); }

cho_join * cho_join_create(int kind, int strictness, int any_take_last_row)
{
    if (kind != CHO_JOIN_INNER && kind != CHO_JOIN_LEFT)
        return NULL;
    if (strictness < CHO_STRICT_ANY || strictness > CHO_STRICT_ANTI)
        return NULL;
    if ((strictness == CHO_STRICT_SEMI || strictness == CHO_STRICT_ANTI) && kind != CHO_JOIN_LEFT)
        return NULL; /* only SEMI/ANTI LEFT are valid here (joinDispatch.h:52-64) */
    cho_join * j = (cho_join *)calloc(1, sizeof(*j));
    j->kind = kind;
    j->strictness = strictness;
    j->any_take_last_row = any_take_last_row;
    j->maps_all = strictness == CHO_STRICT_ALL;                                  /* Left/Inner All -> MapsAll */
    j->flagged = (kind == CHO_JOIN_INNER && strictness == CHO_STRICT_ANY);       /* Inner Any -> MapsOne flagged */
    joinmap_init(&j->map, 0);
    return j;
}

void cho_join_free(cho_join * j)
{
    if (!j)
        return;
    joinmap_destroy(&j->map);
    arena_free(&j->pool);
    free(j);
}

/* Batch::insert (RowRefs.h:51-63) */
static rr_batch * batch_insert(rr_batch * b, row_ref ref, arena_t * pool)
{
    if (b->size == 7)
    {
        rr_batch * nb = (rr_batch *)arena_aligned_alloc(pool, sizeof(rr_batch), 8);
        nb->size = 0;
        nb->next = b;
        nb->row_refs[nb->size++] = ref;
        return nb;
    }
    b->row_refs[b->size++] = ref;
    return b;
}

int64_t cho_join_add_block(cho_join * j, const uint64_t * keys, size_t rows, const uint8_t * null_map, const uint8_t * join_mask)
{
    if (rows > 0xFFFFFFFFull)
        return -1; /* HashJoin.cpp:563-564 */
    int64_t block_id = j->n_blocks++;
    j->total_rows += rows;
    /* insertFromBlockImplTypeCase (HashJoinMethodsImpl.h:220-281) */
    for (size_t i = 0; i < rows; ++i)
    {
        if (null_map && null_map[i])
            continue; /* nulls are not inserted (:261-267) */
        if (join_mask && !join_mask[i])
            continue; /* ON-section mask (:270-272) */
        int inserted;
        joinmap_cell * c = joinmap_emplace(&j->map, keys[i], &inserted);
        if (!j->maps_all)
        {
            /* Inserter::insertOne (HashJoinMethods.h:18-28) */
            if (inserted || j->any_take_last_row)
            {
                c->mapped.block = block_id;
                c->mapped.row_num = (uint32_t)i;
                c->mapped.rows = 1;
                c->mapped.next = NULL;
            }
        }
        else
        {
            /* Inserter::insertAll (:30-43) */
            if (inserted)
            {
                c->mapped.block = block_id;
                c->mapped.row_num = (uint32_t)i;
                c->mapped.rows = 1;
                c->mapped.next = NULL;
            }
            else
            {
                /* RowRefList::insert (RowRefs.h:129-138) */
                if (!c->mapped.next)
                {
                    c->mapped.next = (rr_batch *)arena_aligned_alloc(&j->pool, sizeof(rr_batch), 8);
                    c->mapped.next->size = 0;
                    c->mapped.next->next = NULL;
                }
                row_ref ref = {block_id, (uint32_t)i};
                c->mapped.next = batch_insert(c->mapped.next, ref, &j->pool);
                ++c->mapped.rows;
            }
        }
    }
    return block_id;
}

size_t cho_join_total_rows(const cho_join * j) { return j->total_rows; }
size_t cho_join_keys(const cho_join * j) { return j->map.m_size; }

typedef struct
{
    int64_t * block;
    int64_t * row;
    size_t cap, n;
    int overflow;
} added_cols;

static inline void added_push(added_cols * a, int64_t block, int64_t row)
{
    if (a->n < a->cap)
    {
        a->block[a->n] = block;
        a->row[a->n] = row;
    }
    else
        a->overflow = 1;
    ++a->n;
}

size_t cho_join_probe(cho_join * j, const uint64_t * keys, size_t rows, const uint8_t * null_map,
                      size_t max_joined_block_rows, uint8_t * filter, uint64_t * offsets,
                      int64_t * added_block, int64_t * added_row, size_t added_cap, size_t * n_added)
{
    const int need_filter = jf_need_filter(j);
    const int need_replication = jf_need_replication(j);
    const int add_missing = jf_add_missing(j);
    added_cols added = {added_block, added_row, added_cap, 0, 0};

    if (need_filter)
        memset(filter, 0, rows); /* IColumn::Filter(rows, 0), HashJoinMethodsImpl.h:415-416 */
    if (!max_joined_block_rows)
        max_joined_block_rows = (size_t)-1; /* :108-109 */

    uint64_t current_offset = 0;
    size_t i = 0;
    for (; i < rows; ++i) /* joinRightColumns main loop (:429-545) */
    {
        if (need_replication && current_offset >= max_joined_block_rows)
            break; /* :436-444: the tail is returned as not_processed */

        int right_row_found = 0;
        joinmap_cell * c = NULL;
        if (!(null_map && null_map[i])) /* :451-452 */
            c = joinmap_find(&j->map, keys[i]);

        if (c)
        {
            right_row_found = 1;
            join_mapped * mapped = &c->mapped;
            if (j->strictness == CHO_STRICT_ALL)
            {
                /* is_all_join (:480-486) -> addFoundRowAll (KnownRowsHolder.h:88-142), RowRefList::ForwardIterator
                   order (RowRefs.h:66-108): root row, then newest batch 0..size-1, then older batches */
                if (need_filter)
                    filter[i] = 1;
                added_push(&added, mapped->block, mapped->row_num);
                ++current_offset;
                for (rr_batch * b = mapped->next; b; b = b->next)
                    for (uint32_t p = 0; p < b->size; ++p)
                    {
                        added_push(&added, b->row_refs[p].block, b->row_refs[p].row_num);
                        ++current_offset;
                    }
            }
            else if (j->strictness == CHO_STRICT_ANY && j->kind == CHO_JOIN_INNER)
            {
                /* is_any_join && inner (:498-510): each right cell joins only its first left row (setUsedOnce) */
                if (!mapped->used)
                {
                    mapped->used = 1;
                    filter[i] = 1;
                    added_push(&added, mapped->block, mapped->row_num);
                }
            }
            else if (j->strictness == CHO_STRICT_ANTI)
            {
                /* is_anti_join (:515-519): nothing for found rows */
            }
            else
            {
                /* ANY LEFT, SEMI LEFT (:520-530) */
                if (need_filter)
                    filter[i] = 1;
                added_push(&added, mapped->block, mapped->row_num);
            }
        }

        if (!right_row_found)
        {
            if (j->strictness == CHO_STRICT_ANTI && j->kind == CHO_JOIN_LEFT)
                filter[i] = 1; /* :535-536 */
            /* addNotFoundRow<add_missing, need_replication> (KnownRowsHolder.h:144-153) */
            if (add_missing)
            {
                added_push(&added, -1, -1);
                if (need_replication)
                    ++current_offset;
            }
        }

        if (need_replication)
            offsets[i] = current_offset; /* :541-544 */
    }
    *n_added = added.n;
    return i;
}

/* ------------------------------------------------------------------------------------------------
 * a21  ConcurrentHashJoin sharding
 * ---------------------------------------------------------------------------------------------- */

void cho_hash_to_selector(int type, const void * keys, size_t n, size_t num_shards, uint64_t * selector)
{
    /* hashToSelector (ConcurrentHashJoin.cpp:426-440) over calculateHashes (:442-452): the shard maps are
       two-level (HasGetBucketFromHashMemberFunc) -> getBucketFromHash(hash) & (num_shards - 1) */
    for (size_t i = 0; i < n; ++i)
        selector[i] = cho_two_level_bucket(cho_intHashCRC32(load_key_zext(type, keys, i))) & (num_shards - 1);
}

/* ------------------------------------------------------------------------------------------------
 * CPU-baseline drivers for configs C3 / C4 (bench.py's cpu_baseline leg and the parity check on the sample).
 * Same functions as above, driven natively per Block of block_rows rows by N streams like the reference's pipeline:
 *   GROUP BY: one AggregatingTransform (own AggregatedDataVariants) per stream (AggregatingTransform.cpp:664-693), the hash
 *     table cell of row i + look_ahead prefetched while row i is processed (Aggregator.cpp:1025-1054, HashTable.h:957-961,
 *     Prefetching.h:27-52: look-ahead 4..32; 16 here), tables converted to two-level past the threshold, then the 256 buckets
 *     merged by the streams in parallel, each bucket claimed atomically (AggregatingTransform.cpp:120-136 -> mergeBucketImpl,
 *     Aggregator.cpp:2691-2725).
 *   JOIN: HashJoin built by one stream Block by Block (addBlockToJoin), probed by N streams over their own Blocks (joinBlock is
 *     concurrent on an immutable table, IJoin.h:92-93), payload gathered per appended row (fillFromBlocksAndRowNumbers,
 *     IColumn.cpp:515-526) and summed -- the checksum form `SELECT count(), sum(bv)`.
 * ---------------------------------------------------------------------------------------------- */
#include <time.h>
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* executeImplBatch<prefetch = true> for the one-number methods: the same loop as cho_agg_execute_on_block with the look-ahead */
static int agg_execute_on_block_prefetch(cho_agg * a, const void * keys, const void * const * args, size_t row_begin, size_t row_end, char ** places)
{
    const size_t look_ahead = 16;
    for (size_t i = row_begin; i < row_end; ++i)
    {
        if (i + look_ahead < row_end)
        {
            uint64_t pk = load_key_zext(a->key_type, keys, i + look_ahead);
            size_t ph = cho_intHashCRC32(pk);
            aggmap_t * pt = agg_table_for_hash(a, ph);
            __builtin_prefetch(&pt->buf[aggmap_place(pt, ph)]);
        }
        uint64_t key = load_key_zext(a->key_type, keys, i);
        size_t h = cho_intHashCRC32(key);
        int inserted;
        aggmap_cell * c = aggmap_emplace_hashed(agg_table_for_hash(a, h), key, h, &inserted);
        if (inserted)
        {
            c->mapped = NULL;
            char * place = (char *)arena_aligned_alloc(&a->arenas[0], a->total_size_of_aggregate_states, 8);
            memset(place, 0, a->total_size_of_aggregate_states);
            c->mapped = place;
        }
        places[i - row_begin] = c->mapped;
    }
    for (int j = 0; j < a->n_aggs; ++j)
        for (size_t i = row_begin; i < row_end; ++i)
            agg_add_row(a, j, places[i - row_begin], args ? args[j] : NULL, i);
    if (!a->is_two_level && a->two_level_threshold && a->single.m_size >= a->two_level_threshold)
        agg_convert_to_two_level(a);
    return 0;
}

typedef struct
{
    cho_agg * agg;
    const void * keys;
    const void * const * args;
    size_t lo, hi, block_rows;
} gb_stream;

static void * gb_stream_run(void * p)
{
    gb_stream * s = (gb_stream *)p;
    char ** places = (char **)malloc(sizeof(char *) * (s->block_rows ? s->block_rows : 1));
    for (size_t b = s->lo; b < s->hi; b += s->block_rows)
    {
        size_t e = b + s->block_rows < s->hi ? b + s->block_rows : s->hi;
        agg_execute_on_block_prefetch(s->agg, s->keys, s->args, b, e, places);
    }
    free(places);
    return NULL;
}

typedef struct
{
    cho_agg ** aggs;
    int n;
    int * next_bucket; /* shared, claimed with an atomic increment */
} gb_merge;

static void * gb_merge_run(void * p)
{
    gb_merge * m = (gb_merge *)p;
    for (;;)
    {
        int b = __atomic_fetch_add(m->next_bucket, 1, __ATOMIC_RELAXED);
        if (b >= CHO_NUM_BUCKETS)
            break;
        for (int t = 1; t < m->n; ++t)
            agg_merge_table(m->aggs[0], &m->aggs[0]->impls[b], &m->aggs[t]->impls[b]);
    }
    return NULL;
}

/* returns the merged aggregator (caller frees with cho_agg_free); seconds_out[0] = consume, [1] = merge */
cho_agg * cho_groupby_pipeline(int key_type, int n_aggs, const int * kinds, const int * arg_types, const void * keys, const void * const * args,
                               size_t n, size_t block_rows, int threads, uint64_t two_level_threshold, double * seconds_out)
{
    if (threads < 1)
        threads = 1;
    if (threads > 256)
        threads = 256;
    if (!block_rows)
        block_rows = CHO_DEFAULT_BLOCK_SIZE;
    cho_agg ** aggs = (cho_agg **)calloc((size_t)threads, sizeof(cho_agg *));
    gb_stream * st = (gb_stream *)calloc((size_t)threads, sizeof(gb_stream));
    pthread_t * th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    size_t n_blocks = (n + block_rows - 1) / block_rows;
    double t0 = now_s();
    for (int t = 0; t < threads; ++t)
    {
        aggs[t] = cho_agg_create(key_type, n_aggs, kinds, arg_types, two_level_threshold);
        st[t].agg = aggs[t];
        st[t].keys = keys;
        st[t].args = args;
        st[t].block_rows = block_rows;
        size_t b0 = n_blocks * (size_t)t / (size_t)threads, b1 = n_blocks * (size_t)(t + 1) / (size_t)threads;
        st[t].lo = b0 * block_rows;
        st[t].hi = b1 * block_rows < n ? b1 * block_rows : n;
        if (threads > 1)
            pthread_create(&th[t], NULL, gb_stream_run, &st[t]);
        else
            gb_stream_run(&st[t]);
    }
    if (threads > 1)
        for (int t = 0; t < threads; ++t)
            pthread_join(th[t], NULL);
    double t1 = now_s();
    if (threads > 1 && key_type >= 0)
    {
        int any_two_level = 0;
        for (int t = 0; t < threads; ++t)
            any_two_level |= aggs[t]->is_two_level;
        if (any_two_level)
        {
            /* prepareVariantsToMerge: all become two-level, then bucket-parallel merge */
            for (int t = 0; t < threads; ++t)
                agg_convert_to_two_level(aggs[t]);
            int next = 0;
            gb_merge m = {aggs, threads, &next};
            for (int t = 0; t < threads; ++t)
                pthread_create(&th[t], NULL, gb_merge_run, &m);
            for (int t = 0; t < threads; ++t)
                pthread_join(th[t], NULL);
            for (int t = 1; t < threads; ++t)
                for (int i = 0; i < aggs[t]->n_arenas && aggs[0]->n_arenas < 512; ++i)
                {
                    aggs[0]->arenas[aggs[0]->n_arenas++] = aggs[t]->arenas[i];
                    aggs[t]->arenas[i].head = NULL;
                    aggs[t]->arenas[i].total = 0;
                }
        }
        else
            for (int t = 1; t < threads; ++t)
                cho_agg_merge(aggs[0], aggs[t]);
    }
    else
        for (int t = 1; t < threads; ++t)
            cho_agg_merge(aggs[0], aggs[t]);
    double t2 = now_s();
    if (seconds_out)
    {
        seconds_out[0] = t1 - t0;
        seconds_out[1] = t2 - t1;
    }
    cho_agg * res = aggs[0];
    for (int t = 1; t < threads; ++t)
        cho_agg_free(aggs[t]);
    free(aggs);
    free(st);
    free(th);
    return res;
}

typedef struct
{
    cho_join * j;
    const uint64_t * pk;
    const int64_t * bv; /* payload of the build side, all right Blocks back to back */
    size_t lo, hi, block_rows, build_block_rows;
    uint64_t count, sum;
} jp_stream;

static void * jp_stream_run(void * p)
{
    jp_stream * s = (jp_stream *)p;
    size_t cap = s->block_rows * 4 + 16;
    uint64_t * offsets = (uint64_t *)malloc(8 * s->block_rows);
    uint8_t * filter = (uint8_t *)malloc(s->block_rows);
    int64_t * ab = (int64_t *)malloc(8 * cap);
    int64_t * ar = (int64_t *)malloc(8 * cap);
    uint64_t count = 0, sum = 0;
    for (size_t b = s->lo; b < s->hi;)
    {
        size_t rows = b + s->block_rows < s->hi ? s->block_rows : s->hi - b;
        size_t n_added = 0;
        size_t consumed = cho_join_probe(s->j, s->pk + b, rows, NULL, cap / 2, filter, offsets, ab, ar, cap, &n_added);
        if (n_added > cap)
            n_added = cap; /* cannot happen with cap/2 as max_joined_block_rows unless one key has > cap/2 rows */
        for (size_t k = 0; k < n_added; ++k)
            if (ab[k] >= 0)
                sum += (uint64_t)s->bv[(size_t)ab[k] * s->build_block_rows + (size_t)ar[k]];
        count += n_added;
        b += consumed ? consumed : rows;
    }
    s->count = count;
    s->sum = sum;
    free(offsets);
    free(filter);
    free(ab);
    free(ar);
    return NULL;
}

/* `SELECT count(), sum(bv) FROM probe INNER JOIN build ON pk = bk` (strictness ALL); seconds_out[0] = build, [1] = probe */
int cho_join_count_sum_pipeline(const uint64_t * bk, const int64_t * bv, size_t nb, const uint64_t * pk, size_t np, size_t block_rows, int threads,
                                uint64_t * count_out, uint64_t * sum_out, double * seconds_out)
{
    if (threads < 1)
        threads = 1;
    if (threads > 256)
        threads = 256;
    if (!block_rows)
        block_rows = CHO_DEFAULT_BLOCK_SIZE;
    double t0 = now_s();
    cho_join * j = cho_join_create(CHO_JOIN_INNER, CHO_STRICT_ALL, 0);
    if (!j)
        return -1;
    for (size_t b = 0; b < nb; b += block_rows)
        cho_join_add_block(j, bk + b, b + block_rows < nb ? block_rows : nb - b, NULL, NULL);
    double t1 = now_s();
    jp_stream * st = (jp_stream *)calloc((size_t)threads, sizeof(jp_stream));
    pthread_t * th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    size_t n_blocks = (np + block_rows - 1) / block_rows;
    for (int t = 0; t < threads; ++t)
    {
        st[t].j = j;
        st[t].pk = pk;
        st[t].bv = bv;
        st[t].block_rows = block_rows;
        st[t].build_block_rows = block_rows;
        size_t b0 = n_blocks * (size_t)t / (size_t)threads, b1 = n_blocks * (size_t)(t + 1) / (size_t)threads;
        st[t].lo = b0 * block_rows;
        st[t].hi = b1 * block_rows < np ? b1 * block_rows : np;
        if (threads > 1)
            pthread_create(&th[t], NULL, jp_stream_run, &st[t]);
        else
            jp_stream_run(&st[t]);
    }
    uint64_t count = 0, sum = 0;
    for (int t = 0; t < threads; ++t)
    {
        if (threads > 1)
            pthread_join(th[t], NULL);
        count += st[t].count;
        sum += st[t].sum;
    }
    double t2 = now_s();
    *count_out = count;
    *sum_out = sum;
    if (seconds_out)
    {
        seconds_out[0] = t1 - t0;
        seconds_out[1] = t2 - t1;
    }
    free(st);
    free(th);
    cho_join_free(j);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a15 keys128 / keys256: HashMethodKeysFixed (src/Common/ColumnsHashing/HashMethod.h:236-410) over packFixed
 * (src/Interpreters/AggregationCommon.h:91-158) with HashMap<UInt128 / UInt256, ..., UInt128HashCRC32 / UInt256HashCRC32>
 * (AggregatedData.h:57-60; Hash.h:346-355, 412-423).
 * ---------------------------------------------------------------------------------------------- */

/* packFixed: the key columns' raw element bytes laid consecutively into a zero-initialised key of key_bytes */
void cho_pack_fixed(size_t n_cols, const uint32_t * sizes, const void * const * cols, size_t n, size_t key_bytes, uint8_t * out)
{
    memset(out, 0, n * key_bytes);
    for (size_t i = 0; i < n; ++i)
    {
        size_t offset = 0;
        for (size_t j = 0; j < n_cols; ++j)
        {
            memcpy(out + i * key_bytes + offset, (const char *)cols[j] + i * sizes[j], sizes[j]);
            offset += sizes[j];
        }
    }
}

/* UInt128HashCRC32 / UInt256HashCRC32: crc32c chained over the 64-bit words from -1 */
uint64_t cho_hash_keys_fixed(const uint64_t * words, size_t n_words)
{
    uint64_t crc = ~0ull;
    for (size_t w = 0; w < n_words; ++w)
        crc = _mm_crc32_u64(crc, words[w]);
    return crc;
}

/* The map: open addressing, linear probing, power-of-two buffer (initial 256 cells), grows x4 below 2^23 cells then x2 when more
   than half full (HashTableGrowerWithPrecalculation, HashTable.h:273-330); the all-zero key lives out of line (HashTable.h:358-391).
   The mapped value is the dense number of the key by first appearance: what a test needs to express any GROUP BY / join over it. */
struct cho_widemap
{
    size_t key_words, degree, size;
    uint64_t * keys;   /* [cells][key_words]; all-zero = empty */
    uint64_t * mapped; /* [cells] */
    int has_zero;
    uint64_t zero_mapped;
    uint64_t * by_id;  /* [size][key_words] in id order */
    size_t by_id_cap;
};

cho_widemap * cho_widemap_create(size_t key_bytes)
{
    if (key_bytes != 16 && key_bytes != 32)
        return NULL;
    cho_widemap * m = (cho_widemap *)calloc(1, sizeof(*m));
    m->key_words = key_bytes / 8;
    m->degree = 8;
    m->keys = (uint64_t *)calloc((size_t)1 << m->degree, key_bytes);
    m->mapped = (uint64_t *)calloc((size_t)1 << m->degree, 8);
    return m;
}

void cho_widemap_free(cho_widemap * m)
{
    if (!m)
        return;
    free(m->keys);
    free(m->mapped);
    free(m->by_id);
    free(m);
}

size_t cho_widemap_size(const cho_widemap * m) { return m->size; }

static int wide_is_zero(const uint64_t * k, size_t w)
{
    for (size_t q = 0; q < w; ++q)
        if (k[q])
            return 0;
    return 1;
}

static size_t wide_find_cell(const cho_widemap * m, const uint64_t * key, uint64_t hash)
{
    const size_t mask = ((size_t)1 << m->degree) - 1, w = m->key_words;
    size_t place = hash & mask;
    while (!wide_is_zero(m->keys + place * w, w) && memcmp(m->keys + place * w, key, w * 8) != 0)
        place = (place + 1) & mask;
    return place;
}

static void wide_grow(cho_widemap * m)
{
    const size_t old_cells = (size_t)1 << m->degree, w = m->key_words;
    uint64_t * ok = m->keys, * om = m->mapped;
    m->degree += m->degree >= 23 ? 1 : 2;
    m->keys = (uint64_t *)calloc((size_t)1 << m->degree, w * 8);
    m->mapped = (uint64_t *)calloc((size_t)1 << m->degree, 8);
    for (size_t c = 0; c < old_cells; ++c)
        if (!wide_is_zero(ok + c * w, w))
        {
            size_t place = wide_find_cell(m, ok + c * w, cho_hash_keys_fixed(ok + c * w, w));
            memcpy(m->keys + place * w, ok + c * w, w * 8);
            m->mapped[place] = om[c];
        }
    free(ok);
    free(om);
}

/* emplaceKey (insert != 0) / findKey for n packed keys; ids_out[i] = dense id by first appearance, ~0 when absent (find) */
void cho_widemap_batch(cho_widemap * m, const uint8_t * packed, size_t n, int insert, uint64_t * ids_out)
{
    const size_t w = m->key_words;
    for (size_t i = 0; i < n; ++i)
    {
        uint64_t key[4];
        memcpy(key, packed + i * w * 8, w * 8);
        if (wide_is_zero(key, w))
        {
            if (!m->has_zero && insert)
            {
                m->has_zero = 1;
                m->zero_mapped = m->size++;
                goto record_new;
            }
            ids_out[i] = m->has_zero ? m->zero_mapped : ~0ull;
            continue;
        }
        {
            size_t place = wide_find_cell(m, key, cho_hash_keys_fixed(key, w));
            if (!wide_is_zero(m->keys + place * w, w))
            {
                ids_out[i] = m->mapped[place];
                continue;
            }
            if (!insert)
            {
                ids_out[i] = ~0ull;
                continue;
            }
            memcpy(m->keys + place * w, key, w * 8);
            m->mapped[place] = m->size++;
        }
    record_new:
        if (m->size > m->by_id_cap)
        {
            m->by_id_cap = m->by_id_cap ? m->by_id_cap * 2 : 1024;
            m->by_id = (uint64_t *)realloc(m->by_id, m->by_id_cap * w * 8);
        }
        memcpy(m->by_id + (m->size - 1) * w, key, w * 8);
        ids_out[i] = m->size - 1;
        if (m->size - (m->has_zero ? 1 : 0) > ((size_t)1 << (m->degree - 1)))
            wide_grow(m);
    }
}

/* the packed keys in id order: out[size][key_bytes] */
void cho_widemap_keys(const cho_widemap * m, uint8_t * out) { memcpy(out, m->by_id, m->size * m->key_words * 8); }
