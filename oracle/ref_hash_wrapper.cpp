/*
 * ref_hash_wrapper.cpp — thin extern "C" exports over the REFERENCE's own src/Common/HashTable/Hash.h,
 * compiled in place from /root/reference (never copied) into oracle/_ref/libchref_hash.so by oracle/Makefile.
 * TEST INFRASTRUCTURE ONLY: used to pin oracle/ch_oracle.c's hash restatement against the real thing.
 * Hash.h is the only file of the hot path that compiles without the reference's absent submodules
 * (SURVEY.md §8c); everything else is pinned by the reference's golden test outputs.
 */
#include <Common/HashTable/Hash.h>

extern "C" {
unsigned long long ref_intHash64(unsigned long long x) { return intHash64(x); }
unsigned long long ref_intHashCRC32(unsigned long long x) { return intHashCRC32(x); }
unsigned long long ref_intHashCRC32_seed(unsigned long long x, unsigned long long seed) { return intHashCRC32(x, seed); }
unsigned int ref_intHash32_salt0(unsigned long long x) { return intHash32<0>(x); }
unsigned int ref_intHash32_sql(unsigned long long x) { return intHash32<0x75D9543DE018BF45ULL>(x); }
unsigned long long ref_HashCRC32_UInt64(unsigned long long x) { return HashCRC32<UInt64>()(x); }
unsigned long long ref_HashCRC32_UInt32(unsigned int x) { return HashCRC32<UInt32>()(x); }
unsigned long long ref_HashCRC32_Int64(long long x) { return HashCRC32<Int64>()(x); }
unsigned long long ref_hashCRC32_UInt64_seed(unsigned long long x, unsigned long long seed) { return hashCRC32<UInt64>(x, seed); }
unsigned long long ref_hashCRC32_UInt32_seed(unsigned int x, unsigned long long seed) { return hashCRC32<UInt32>(x, seed); }
void ref_intHashCRC32_batch(const unsigned long long * keys, unsigned long long n, unsigned long long * out)
{
    for (unsigned long long i = 0; i < n; ++i)
        out[i] = intHashCRC32(keys[i]);
}
void ref_intHash64_batch(const unsigned long long * keys, unsigned long long n, unsigned long long * out)
{
    for (unsigned long long i = 0; i < n; ++i)
        out[i] = intHash64(keys[i]);
}
}

/* keys128 / keys256: the hashes of the wide-key maps (Hash.h:346-355, 412-423) */
extern "C" {
unsigned long long ref_UInt128HashCRC32(unsigned long long w0, unsigned long long w1)
{
    UInt128 x;
    x.items[0] = w0;
    x.items[1] = w1;
    return UInt128HashCRC32()(x);
}
unsigned long long ref_UInt256HashCRC32(unsigned long long w0, unsigned long long w1, unsigned long long w2, unsigned long long w3)
{
    UInt256 x;
    x.items[0] = w0;
    x.items[1] = w1;
    x.items[2] = w2;
    x.items[3] = w3;
    return UInt256HashCRC32()(x);
}
}
