// feed.hip — the wire formats on either side of the hot path: compressed-frame checksums, the Native block header walk and the
// serialized bytes of aggregate-function states.
//
// Reference (file:line in the reference checkout):
//   frame checksum    CompressedReadBufferBase::readCompressedData / validateChecksum: CityHash128 (v1.0.2) over the frame's header and
//                     payload, stored in front of it as {low64, high64}; sizes above DBMS_MAX_COMPRESSED_SIZE (1 GiB) are refused
//                     (src/Compression/CompressedReadBufferBase.cpp:49-127,130-222; src/Compression/CompressionInfo.h:10-51)
//   Native format     NativeReader::read (src/Formats/NativeReader.cpp:113-260), BlockInfo::read (src/Core/BlockInfo.cpp:38-62)
//   state bytes       AggregateFunctionSum::serialize (AggregateFunctionSum.h:288-291: the 8-byte sum), AggregateFunctionCount::serialize
//                     (AggregateFunctionCount.h:126-129: VarUInt), AggregateFunctionAvg (AvgFraction: numerator, VarUInt denominator);
//                     a ColumnAggregateFunction is its rows' states one after the other (SerializationAggregateFunction.cpp)
// CityHash128 below is written from the algorithm as Google published it (cityhash 1.0.2: the version ClickHouse froze, because its
// checksums are on disk), not taken from the reference's contrib/ copy; oracle/ref_city_wrapper.cpp compiles that copy in place to pin it.
#include "chgpu_internal.h"
#include <dlfcn.h>

#include <string>
#include <utility>
#include <vector>

// ---------------------------------------------------------------------------------------------
// CityHash128, version 1.0.2
// ---------------------------------------------------------------------------------------------
namespace city102
{
struct U128
{
    u64 lo, hi;
};
static constexpr u64 K0 = 0xc3a5c85c97cb3127ull, K1 = 0xb492b66fbe98f273ull, K2 = 0x9ae16a3b2f90404full, K3 = 0xc949d7c7509e6557ull;
static inline u64 ld64(const u8 * p)
{
    u64 v;
    memcpy(&v, p, 8);
    return v;
}
static inline u64 ld32(const u8 * p)
{
    u32 v;
    memcpy(&v, p, 4);
    return v;
}
static inline u64 ror(u64 v, int s) { return s == 0 ? v : (v >> s) | (v << (64 - s)); }
static inline u64 smix(u64 v) { return v ^ (v >> 47); }
static inline u64 fold(u64 lo, u64 hi) // Hash128to64
{
    const u64 m = 0x9ddfea08eb382d69ull;
    u64 a = (lo ^ hi) * m;
    a ^= a >> 47;
    u64 b = (hi ^ a) * m;
    b ^= b >> 47;
    return b * m;
}
static u64 short_hash(const u8 * s, size_t len) // HashLen0to16
{
    if (len > 8)
    {
        const u64 a = ld64(s), b = ld64(s + len - 8);
        return fold(a, ror(b + len, (int)len)) ^ b;
    }
    if (len >= 4)
        return fold(len + (ld32(s) << 3), ld32(s + len - 4));
    if (len > 0)
    {
        const u32 y = (u32)s[0] + ((u32)s[len >> 1] << 8), z = (u32)len + ((u32)s[len - 1] << 2);
        return smix(y * K2 ^ z * K3) * K2;
    }
    return K2;
}
static inline U128 weak32(u64 w, u64 x, u64 y, u64 z, u64 a, u64 b) // WeakHashLen32WithSeeds
{
    a += w;
    b = ror(b + a + z, 21);
    const u64 c = a;
    a += x;
    a += y;
    b += ror(a, 44);
    return {a + z, b + c};
}
static inline U128 weak32(const u8 * s, u64 a, u64 b) { return weak32(ld64(s), ld64(s + 8), ld64(s + 16), ld64(s + 24), a, b); }

static U128 murmur(const u8 * s, size_t len, U128 seed) // CityMurmur: inputs shorter than 128 bytes
{
    u64 a = seed.lo, b = seed.hi, c = 0, d = 0;
    long l = (long)len - 16;
    if (l <= 0)
    {
        a = smix(a * K1) * K1;
        c = b * K1 + short_hash(s, len);
        d = smix(a + (len >= 8 ? ld64(s) : c));
    }
    else
    {
        c = fold(ld64(s + len - 8) + K1, a);
        d = fold(b + len, c + ld64(s + len - 16));
        a += d;
        do
        {
            a ^= smix(ld64(s) * K1) * K1;
            a *= K1;
            b ^= a;
            c ^= smix(ld64(s + 8) * K1) * K1;
            c *= K1;
            d ^= c;
            s += 16;
            l -= 16;
        } while (l > 0);
    }
    a = fold(a, c);
    b = fold(d, b);
    return {a ^ b, fold(b, a)};
}

static U128 with_seed(const u8 * s, size_t len, U128 seed)
{
    if (len < 128)
        return murmur(s, len, seed);
    U128 v, w;
    u64 x = seed.lo, y = seed.hi, z = len * K1;
    v.lo = ror(y ^ K1, 49) * K1 + ld64(s);
    v.hi = ror(v.lo, 42) * K1 + ld64(s + 8);
    w.lo = ror(y + z, 35) * K1 + x;
    w.hi = ror(x + ld64(s + 88), 53) * K1;
    do // 128 bytes per iteration, as two identical 64-byte rounds
    {
        for (int half = 0; half < 2; ++half)
        {
            x = ror(x + y + v.lo + ld64(s + 16), 37) * K1;
            y = ror(y + v.hi + ld64(s + 48), 42) * K1;
            x ^= w.hi;
            y ^= v.lo;
            z = ror(z ^ w.lo, 33);
            v = weak32(s, v.hi * K1, x + w.lo);
            w = weak32(s + 32, z + w.hi, y);
            std::swap(z, x);
            s += 64;
        }
        len -= 128;
    } while (len >= 128);
    y += ror(w.lo, 37) * K0 + z;
    x += ror(v.lo + z, 49) * K0;
    for (size_t done = 0; done < len;) // the last < 128 bytes, 32 at a time from the end
    {
        done += 32;
        y = ror(y - x, 42) * K0 + v.hi;
        w.lo += ld64(s + len - done + 16);
        x = ror(x, 49) * K0 + w.lo;
        w.lo += v.lo;
        v = weak32(s + len - done, v.lo, v.hi);
    }
    x = fold(x, v.lo);
    y = fold(y, w.lo);
    return {fold(x + v.hi, w.hi) + y, fold(x + w.hi, y + v.hi)};
}

static U128 hash128(const u8 * s, size_t len)
{
    if (len >= 16)
        return with_seed(s + 16, len - 16, {ld64(s) ^ K3, ld64(s + 8)});
    if (len >= 8)
        return with_seed(nullptr, 0, {ld64(s) ^ (len * K0), ld64(s + len - 8) ^ K1});
    return with_seed(s, len, {K0, K1});
}
} // namespace city102

extern "C" int chgpu_city_hash128(const void * data, uint64_t size, uint64_t out_low_high[2])
{
    CHGPU_REQUIRE((data || size == 0) && out_low_high, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    const city102::U128 h = city102::hash128((const u8 *)data, size);
    out_low_high[0] = h.lo;
    out_low_high[1] = h.hi;
    return CHGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// compressed frames
// ---------------------------------------------------------------------------------------------
static constexpr size_t CKSUM = 16, HDR = 9;
static constexpr u32 MAX_COMPRESSED_SIZE = 0x40000000u; // DBMS_MAX_COMPRESSED_SIZE: 1 GiB

static inline u32 rd_u32(const u8 * p)
{
    u32 v;
    memcpy(&v, p, 4);
    return v;
}

extern "C" int chgpu_compressed_walk_frames(const uint8_t * file, uint64_t size, int verify_checksums, uint32_t capacity, uint32_t * n_frames, uint64_t * payload_offsets,
                                            uint32_t * payload_sizes, uint32_t * decompressed_sizes, uint8_t * methods, uint8_t * post_methods, uint32_t * stage_sizes)
{
    CHGPU_REQUIRE((file || size == 0) && n_frames, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(capacity == 0 || (payload_offsets && payload_sizes && decompressed_sizes && methods && post_methods && stage_sizes), CHGPU_ERR_BAD_ARGUMENTS, "NULL frame arrays");
    u32 n = 0;
    for (u64 pos = 0; pos < size;)
    {
        CHGPU_REQUIRE(size - pos >= CKSUM + HDR, CHGPU_ERR_BAD_ARGUMENTS, "Cannot read all data: truncated frame header at byte %llu", (unsigned long long)pos);
        u8 method = file[pos + CKSUM];
        const u32 csize = rd_u32(file + pos + CKSUM + 1), dsize = rd_u32(file + pos + CKSUM + 5);
        // CompressedReadBufferBase.cpp:163-172: a corrupted header must not drive a multi-gigabyte allocation
        CHGPU_REQUIRE(csize <= MAX_COMPRESSED_SIZE, CHGPU_ERR_BAD_ARGUMENTS, "Too large size_compressed_without_checksum: %u. Most likely corrupted data. (TOO_LARGE_SIZE_COMPRESSED)", csize);
        CHGPU_REQUIRE(dsize <= MAX_COMPRESSED_SIZE, CHGPU_ERR_BAD_ARGUMENTS, "Too large size_decompressed: %u. Most likely corrupted data. (TOO_LARGE_SIZE_COMPRESSED)", dsize);
        CHGPU_REQUIRE(csize >= HDR && pos + CKSUM + csize <= size, CHGPU_ERR_BAD_ARGUMENTS, "Cannot decompress: frame size out of range");
        if (verify_checksums)
        {
            const city102::U128 h = city102::hash128(file + pos + CKSUM, csize);
            u64 want[2];
            memcpy(want, file + pos, 16);
            CHGPU_REQUIRE(h.lo == want[0] && h.hi == want[1], CHGPU_ERR_BAD_ARGUMENTS,
                          "Checksum doesn't match: corrupted data. Reference: %016llx%016llx. Actual: %016llx%016llx. Size of compressed block: %u (CHECKSUM_DOESNT_MATCH)",
                          (unsigned long long)want[1], (unsigned long long)want[0], (unsigned long long)h.hi, (unsigned long long)h.lo, csize);
        }
        u64 off = pos + CKSUM + HDR;
        u32 sz = csize - (u32)HDR, stage = dsize;
        u8 post = 0;
        // CODEC(Delta, LZ4), CODEC(DoubleDelta, ZSTD), CODEC(T64, LZ4), CODEC(Gorilla, LZ4) ...: Multiple (0x91) = [n methods][method bytes][the
        // last stage's own header + payload] (CompressionCodecMultiple.cpp:68-130); the pairs {a column codec, a general-purpose codec}
        if (method == 0x91 && sz >= 3 + HDR && file[off] == 2 && file[off + 1] >= 0x92 && file[off + 1] <= 0x95 && (file[off + 2] == 0x82 || file[off + 2] == 0x90 || file[off + 2] == 0x02))
        {
            const u8 codec = file[off + 1], general = file[off + 2];
            const u32 c2 = rd_u32(file + off + 3 + 1), d2 = rd_u32(file + off + 3 + 5);
            CHGPU_REQUIRE(file[off + 3] == general && c2 >= HDR && 3 + (u64)c2 <= sz && d2 <= MAX_COMPRESSED_SIZE, CHGPU_ERR_BAD_ARGUMENTS,
                          "Cannot decompress: bad stage header in codec Multiple");
            method = general, post = codec, stage = d2;
            off += 3 + HDR, sz = c2 - (u32)HDR;
        }
        if (n < capacity)
        {
            payload_offsets[n] = off;
            payload_sizes[n] = sz;
            decompressed_sizes[n] = dsize;
            methods[n] = method;
            post_methods[n] = post;
            stage_sizes[n] = stage;
        }
        ++n;
        pos += CKSUM + csize;
    }
    *n_frames = n;
    return CHGPU_OK;
}

namespace
{
struct ZstdLib
{
    size_t (*decompress)(void *, size_t, const void *, size_t);
    unsigned (*is_error)(size_t);
    const char * (*error_name)(size_t);
};
// libzstd of the host (the reference links the same library: contrib/zstd), loaded once; nullptr when the host has none
const ZstdLib * zstd_lib()
{
    static ZstdLib lib;
    static const bool ok = [] {
        void * h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h)
            h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
        if (!h)
            return false;
        lib.decompress = (size_t(*)(void *, size_t, const void *, size_t))dlsym(h, "ZSTD_decompress");
        lib.is_error = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
        lib.error_name = (const char * (*)(size_t))dlsym(h, "ZSTD_getErrorName");
        return lib.decompress && lib.is_error && lib.error_name;
    }();
    return ok ? &lib : nullptr;
}
} // namespace

extern "C" int chgpu_read_compressed_column(chgpu_ctx * ctx, const uint8_t * file, uint64_t size, int type, int verify_checksums, chgpu_col ** out)
{
    CHGPU_REQUIRE(ctx && (file || size == 0) && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    const size_t es = chgpu_type_size(type);
    CHGPU_REQUIRE(es, CHGPU_ERR_BAD_ARGUMENTS, "unknown column type %d", type);
    ChgpuDeviceGuard guard(ctx);
    u32 n = 0;
    CHGPU_TRY(chgpu_compressed_walk_frames(file, size, verify_checksums, 0, &n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)); // validates, counts
    std::vector<u64> offs(n ? n : 1);
    std::vector<u32> sizes(n ? n : 1), dsizes(n ? n : 1), stages(n ? n : 1);
    std::vector<u8> methods(n ? n : 1), posts(n ? n : 1);
    CHGPU_TRY(chgpu_compressed_walk_frames(file, size, 0, n, &n, offs.data(), sizes.data(), dsizes.data(), methods.data(), posts.data(), stages.data()));
    u64 total = 0;
    for (u32 f = 0; f < n; ++f)
        total += dsizes[f];
    CHGPU_REQUIRE(total % es == 0, CHGPU_ERR_SIZES_MISMATCH, "Cannot read all data: %llu decompressed bytes are not a multiple of the element size %zu",
                  (unsigned long long)total, es);
    // ZSTD (0x90, CompressionCodecZSTD.cpp:60-66: the payload is one zstd frame) is undone on the HOST by the library the reference links --
    // libzstd, loaded on first use -- into an image of the file in which those frames are stored ones (method NONE); what crosses PCIe is
    // then the decompressed bytes of these frames and the compressed bytes of all others.  A column codec behind it (CODEC(DoubleDelta,
    // ZSTD)) still runs on the device.
    std::vector<u8> image;
    bool any_zstd = false;
    for (u32 f = 0; f < n; ++f)
        any_zstd = any_zstd || methods[f] == 0x90;
    if (any_zstd)
    {
        const ZstdLib * z = zstd_lib();
        CHGPU_REQUIRE(z, CHGPU_ERR_NOT_IMPLEMENTED, "compression method 0x90 (ZSTD) needs libzstd.so.1 on the host: CPU path");
        u64 image_bytes = 0;
        for (u32 f = 0; f < n; ++f)
            image_bytes += methods[f] == 0x90 ? stages[f] : sizes[f];
        image.resize(image_bytes ? image_bytes : 1);
        u64 at = 0;
        for (u32 f = 0; f < n; ++f)
        {
            if (methods[f] == 0x90)
            {
                const size_t got = z->decompress(image.data() + at, stages[f], file + offs[f], sizes[f]);
                CHGPU_REQUIRE(!z->is_error(got) && got == stages[f], CHGPU_ERR_BAD_ARGUMENTS, "Cannot decompress ZSTD-encoded data: %s (CANNOT_DECOMPRESS)",
                              z->is_error(got) ? z->error_name(got) : "wrong decompressed size");
                methods[f] = 0x02;
                offs[f] = at;
                sizes[f] = stages[f];
            }
            else
            {
                memcpy(image.data() + at, file + offs[f], sizes[f]);
                offs[f] = at;
            }
            at += sizes[f];
        }
        file = image.data();
        size = image_bytes;
    }
    chgpu_col * compressed = nullptr;
    CHGPU_TRY(chgpu_col_upload(ctx, CHGPU_U8, file, size, &compressed));
    chgpu_col * raw = nullptr;
    int rc = chgpu_decompress_frames(ctx, compressed, n, offs.data(), sizes.data(), dsizes.data(), methods.data(), posts.data(), stages.data(), &raw);
    chgpu_col_free(compressed);
    if (rc != CHGPU_OK)
        return rc;
    rc = chgpu_col_from_bytes(ctx, raw, 0, type, total / es, out);
    chgpu_col_free(raw);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// Native block format
// ---------------------------------------------------------------------------------------------
namespace
{
struct Reader
{
    const u8 * p;
    const u8 * end;
    bool ok = true;
    u64 varuint()
    {
        u64 x = 0;
        for (int i = 0; i < 10; ++i) // readVarUInt (src/IO/VarInt.h): 7 bits per byte, least significant first
        {
            if (p >= end)
            {
                ok = false;
                return 0;
            }
            const u8 b = *p++;
            x |= (u64)(b & 0x7F) << (7 * i);
            if (!(b & 0x80))
                return x;
        }
        ok = false;
        return x;
    }
    bool bytes(void * dst, size_t n)
    {
        if ((size_t)(end - p) < n)
        {
            ok = false;
            return false;
        }
        memcpy(dst, p, n);
        p += n;
        return true;
    }
};

// the element type of a numeric DataType name (DataTypeFactory would resolve it); -1 = not a type this path carries
int native_type_tag(const std::string & name)
{
    static const std::pair<const char *, int> names[] = {
        {"UInt8", CHGPU_U8}, {"UInt16", CHGPU_U16}, {"UInt32", CHGPU_U32}, {"UInt64", CHGPU_U64}, {"Int8", CHGPU_I8}, {"Int16", CHGPU_I16},
        {"Int32", CHGPU_I32}, {"Int64", CHGPU_I64}, {"Float32", CHGPU_F32}, {"Float64", CHGPU_F64}, {"Date", CHGPU_U16}, {"DateTime", CHGPU_U32},
        {"Bool", CHGPU_U8}, {"Date32", CHGPU_I32}, {"IPv4", CHGPU_U32}};
    for (auto & kv : names)
        if (name == kv.first)
            return kv.second;
    return -1;
}
} // namespace

extern "C" int chgpu_native_walk_block(const uint8_t * data, uint64_t size, uint64_t server_revision, uint32_t capacity, chgpu_native_column * columns, uint32_t * n_columns,
                                       uint64_t * n_rows, int32_t * bucket_num, int * is_overflows, uint64_t * bytes_consumed)
{
    CHGPU_REQUIRE((data || size == 0) && n_columns && n_rows && bytes_consumed, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    Reader r{data, data + size};
    i32 bucket = -1;
    u8 overflows = 0;
    if (server_revision > 0)
    {
        // BlockInfo::read: (field number, value)* 0
        for (;;)
        {
            const u64 field = r.varuint();
            CHGPU_REQUIRE(r.ok, CHGPU_ERR_BAD_ARGUMENTS, "Cannot read all data: truncated BlockInfo");
            if (field == 0)
                break;
            if (field == 1)
                r.bytes(&overflows, 1);
            else if (field == 2)
                r.bytes(&bucket, 4);
            else
                return chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "Unknown BlockInfo field number: %llu (UNKNOWN_BLOCK_INFO_FIELD)", (unsigned long long)field);
            CHGPU_REQUIRE(r.ok, CHGPU_ERR_BAD_ARGUMENTS, "Cannot read all data: truncated BlockInfo");
        }
    }
    const u64 cols = r.varuint(), rows = r.varuint();
    CHGPU_REQUIRE(r.ok, CHGPU_ERR_BAD_ARGUMENTS, "Cannot read all data: truncated block dimensions");
    CHGPU_REQUIRE(cols <= 1000000ull, CHGPU_ERR_BAD_ARGUMENTS, "Suspiciously many columns in Native format: %llu (TOO_LARGE_ARRAY_SIZE)", (unsigned long long)cols);
    CHGPU_REQUIRE(rows <= 1000000000000ull, CHGPU_ERR_BAD_ARGUMENTS, "Suspiciously many rows in Native format: %llu (TOO_LARGE_ARRAY_SIZE)", (unsigned long long)rows);
    CHGPU_REQUIRE(!(cols == 0 && rows != 0), CHGPU_ERR_BAD_ARGUMENTS, "Zero columns but %llu rows in Native format. (INCORRECT_DATA)", (unsigned long long)rows);
    for (u64 c = 0; c < cols; ++c)
    {
        std::string name, type;
        for (std::string * s : {&name, &type})
        {
            const u64 len = r.varuint();
            CHGPU_REQUIRE(r.ok && len <= (u64)(r.end - r.p), CHGPU_ERR_BAD_ARGUMENTS, "Cannot read all data: truncated column header");
            s->assign((const char *)r.p, len);
            r.p += len;
        }
        if (server_revision >= 54454) // DBMS_MIN_REVISION_WITH_CUSTOM_SERIALIZATION: a flag byte; custom (sparse) kinds are not carried
        {
            u8 has_custom = 0;
            r.bytes(&has_custom, 1);
            CHGPU_REQUIRE(r.ok, CHGPU_ERR_BAD_ARGUMENTS, "Cannot read all data: truncated column header");
            CHGPU_REQUIRE(!has_custom, CHGPU_ERR_NOT_IMPLEMENTED, "column %s has a custom (sparse) serialization: CPU path", name.c_str());
        }
        chgpu_native_column o;
        memset(&o, 0, sizeof(o));
        snprintf(o.name, sizeof(o.name), "%s", name.c_str());
        snprintf(o.type_name, sizeof(o.type_name), "%s", type.c_str());
        // the type name, outside in: LowCardinality(...) and Nullable(...) wrap String / FixedString(N) / a number
        std::string inner = type;
        bool lc = false;
        auto unwrap = [&](const char * prefix) {
            const size_t n = strlen(prefix);
            if (inner.size() > n + 1 && inner.compare(0, n, prefix) == 0 && inner.back() == ')')
            {
                inner = inner.substr(n, inner.size() - n - 1);
                return true;
            }
            return false;
        };
        lc = unwrap("LowCardinality(");
        o.is_nullable = unwrap("Nullable(") ? 1 : 0;
        int tag = native_type_tag(inner);
        if (inner == "String")
            o.kind = CHGPU_NATIVE_STRING, tag = CHGPU_U8;
        else if (unwrap("FixedString("))
        {
            char * endp = nullptr;
            const unsigned long long fn = strtoull(inner.c_str(), &endp, 10);
            CHGPU_REQUIRE(endp && *endp == 0 && fn >= 1 && fn <= 0xFFFFFFull, CHGPU_ERR_BAD_ARGUMENTS, "column %s: bad type %s", name.c_str(), type.c_str());
            o.kind = CHGPU_NATIVE_FIXED_STRING, o.fixed_n = (u32)fn, tag = CHGPU_U8;
        }
        CHGPU_REQUIRE(tag >= 0, CHGPU_ERR_NOT_IMPLEMENTED, "column %s has type %s: numbers, String, FixedString and their Nullable / LowCardinality(String) forms are read on this path (CPU path)",
                      name.c_str(), type.c_str());
        CHGPU_REQUIRE(!lc || o.kind == CHGPU_NATIVE_STRING, CHGPU_ERR_NOT_IMPLEMENTED, "column %s has type %s: LowCardinality of String only (CPU path)", name.c_str(), type.c_str());
        const char * short_msg = "Cannot read all data in NativeReader. Rows expected: %llu (CANNOT_READ_ALL_DATA)";
        // SerializationString::deserializeBinaryBulk: (VarUInt length, bytes) per value -- the walk finds the end and counts the bytes
        auto walk_strings = [&](u64 n_values, u64 * chars_bytes) -> bool {
            u64 total = 0;
            for (u64 i = 0; i < n_values; ++i)
            {
                const u64 len = r.varuint();
                if (!r.ok || len > (u64)(r.end - r.p))
                    return false;
                r.p += len;
                total += len;
            }
            *chars_bytes = total;
            return true;
        };
        if (rows) // (NativeReader.cpp:240-245: no rows, nothing to read -- not even the state prefix)
        {
            if (lc)
            {
                // SerializationLowCardinality: the state prefix (keys version), then [index type + flags][additional keys][rows][indexes]
                // (SerializationLowCardinality.cpp:84-164, :560-700; the byte layout 02010_lc_native.python writes by hand)
                u64 version = 0, itype = 0, num_keys = 0, num_rows = 0;
                CHGPU_REQUIRE(r.bytes(&version, 8), CHGPU_ERR_BAD_ARGUMENTS, short_msg, (unsigned long long)rows);
                CHGPU_REQUIRE(version == 1, CHGPU_ERR_BAD_ARGUMENTS, "Invalid version for SerializationLowCardinality key column. (INCORRECT_DATA)");
                CHGPU_REQUIRE(r.bytes(&itype, 8), CHGPU_ERR_BAD_ARGUMENTS, short_msg, (unsigned long long)rows);
                const u64 flags = itype & 0x700ull, width_code = itype & ~0x700ull;
                CHGPU_REQUIRE(width_code <= 3, CHGPU_ERR_BAD_ARGUMENTS, "Invalid type for SerializationLowCardinality index column. (INCORRECT_DATA)");
                CHGPU_REQUIRE(!(flags & 0x100), CHGPU_ERR_BAD_ARGUMENTS, "LowCardinality indexes serialization type for Native format cannot use global dictionary (INCORRECT_DATA)");
                CHGPU_REQUIRE(flags & 0x200, CHGPU_ERR_BAD_ARGUMENTS, "No additional keys found. (INCORRECT_DATA)");
                CHGPU_REQUIRE(r.bytes(&num_keys, 8), CHGPU_ERR_BAD_ARGUMENTS, short_msg, (unsigned long long)rows);
                o.kind = CHGPU_NATIVE_LC_STRING;
                o.lc_num_keys = num_keys;
                o.lc_keys_offset = (u64)(r.p - data);
                CHGPU_REQUIRE(num_keys <= (u64)(r.end - r.p) && walk_strings(num_keys, &o.lc_keys_chars_bytes), CHGPU_ERR_BAD_ARGUMENTS, short_msg, (unsigned long long)rows);
                o.lc_keys_bytes = (u64)(r.p - data) - o.lc_keys_offset;
                CHGPU_REQUIRE(r.bytes(&num_rows, 8), CHGPU_ERR_BAD_ARGUMENTS, short_msg, (unsigned long long)rows);
                CHGPU_REQUIRE(num_rows == rows, CHGPU_ERR_BAD_ARGUMENTS, "Cannot read all data in NativeReader. Rows read: %llu. Rows expected: %llu (CANNOT_READ_ALL_DATA)",
                              (unsigned long long)num_rows, (unsigned long long)rows);
                static const int index_tags[4] = {CHGPU_U8, CHGPU_U16, CHGPU_U32, CHGPU_U64};
                tag = index_tags[width_code];
                const u64 w = 1ull << width_code, nbytes = rows * w;
                CHGPU_REQUIRE(nbytes <= (u64)(r.end - r.p), CHGPU_ERR_BAD_ARGUMENTS, short_msg, (unsigned long long)rows);
                // ColumnLowCardinality::Index::checkSizeOfType / insertRangeFromDictionaryEncodedColumn (ColumnLowCardinality.cpp:240-252)
                for (u64 i = 0; i < rows; ++i)
                {
                    u64 ix = 0;
                    memcpy(&ix, r.p + i * w, w);
                    CHGPU_REQUIRE(ix < num_keys, CHGPU_ERR_BAD_ARGUMENTS, "Index for LowCardinality is out of range. Dictionary size is %llu, but found index with value %llu (INCORRECT_DATA)",
                                  (unsigned long long)num_keys, (unsigned long long)ix);
                }
                o.data_offset = (u64)(r.p - data);
                o.data_bytes = nbytes;
                r.p += nbytes;
            }
            else
            {
                if (o.is_nullable) // SerializationNullable: the null map (one byte per row), then the nested column
                {
                    CHGPU_REQUIRE(rows <= (u64)(r.end - r.p), CHGPU_ERR_BAD_ARGUMENTS, short_msg, (unsigned long long)rows);
                    o.null_map_offset = (u64)(r.p - data);
                    r.p += rows;
                }
                o.data_offset = (u64)(r.p - data);
                if (o.kind == CHGPU_NATIVE_STRING)
                {
                    CHGPU_REQUIRE(walk_strings(rows, &o.chars_bytes), CHGPU_ERR_BAD_ARGUMENTS, short_msg, (unsigned long long)rows);
                    o.data_bytes = (u64)(r.p - data) - o.data_offset;
                }
                else
                {
                    const u64 nbytes = rows * (o.kind == CHGPU_NATIVE_FIXED_STRING ? (u64)o.fixed_n : (u64)chgpu_type_size(tag));
                    CHGPU_REQUIRE(nbytes <= (u64)(r.end - r.p), CHGPU_ERR_BAD_ARGUMENTS, short_msg, (unsigned long long)rows);
                    o.data_bytes = nbytes;
                    r.p += nbytes;
                }
            }
        }
        else if (lc)
            o.kind = CHGPU_NATIVE_LC_STRING, tag = CHGPU_U8;
        o.type = tag;
        if (c < capacity)
            columns[c] = o;
    }
    *n_columns = (u32)cols;
    *n_rows = rows;
    if (bucket_num)
        *bucket_num = bucket;
    if (is_overflows)
        *is_overflows = overflows;
    *bytes_consumed = (u64)(r.p - data);
    return CHGPU_OK;
}

/* SerializationString::deserializeBinaryBulk (src/DataTypes/Serializations/SerializationString.cpp): `rows` values, each a VarUInt length and
   that many bytes, in host memory -> a ColumnString in HBM (chars: every value followed by a zero byte; offsets[i] = end of value i incl. the
   zero).  The lengths form a chain only the host can follow; it lays the bytes out once and the two buffers cross PCIe. */
extern "C" int chgpu_native_read_strings(chgpu_ctx * ctx, const uint8_t * serialized, uint64_t bytes, uint64_t rows, chgpu_col ** offsets_u64, chgpu_col ** chars_u8)
{
    CHGPU_REQUIRE(ctx && (serialized || bytes == 0) && offsets_u64 && chars_u8, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    ChgpuDeviceGuard guard(ctx);
    Reader r{serialized, serialized + bytes};
    std::vector<u64> offs(rows ? rows : 1);
    std::vector<u8> chars;
    chars.reserve(bytes + rows + 8);
    for (u64 i = 0; i < rows; ++i)
    {
        const u64 len = r.varuint();
        CHGPU_REQUIRE(r.ok && len <= (u64)(r.end - r.p), CHGPU_ERR_BAD_ARGUMENTS, "Cannot read all data: truncated String value %llu (CANNOT_READ_ALL_DATA)", (unsigned long long)i);
        chars.insert(chars.end(), r.p, r.p + len);
        chars.push_back(0);
        r.p += len;
        offs[i] = chars.size();
    }
    chgpu_col * o = nullptr;
    CHGPU_TRY(chgpu_col_upload(ctx, CHGPU_U64, offs.data(), rows, &o));
    chgpu_col * ch = nullptr;
    const int rc = chgpu_col_upload(ctx, CHGPU_U8, chars.data(), chars.size(), &ch);
    if (rc != CHGPU_OK)
    {
        chgpu_col_free(o);
        return rc;
    }
    *offsets_u64 = o;
    *chars_u8 = ch;
    return CHGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// serialized aggregate-function states
// ---------------------------------------------------------------------------------------------
static constexpr u32 ST = 256;

__device__ __forceinline__ u32 varuint_len(u64 x)
{
    // writeVarUInt: 7 bits per byte
    return x == 0 ? 1u : (u32)((64 - __clzll((long long)x) + 6) / 7);
}

// bytes per serialized state: fixed 8 (sum), varuint (count), 8 + varuint (avg)
__global__ __launch_bounds__(ST) void k_state_sizes(int kind, const u64 * __restrict__ w0, const u64 * __restrict__ w1, u64 n, u32 * __restrict__ sizes)
{
    for (u64 i = (u64)blockIdx.x * ST + threadIdx.x; i < n; i += (u64)gridDim.x * ST)
        sizes[i] = kind == CHGPU_AGG_SUM ? 8u : kind == CHGPU_AGG_COUNT ? varuint_len(w0[i]) : 8u + varuint_len(w1[i]);
}

__global__ __launch_bounds__(ST) void k_state_write(int kind, const u64 * __restrict__ w0, const u64 * __restrict__ w1, u64 n, const u64 * __restrict__ offsets, u8 * __restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * ST + threadIdx.x; i < n; i += (u64)gridDim.x * ST)
    {
        u8 * p = out + offsets[i];
        u64 var;
        if (kind == CHGPU_AGG_COUNT)
            var = w0[i];
        else
        {
            const u64 v = w0[i]; // the sum / numerator: 8 bytes little endian whatever its type (Int64, UInt64 or Float64 bits)
            for (int b = 0; b < 8; ++b)
                p[b] = (u8)(v >> (8 * b));
            if (kind == CHGPU_AGG_SUM)
                continue;
            p += 8;
            var = w1[i];
        }
        while (var >= 0x80)
        {
            *p++ = (u8)(var | 0x80);
            var >>= 7;
        }
        *p = (u8)var;
    }
}

// One lane walks one stream of `rows` states from byte `begin`: the length of a state is only known once its VarUInt has been read, so
// a stream is sequential; the streams of a bucket-wise exchange (256 two-level buckets per rank) run in parallel, one per lane.
__global__ __launch_bounds__(ST) void k_state_read(int kind, const u8 * __restrict__ bytes, u64 n_bytes, const u64 * __restrict__ stream_begin, const u64 * __restrict__ stream_row0,
                                                  const u64 * __restrict__ stream_rows, u32 n_streams, u64 * __restrict__ w0, u64 * __restrict__ w1, u32 * __restrict__ err)
{
    const u32 sidx = blockIdx.x * ST + threadIdx.x;
    if (sidx >= n_streams)
        return;
    u64 p = stream_begin[sidx];
    const u64 r0 = stream_row0[sidx], rows = stream_rows[sidx];
    for (u64 i = 0; i < rows; ++i)
    {
        if (kind != CHGPU_AGG_COUNT)
        {
            if (p + 8 > n_bytes)
            {
                atomicOr(err, 1u);
                return;
            }
            u64 v = 0;
            for (int b = 0; b < 8; ++b)
                v |= (u64)bytes[p + b] << (8 * b);
            w0[r0 + i] = v;
            p += 8;
            if (kind == CHGPU_AGG_SUM)
                continue;
        }
        u64 x = 0;
        bool done = false;
        for (int k = 0; k < 10 && !done; ++k)
        {
            if (p >= n_bytes)
            {
                atomicOr(err, 1u);
                return;
            }
            const u8 b = bytes[p++];
            x |= (u64)(b & 0x7F) << (7 * k);
            done = !(b & 0x80);
        }
        if (!done)
        {
            atomicOr(err, 2u);
            return;
        }
        (kind == CHGPU_AGG_COUNT ? w0 : w1)[r0 + i] = x;
    }
    if (sidx + 1 == n_streams && p != n_bytes)
        atomicOr(err, 4u); // trailing bytes
    else if (sidx + 1 < n_streams && p != stream_begin[sidx + 1])
        atomicOr(err, 8u); // the stream did not end where the next one starts
}

extern "C" int chgpu_agg_serialize_states(chgpu_ctx * ctx, int kind, const chgpu_col * word0, const chgpu_col * word1, chgpu_col ** bytes_u8, chgpu_col ** offsets_u64)
{
    CHGPU_REQUIRE(ctx && word0 && bytes_u8, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(kind == CHGPU_AGG_SUM || kind == CHGPU_AGG_COUNT || kind == CHGPU_AGG_AVG, CHGPU_ERR_NOT_IMPLEMENTED, "aggregate function kind %d: CPU path", kind);
    CHGPU_REQUIRE(chgpu_type_size(word0->type) == 8, CHGPU_ERR_BAD_ARGUMENTS, "state words are 8-byte columns");
    CHGPU_REQUIRE(kind != CHGPU_AGG_AVG || (word1 && chgpu_type_size(word1->type) == 8 && word1->rows == word0->rows), CHGPU_ERR_BAD_ARGUMENTS,
                  "avg states are two word columns (numerator, denominator) of one length");
    ChgpuDeviceGuard guard(ctx);
    const u64 n = word0->rows;
    chgpu_col * offs = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, n + 1, &offs));
    auto fail = [&](int code, chgpu_col * extra = nullptr) {
        chgpu_col_free(offs);
        chgpu_col_free(extra);
        return code;
    };
    u64 total = 0;
    if (n)
    {
        auto al = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t tmp_b = chgpu_scan_tmp_bytes(n);
        void * scratch = nullptr;
        int rc = chgpu_scratch(ctx, al(n * 4) + 256 + tmp_b, &scratch);
        if (rc != CHGPU_OK)
            return fail(rc);
        u32 * sizes = (u32 *)scratch;
        u64 * total_dev = (u64 *)((char *)scratch + al(n * 4));
        const u32 grid = chgpu_grid_for(ctx, n, ST, 8);
        hipLaunchKernelGGL(k_state_sizes, dim3(grid), dim3(ST), 0, ctx->stream, kind, (const u64 *)word0->data, word1 ? (const u64 *)word1->data : nullptr, n, sizes);
        rc = chgpu_scan_exclusive_u32_u64(ctx, sizes, (u64 *)offs->data, n, total_dev, (char *)scratch + al(n * 4) + 256, tmp_b);
        if (rc == CHGPU_OK)
            rc = chgpu_read_back(ctx, total_dev, &total, 8);
        if (rc != CHGPU_OK)
            return fail(rc);
        if (hipMemcpyAsync((u64 *)offs->data + n, total_dev, 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            return fail(chgpu_set_error(CHGPU_ERR_DEVICE, "serialize_states: copy failed"));
        ctx->counters[6] += 1;
    }
    else if (hipMemsetAsync(offs->data, 0, 8, ctx->stream) != hipSuccess)
        return fail(chgpu_set_error(CHGPU_ERR_DEVICE, "serialize_states: memset failed"));
    chgpu_col * out = nullptr;
    int rc = chgpu_col_new(ctx, CHGPU_U8, total, &out);
    if (rc != CHGPU_OK)
        return fail(rc);
    if (n)
    {
        hipLaunchKernelGGL(k_state_write, dim3(chgpu_grid_for(ctx, n, ST, 8)), dim3(ST), 0, ctx->stream, kind, (const u64 *)word0->data, word1 ? (const u64 *)word1->data : nullptr, n,
                           (const u64 *)offs->data, (u8 *)out->data);
        ctx->counters[6] += 1;
        if (hipGetLastError() != hipSuccess)
            return fail(chgpu_set_error(CHGPU_ERR_DEVICE, "serialize_states launch failed"), out);
    }
    *bytes_u8 = out;
    if (offsets_u64)
        *offsets_u64 = offs;
    else
        chgpu_col_free(offs);
    return CHGPU_OK;
}

extern "C" int chgpu_agg_deserialize_states(chgpu_ctx * ctx, int kind, const chgpu_col * bytes_u8, uint32_t n_streams, const uint64_t * stream_byte_begin,
                                            const uint64_t * stream_rows, chgpu_col ** word0, chgpu_col ** word1)
{
    CHGPU_REQUIRE(ctx && bytes_u8 && word0 && stream_rows && n_streams >= 1, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(kind == CHGPU_AGG_SUM || kind == CHGPU_AGG_COUNT || kind == CHGPU_AGG_AVG, CHGPU_ERR_NOT_IMPLEMENTED, "aggregate function kind %d: CPU path", kind);
    CHGPU_REQUIRE(kind != CHGPU_AGG_AVG || word1, CHGPU_ERR_BAD_ARGUMENTS, "avg states come back as two word columns");
    CHGPU_REQUIRE(bytes_u8->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS, "serialized states are a UInt8 column");
    CHGPU_REQUIRE(n_streams == 1 || stream_byte_begin, CHGPU_ERR_BAD_ARGUMENTS, "several streams need their first bytes");
    ChgpuDeviceGuard guard(ctx);
    std::vector<u64> host(3 * (size_t)n_streams);
    u64 rows = 0;
    for (u32 s = 0; s < n_streams; ++s)
    {
        host[s] = stream_byte_begin ? stream_byte_begin[s] : 0;
        host[n_streams + s] = rows;
        host[2 * (size_t)n_streams + s] = stream_rows[s];
        rows += stream_rows[s];
        CHGPU_REQUIRE(host[s] <= bytes_u8->rows, CHGPU_ERR_BAD_ARGUMENTS, "stream %u begins outside the buffer", s);
    }
    chgpu_col * c0 = nullptr;
    chgpu_col * c1 = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U64, rows, &c0));
    auto fail = [&](int code) {
        chgpu_col_free(c0);
        chgpu_col_free(c1);
        return code;
    };
    int rc = kind == CHGPU_AGG_AVG ? chgpu_col_new(ctx, CHGPU_U64, rows, &c1) : CHGPU_OK;
    if (rc != CHGPU_OK)
        return fail(rc);
    void * scratch = nullptr;
    if ((rc = chgpu_scratch(ctx, host.size() * 8 + 256, &scratch)) != CHGPU_OK)
        return fail(rc);
    u64 * dev = (u64 *)scratch;
    u32 * err = (u32 *)(dev + host.size());
    if (hipMemcpyAsync(dev, host.data(), host.size() * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipMemsetAsync(err, 0, 4, ctx->stream) != hipSuccess)
        return fail(chgpu_set_error(CHGPU_ERR_DEVICE, "deserialize_states: staging failed"));
    hipLaunchKernelGGL(k_state_read, dim3((n_streams + ST - 1) / ST), dim3(ST), 0, ctx->stream, kind, (const u8 *)bytes_u8->data, (u64)bytes_u8->rows, (const u64 *)dev,
                       (const u64 *)(dev + n_streams), (const u64 *)(dev + 2 * (size_t)n_streams), n_streams, (u64 *)c0->data, c1 ? (u64 *)c1->data : nullptr, err);
    ctx->counters[6] += 1;
    u32 failed = 0;
    if ((rc = chgpu_read_back(ctx, err, &failed, 4)) != CHGPU_OK) // also: `host` stays alive until the copy has run
        return fail(rc);
    if (failed)
        return fail(chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "Cannot read all data: serialized states are truncated or malformed (code %u)", failed));
    *word0 = c0;
    if (word1)
        *word1 = c1;
    return CHGPU_OK;
}
