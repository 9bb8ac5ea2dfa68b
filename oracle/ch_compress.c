/* TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's frame decompression for the feed path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * cho_lz4_decompress follows LZ4::decompress / decompressImpl (src/Compression/LZ4_decompress_faster.cpp:470-684): the LZ4 block
 * format — token, optional literal-length bytes (:489-496, :518-523), literals, 2-byte little-endian offset, optional match-length
 * bytes, match of length + 4 copied from `offset` bytes back, overlap allowed (copyOverlap*, :41-420) — without the reference's
 * wide-copy over-writes (which only touch padding).  Returns 0 on success, -1 where the reference returns false
 * (CANNOT_DECOMPRESS).  The byte-wise match copy is the definition of an overlapping LZ4 match.
 * PARITY UNPINNED by the reference's own fixtures (it ships no small compressed column files); cross-checked instead against an
 * independent implementation of the same format: tests compress with Apache Arrow's bundled liblz4 (pyarrow codec "lz4_raw", the
 * library the reference links as contrib/lz4) and require this decoder — and the device decoder — to return the original bytes.
 * cho_delta_decode follows CompressionCodecDelta::doDecompressData (src/Compression/CompressionCodecDelta.cpp:84-175).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

int cho_lz4_decompress(const uint8_t * src, size_t src_size, uint8_t * dst, size_t dst_size)
{
    size_t ip = 0, op = 0;
    for (;;)
    {
        if (ip >= src_size)
            return -1;
        const unsigned token = src[ip++];
        size_t length = token >> 4;
        if (length == 15)
        {
            unsigned s;
            do
            {
                if (ip >= src_size)
                    return -1;
                s = src[ip++];
                length += s;
            } while (s == 255);
        }
        if (length > src_size - ip || length > dst_size - op)
            return -1;
        memcpy(dst + op, src + ip, length);
        ip += length;
        op += length;
        if (ip >= src_size)
            break;
        if (src_size - ip < 2)
            return -1;
        const size_t offset = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
        ip += 2;
        length = token & 15;
        if (length == 15)
        {
            unsigned s;
            do
            {
                if (ip >= src_size)
                    return -1;
                s = src[ip++];
                length += s;
            } while (s == 255);
        }
        length += 4;
        if (offset == 0 || offset > op || length > dst_size - op)
            return -1;
        for (size_t k = 0; k < length; ++k)
            dst[op + k] = dst[op - offset + k];
        op += length;
    }
    return op == dst_size ? 0 : -1;
}

/* payload: [delta_bytes_size][bytes_to_skip][skipped bytes][deltas...] -> running sums of the element width, wrap-around */
int cho_delta_decode(const uint8_t * src, size_t src_size, uint8_t * dst, size_t dst_size)
{
    if (src_size < 2)
        return -1;
    if (dst_size == 0)
        return 0;
    const unsigned w = src[0], skip = src[1];
    if (!(w == 1 || w == 2 || w == 4 || w == 8) || skip > dst_size || 2 + (size_t)skip > src_size)
        return -1;
    memcpy(dst, src + 2, skip);
    const size_t n = src_size - 2 - skip;
    if (n % w != 0 || n != dst_size - skip)
        return -1;
    uint64_t acc = 0;
    for (size_t i = 0; i < n; i += w)
    {
        uint64_t d = 0;
        memcpy(&d, src + 2 + skip + i, w);
        acc += d;
        memcpy(dst + skip + i, &acc, w);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * CityHash128, cityhash version 1.0.2 -- the checksum in front of every compressed frame (CompressedReadBufferBase.cpp:49-51 calls
 * CityHash_v1_0_2::CityHash128 over header + payload).  Third-party algorithm (Google, MIT licence), frozen by ClickHouse at 1.0.2 in
 * contrib/cityhash102; restated here from the published algorithm, pinned against that copy compiled in place
 * (oracle/ref_city_wrapper.cpp -> oracle/_ref/libchref_city.so) and by the vectors in tests/golden/round2_kat.json.
 * ---------------------------------------------------------------------------------------------- */
#include <stdint.h>
#include <string.h>

#define CK0 0xc3a5c85c97cb3127ULL
#define CK1 0xb492b66fbe98f273ULL
#define CK2 0x9ae16a3b2f90404fULL
#define CK3 0xc949d7c7509e6557ULL
typedef struct { uint64_t first, second; } cpair;

static uint64_t c_f64(const unsigned char * p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint64_t c_f32(const unsigned char * p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t c_rot(uint64_t v, int s) { return s == 0 ? v : ((v >> s) | (v << (64 - s))); }
static uint64_t c_mix(uint64_t v) { return v ^ (v >> 47); }
static uint64_t c_h16(uint64_t u, uint64_t v)
{
    const uint64_t mul = 0x9ddfea08eb382d69ULL;
    uint64_t a = (u ^ v) * mul;
    a ^= (a >> 47);
    uint64_t b = (v ^ a) * mul;
    b ^= (b >> 47);
    b *= mul;
    return b;
}
static uint64_t c_len0to16(const unsigned char * s, size_t len)
{
    if (len > 8)
    {
        uint64_t a = c_f64(s), b = c_f64(s + len - 8);
        return c_h16(a, c_rot(b + len, (int)len)) ^ b;
    }
    if (len >= 4)
    {
        uint64_t a = c_f32(s);
        return c_h16(len + (a << 3), c_f32(s + len - 4));
    }
    if (len > 0)
    {
        unsigned char a = s[0], b = s[len >> 1], c = s[len - 1];
        uint32_t y = (uint32_t)a + ((uint32_t)b << 8);
        uint32_t z = (uint32_t)len + ((uint32_t)c << 2);
        return c_mix(y * CK2 ^ z * CK3) * CK2;
    }
    return CK2;
}
static cpair c_weak(uint64_t w, uint64_t x, uint64_t y, uint64_t z, uint64_t a, uint64_t b)
{
    a += w;
    b = c_rot(b + a + z, 21);
    uint64_t c = a;
    a += x;
    a += y;
    b += c_rot(a, 44);
    cpair r = {a + z, b + c};
    return r;
}
static cpair c_weak_s(const unsigned char * s, uint64_t a, uint64_t b) { return c_weak(c_f64(s), c_f64(s + 8), c_f64(s + 16), c_f64(s + 24), a, b); }

static cpair c_murmur(const unsigned char * s, size_t len, cpair seed)
{
    uint64_t a = seed.first, b = seed.second, c = 0, d = 0;
    long l = (long)len - 16;
    if (l <= 0)
    {
        a = c_mix(a * CK1) * CK1;
        c = b * CK1 + c_len0to16(s, len);
        d = c_mix(a + (len >= 8 ? c_f64(s) : c));
    }
    else
    {
        c = c_h16(c_f64(s + len - 8) + CK1, a);
        d = c_h16(b + len, c + c_f64(s + len - 16));
        a += d;
        do
        {
            a ^= c_mix(c_f64(s) * CK1) * CK1;
            a *= CK1;
            b ^= a;
            c ^= c_mix(c_f64(s + 8) * CK1) * CK1;
            c *= CK1;
            d ^= c;
            s += 16;
            l -= 16;
        } while (l > 0);
    }
    a = c_h16(a, c);
    b = c_h16(d, b);
    cpair r = {a ^ b, c_h16(b, a)};
    return r;
}

static cpair c_seeded(const unsigned char * s, size_t len, cpair seed)
{
    if (len < 128)
        return c_murmur(s, len, seed);
    cpair v, w;
    uint64_t x = seed.first, y = seed.second, z = len * CK1, t;
    v.first = c_rot(y ^ CK1, 49) * CK1 + c_f64(s);
    v.second = c_rot(v.first, 42) * CK1 + c_f64(s + 8);
    w.first = c_rot(y + z, 35) * CK1 + x;
    w.second = c_rot(x + c_f64(s + 88), 53) * CK1;
    do
    {
        int rep;
        for (rep = 0; rep < 2; ++rep)
        {
            x = c_rot(x + y + v.first + c_f64(s + 16), 37) * CK1;
            y = c_rot(y + v.second + c_f64(s + 48), 42) * CK1;
            x ^= w.second;
            y ^= v.first;
            z = c_rot(z ^ w.first, 33);
            v = c_weak_s(s, v.second * CK1, x + w.first);
            w = c_weak_s(s + 32, z + w.second, y);
            t = z, z = x, x = t;
            s += 64;
        }
        len -= 128;
    } while (len >= 128);
    y += c_rot(w.first, 37) * CK0 + z;
    x += c_rot(v.first + z, 49) * CK0;
    for (size_t tail_done = 0; tail_done < len;)
    {
        tail_done += 32;
        y = c_rot(y - x, 42) * CK0 + v.second;
        w.first += c_f64(s + len - tail_done + 16);
        x = c_rot(x, 49) * CK0 + w.first;
        w.first += v.first;
        v = c_weak_s(s + len - tail_done, v.first, v.second);
    }
    x = c_h16(x, v.first);
    y = c_h16(y, w.first);
    cpair r = {c_h16(x + v.second, w.second) + y, c_h16(x + w.second, y + v.second)};
    return r;
}

void cho_city_hash128(const void * data, size_t len, uint64_t out_low_high[2])
{
    const unsigned char * s = (const unsigned char *)data;
    cpair r, seed;
    if (len >= 16)
    {
        seed.first = c_f64(s) ^ CK3;
        seed.second = c_f64(s + 8);
        r = c_seeded(s + 16, len - 16, seed);
    }
    else if (len >= 8)
    {
        seed.first = c_f64(s) ^ (len * CK0);
        seed.second = c_f64(s + len - 8) ^ CK1;
        r = c_seeded(NULL, 0, seed);
    }
    else
    {
        seed.first = CK0;
        seed.second = CK1;
        r = c_seeded(s, len, seed);
    }
    out_low_high[0] = r.first;
    out_low_high[1] = r.second;
}

/* ------------------------------------------------------------------------------------------------
 * DoubleDelta and T64: restated from src/Compression/CompressionCodecDoubleDelta.cpp:140-560 and
 * src/Compression/CompressionCodecT64.cpp:231-677, bit writer / reader from src/IO/BitHelpers.h (big-endian bit order inside 64-bit
 * chunks, the buffer a 128-bit window).  Encoders are restated too: tests build frames with them; DoubleDelta's bytes are PINNED by the
 * reference's compatibility vectors (src/Compression/tests/gtest_compressionCodec.cpp:1171-1209, tests/golden/codec_kat.json) in both
 * directions.  T64's byte layout has no reference byte fixture (PARITY UNPINNED for the bytes; the stateless tests 00870-00873 pin the
 * round trip, which tests/test_compression.py repeats over the same value ranges).
 * Payload layout (what ICompressionCodec::compress writes after the 9-byte frame header):
 *   DoubleDelta: [data_bytes_size][bytes_to_skip][skipped bytes][items u32][first value][first delta][bit stream]
 *   T64:         [cookie = type magic | variant << 7][min 8 B][max 8 B][per 64 values: num_bits x UInt64 of the (bit-)transposed matrix]
 * ------------------------------------------------------------------------------------------------ */
typedef unsigned __int128 cho_u128;

typedef struct { uint8_t * cur; uint8_t * end; cho_u128 buf; unsigned bits; int overflow; } cho_bitw;
static void bw_flush_to(cho_bitw * w, unsigned to_write_bits)
{
    /* BitWriter::doFlush: the top `to_write_bits` (whole bytes) of the window leave in big-endian order */
    uint64_t hi = (uint64_t)(w->buf >> 64);
    const unsigned nbytes = to_write_bits / 8;
    for (unsigned k = 0; k < nbytes; ++k)
    {
        if (w->cur >= w->end) { w->overflow = 1; return; }
        *w->cur++ = (uint8_t)(hi >> (56 - 8 * k));
    }
    w->buf <<= to_write_bits;
    w->bits -= to_write_bits;
}
static void bw_write(cho_bitw * w, unsigned nbits, uint64_t value)
{
    if (nbits == 0) return;
    if (nbits < 64) value &= ((1ull << nbits) - 1);
    if (w->bits + nbits > 128)
        bw_flush_to(w, 64); /* (the reference flushes whole 64-bit words when the window cannot take the value) */
    w->buf |= (cho_u128)value << (128 - w->bits - nbits);
    w->bits += nbits;
    if (w->bits >= 64)
        bw_flush_to(w, 64);
}
static void bw_finish(cho_bitw * w)
{
    const unsigned whole = (w->bits + 7) / 8 * 8;
    if (whole)
    {
        uint64_t hi = (uint64_t)(w->buf >> 64);
        for (unsigned k = 0; k < whole / 8; ++k)
        {
            if (w->cur >= w->end) { w->overflow = 1; return; }
            *w->cur++ = k < 8 ? (uint8_t)(hi >> (56 - 8 * k)) : (uint8_t)((uint64_t)w->buf >> (56 - 8 * (k - 8)));
        }
    }
    w->bits = 0;
}

typedef struct { const uint8_t * cur; const uint8_t * end; cho_u128 buf; unsigned bits; } cho_bitr;
static void br_fill(cho_bitr * r)
{
    size_t avail = (size_t)(r->end - r->cur), n = avail < 8 ? avail : 8;
    if (n == 0) return;
    uint64_t tmp = 0;
    for (size_t k = 0; k < n; ++k) /* memcpy into a little-endian word, then byteswap: byte k lands at bits 63-8k .. 56-8k */
        tmp |= (uint64_t)r->cur[k] << (56 - 8 * k);
    r->cur += n;
    r->buf |= (cho_u128)tmp << (64 - r->bits);
    r->bits += (unsigned)n * 8;
}
static uint64_t br_read(cho_bitr * r, unsigned nbits)
{
    if (nbits == 0) return 0;
    if (nbits > r->bits) br_fill(r);
    const uint64_t v = (uint64_t)(r->buf >> (128 - nbits));
    r->buf <<= nbits;
    r->bits = r->bits >= nbits ? r->bits - nbits : 0;
    return v;
}
static unsigned br_peek_byte(cho_bitr * r)
{
    if (r->bits < 8) br_fill(r);
    return (unsigned)(uint64_t)(r->buf >> 120);
}
static int br_eof(const cho_bitr * r) { return r->bits == 0 && r->cur >= r->end; }

static uint64_t dd_load(const uint8_t * p, unsigned w) { uint64_t v = 0; memcpy(&v, p, w); return v; }
static void dd_store(uint8_t * p, unsigned w, uint64_t v) { memcpy(p, &v, w); }
static uint64_t dd_mask(unsigned w) { return w == 8 ? ~0ull : ((1ull << (8 * w)) - 1); }
static int64_t dd_signed(uint64_t v, unsigned w) { const unsigned sh = 64 - 8 * w; return (int64_t)(v << sh) >> sh; }

/* CompressionCodecDoubleDelta::doCompressData; returns the payload size, or -1 when dst is too small */
long cho_double_delta_encode(const uint8_t * src, size_t src_size, unsigned width, uint8_t * dst, size_t dst_cap)
{
    if (!(width == 1 || width == 2 || width == 4 || width == 8)) return -1;
    const unsigned skip = (unsigned)(src_size % width);
    if (dst_cap < 2 + skip + 4 + 2 * (size_t)width + 16) return -1;
    dst[0] = (uint8_t)width;
    dst[1] = (uint8_t)skip;
    memcpy(dst + 2, src, skip);
    uint8_t * d = dst + 2 + skip;
    const uint8_t * s = src + skip, * s_end = src + src_size;
    const uint32_t items = (uint32_t)((src_size - skip) / width);
    memcpy(d, &items, 4);
    d += 4;
    const uint64_t M = dd_mask(width);
    uint64_t prev_value = 0, prev_delta = 0;
    if (s < s_end) { prev_value = dd_load(s, width); dd_store(d, width, prev_value); s += width; d += width; }
    if (s < s_end) { const uint64_t cur = dd_load(s, width); prev_delta = (cur - prev_value) & M; dd_store(d, width, prev_delta); s += width; d += width; prev_value = cur; }
    cho_bitw w = {d, dst + dst_cap, 0, 0, 0};
    for (; s < s_end; s += width)
    {
        const uint64_t cur = dd_load(s, width);
        const uint64_t delta = (cur - prev_value) & M, dd = (delta - prev_delta) & M;
        prev_delta = delta;
        prev_value = cur;
        if (dd == 0) { bw_write(&w, 1, 0); continue; }
        const int64_t sdd = dd_signed(dd, width);
        const int64_t smin = width == 8 ? INT64_MIN : -(int64_t)(1ull << (8 * width - 1));
        const uint64_t abs_value = sdd == smin ? (M >> 1) : (uint64_t)((sdd < 0 ? -sdd : sdd) - 1);
        unsigned pbits, prefix, dbits;
        if (sdd > -63 && sdd < 64) { pbits = 2; prefix = 2; dbits = 7; }
        else if (sdd > -255 && sdd < 256) { pbits = 3; prefix = 6; dbits = 9; }
        else if (sdd > -2047 && sdd < 2048) { pbits = 4; prefix = 14; dbits = 12; }
        else if (sdd > INT32_MIN && sdd < INT32_MAX) { pbits = 5; prefix = 30; dbits = 32; }
        else { pbits = 5; prefix = 31; dbits = 64; }
        bw_write(&w, pbits, prefix);
        bw_write(&w, 1, sdd < 0);
        bw_write(&w, dbits - 1, abs_value);
    }
    bw_finish(&w);
    if (w.overflow) return -1;
    return (long)(w.cur - dst);
}

/* CompressionCodecDoubleDelta::doDecompressData; 0 ok, -1 CANNOT_DECOMPRESS */
int cho_double_delta_decode(const uint8_t * src, size_t src_size, uint8_t * dst, size_t dst_size)
{
    if (src_size < 2) return -1;
    const unsigned width = src[0], skip = src[1];
    if (!(width == 1 || width == 2 || width == 4 || width == 8)) return -1;
    if (skip > dst_size || 2 + (size_t)skip > src_size) return -1;
    memcpy(dst, src + 2, skip);
    const uint8_t * s = src + 2 + skip, * s_end = src + src_size;
    uint8_t * d = dst + skip, * d_end = dst + dst_size;
    if (s + 4 > s_end) return 0;
    uint32_t items;
    memcpy(&items, s, 4);
    s += 4;
    const uint64_t M = dd_mask(width);
    static const uint8_t PB[32] = {1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1, 2,2,2,2,2,2,2,2, 3,3,3,3, 4,4, 5,5};
    static const uint8_t DB[32] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 7,7,7,7,7,7,7,7, 9,9,9,9, 12,12, 32,64};
    if (s + width > s_end || items < 1) return 0;
    uint64_t prev_value = dd_load(s, width), prev_delta;
    if (d + width > d_end) return -1;
    dd_store(d, width, prev_value);
    s += width; d += width;
    if (s + width > s_end || items < 2) return 0;
    prev_delta = dd_load(s, width);
    prev_value = (prev_value + prev_delta) & M;
    if (d + width > d_end) return -1;
    dd_store(d, width, prev_value);
    s += width; d += width;
    cho_bitr r = {s, s_end, 0, 0};
    for (uint32_t read = 2; read < items && !br_eof(&r); ++read)
    {
        const unsigned idx = br_peek_byte(&r) >> 3;
        (void)br_read(&r, PB[idx]); /* skipBufferedBits */
        uint64_t dd = 0;
        if (DB[idx])
        {
            const unsigned sign = (unsigned)br_read(&r, 1);
            dd = (br_read(&r, DB[idx] - 1) + 1) & M;
            if (sign) dd = (0 - dd) & M;
        }
        const uint64_t delta = (dd + prev_delta) & M, cur = (prev_value + delta) & M;
        if (d + width > d_end) return -1;
        dd_store(d, width, cur);
        d += width;
        prev_delta = (cur - prev_value) & M;
        prev_value = cur;
    }
    return 0;
}

/* ---- T64 ---- */
static unsigned t64_bits_u(uint64_t mn, uint64_t mx) { const uint64_t x = mn ^ mx; return x ? 64 - (unsigned)__builtin_clzll(x) : 0; }
static unsigned t64_bits_s(int64_t mn, int64_t mx)
{
    if (mn < 0 && mx >= 0)
        return (mn + mx >= 0) ? t64_bits_u(0, (uint64_t)mx) + 1 : t64_bits_u(0, (uint64_t)~mn) + 1; /* opposite signs: the sum cannot overflow */
    return t64_bits_u((uint64_t)mn, (uint64_t)mx);
}
static void t64_transpose64x8(uint64_t * m)
{
    const uint8_t * s8 = (const uint8_t *)m;
    uint64_t dst[8] = {0};
    for (unsigned i = 0; i < 64; ++i)
        for (unsigned b = 0; b < 8; ++b)
            dst[b] |= (uint64_t)((s8[i] >> b) & 1u) << i;
    memcpy(m, dst, 64);
}
static void t64_reverse64x8(uint64_t * m)
{
    uint8_t d8[64];
    for (unsigned i = 0; i < 64; ++i)
    {
        unsigned v = 0;
        for (unsigned b = 0; b < 8; ++b)
            v |= (unsigned)((m[b] >> i) & 1u) << b;
        d8[i] = (uint8_t)v;
    }
    memcpy(m, d8, 64);
}
/* type_magic: the cookie's low 7 bits (MagicNumber: 1..4 UInt8..64, 6..9 Int8..64); full = the 'bit' variant */
long cho_t64_encode(const uint8_t * src, size_t src_size, unsigned width, int is_signed, unsigned type_magic, int full, uint8_t * dst, size_t dst_cap)
{
    if (!(width == 1 || width == 2 || width == 4 || width == 8) || src_size % width || src_size == 0) return -1;
    const size_t n = src_size / width;
    uint64_t mn_u = 0, mx_u = 0; int64_t mn_s = 0, mx_s = 0;
    for (size_t i = 0; i < n; ++i)
    {
        const uint64_t v = dd_load(src + i * width, width);
        const int64_t sv = dd_signed(v, width);
        if (i == 0) { mn_u = mx_u = v; mn_s = mx_s = sv; }
        if (v < mn_u) mn_u = v;
        if (v > mx_u) mx_u = v;
        if (sv < mn_s) mn_s = sv;
        if (sv > mx_s) mx_s = sv;
    }
    const unsigned num_bits = is_signed ? t64_bits_s(mn_s, mx_s) : t64_bits_u(mn_u, mx_u);
    const size_t blocks = (n + 63) / 64;
    if (dst_cap < 17 + blocks * 8 * (size_t)num_bits) return -1;
    dst[0] = (uint8_t)(type_magic | (full ? 0x80u : 0u));
    if (is_signed) { memcpy(dst + 1, &mn_s, 8); memcpy(dst + 9, &mx_s, 8); }
    else { memcpy(dst + 1, &mn_u, 8); memcpy(dst + 9, &mx_u, 8); }
    uint8_t * d = dst + 17;
    if (!num_bits) return 17;
    const unsigned full_bytes = num_bits / 8, part_bits = num_bits % 8;
    for (size_t b = 0; b < blocks; ++b)
    {
        const unsigned tail = (unsigned)((n - b * 64) < 64 ? (n - b * 64) : 64);
        uint64_t matrix[64];
        memset(matrix, 0, sizeof(matrix));
        uint8_t * m8 = (uint8_t *)matrix;
        for (unsigned col = 0; col < tail; ++col)
            for (unsigned k = 0; k < width; ++k)
                m8[64 * k + col] = src[(b * 64 + col) * width + k];
        if (full)
            for (unsigned k = 0; k < full_bytes; ++k)
                t64_transpose64x8(matrix + 8 * k);
        memcpy(d, matrix, 8 * (size_t)(num_bits - part_bits));
        d += 8 * (size_t)(num_bits - part_bits);
        if (part_bits)
        {
            t64_transpose64x8(matrix + 8 * full_bytes);
            memcpy(d, matrix + 8 * full_bytes, 8 * (size_t)part_bits);
            d += 8 * (size_t)part_bits;
        }
    }
    return (long)(d - dst);
}

int cho_t64_decode(const uint8_t * src, size_t src_size, uint8_t * dst, size_t dst_size)
{
    if (src_size < 1) return -1;
    const unsigned cookie = src[0], magic = cookie & 0x7F, full = cookie >> 7;
    unsigned width; int is_signed;
    switch (magic)
    {
        case 1: case 17: width = 1; is_signed = magic == 17; break;             /* UInt8, Enum8 (Int8 base) */
        case 2: case 13: width = 2; is_signed = 0; break;                       /* UInt16, Date */
        case 3: case 14: case 21: width = 4; is_signed = 0; break;              /* UInt32, DateTime, IPv4 */
        case 4: width = 8; is_signed = 0; break;
        case 6: width = 1; is_signed = 1; break;
        case 7: case 18: width = 2; is_signed = 1; break;                       /* Int16, Enum16 */
        case 8: case 19: case 22: width = 4; is_signed = 1; break;              /* Int32, Decimal32, Date32 */
        case 9: case 15: case 20: width = 8; is_signed = 1; break;              /* Int64, DateTime64, Decimal64 */
        default: return -1;
    }
    src += 1; src_size -= 1;
    if (src_size < 16 || dst_size % width) return -1;
    const uint64_t n = dst_size / width;
    uint64_t mn_u, mx_u; int64_t mn_s, mx_s;
    memcpy(&mn_u, src, 8); memcpy(&mx_u, src + 8, 8);
    memcpy(&mn_s, src, 8); memcpy(&mx_s, src + 8, 8);
    src += 16; src_size -= 16;
    const unsigned num_bits = is_signed ? t64_bits_s(mn_s, mx_s) : t64_bits_u(mn_u, mx_u);
    const uint64_t M = dd_mask(width);
    if (!num_bits)
    {
        for (uint64_t i = 0; i < n; ++i)
            dd_store(dst + i * width, width, mn_u & M);
        return 0;
    }
    const size_t src_shift = 8 * (size_t)num_bits;
    if (!src_size || src_size % src_shift) return -1;
    uint64_t num_full = src_size / src_shift;
    const unsigned tail = (unsigned)(n % 64);
    if (tail) --num_full;
    if (num_full * 64 + tail != n) return -1;
    uint64_t upper_min = 0, upper_max = 0, sign_bit = 0;
    if (num_bits < 64) upper_min = (mn_u >> num_bits << num_bits) & M;
    if (is_signed && mn_s < 0 && mx_s >= 0 && num_bits < 64)
    {
        sign_bit = (1ull << (num_bits - 1)) & M;
        upper_max = (mx_u >> num_bits << num_bits) & M;
    }
    const unsigned full_bytes = num_bits / 8, part_bits = num_bits % 8;
    const uint64_t blocks = num_full + (tail ? 1 : 0);
    for (uint64_t b = 0; b < blocks; ++b)
    {
        const unsigned cnt = (b == num_full) ? tail : 64;
        uint64_t matrix[64];
        memset(matrix, 0, sizeof(matrix));
        memcpy(matrix, src + b * src_shift, src_shift);
        if (full)
            for (unsigned k = 0; k < full_bytes; ++k)
                t64_reverse64x8(matrix + 8 * k);
        if (part_bits)
            t64_reverse64x8(matrix + 8 * full_bytes);
        const uint8_t * m8 = (const uint8_t *)matrix;
        for (unsigned col = 0; col < cnt; ++col)
        {
            uint64_t v = 0;
            for (unsigned k = 0; k < width; ++k)
                v |= (uint64_t)m8[64 * k + col] << (8 * k);
            if (is_signed && sign_bit)
                v |= (v & sign_bit) ? upper_min : upper_max;
            else
                v |= upper_min;
            dd_store(dst + (b * 64 + col) * width, width, v & M);
        }
    }
    return 0;
}

/* ---- Gorilla (src/Compression/CompressionCodecGorilla.cpp:196-330): [width][bytes_to_skip][skipped][items u32][first value] and a bit
 * stream of XOR differences: 0 = same value | 10 + the meaningful bits inside the PREVIOUS window of leading / trailing zeros | 11 +
 * leading zeros (W - 1 bits) + length (W bits) + the meaningful bits, W = 4 / 5 / 6 / 7 for 1 / 2 / 4 / 8-byte values.  Pinned by the worked
 * example in the codec's documentation comment (:58-104, tests/golden/codec_kat.json); the reference ships no compatibility vectors. ---- */
static unsigned gorilla_len_bits(unsigned width) { return width == 1 ? 4 : width == 2 ? 5 : width == 4 ? 6 : 7; }

long cho_gorilla_encode(const uint8_t * src, size_t src_size, unsigned width, uint8_t * dst, size_t dst_cap)
{
    if (!(width == 1 || width == 2 || width == 4 || width == 8)) return -1;
    const unsigned skip = (unsigned)(src_size % width), X = 8 * width, DBL = gorilla_len_bits(width), LZL = DBL - 1;
    if (dst_cap < 2 + skip + 4 + (size_t)width + 16) return -1;
    dst[0] = (uint8_t)width;
    dst[1] = (uint8_t)skip;
    memcpy(dst + 2, src, skip);
    uint8_t * d = dst + 2 + skip;
    const uint8_t * s = src + skip, * s_end = src + src_size;
    const uint32_t items = (uint32_t)((src_size - skip) / width);
    memcpy(d, &items, 4);
    d += 4;
    uint64_t prev = 0;
    unsigned p_lz = 0, p_db = 0, p_tz = 0;
    if (s < s_end) { prev = dd_load(s, width); dd_store(d, width, prev); s += width; d += width; }
    cho_bitw w = {d, dst + dst_cap, 0, 0, 0};
    for (; s < s_end; s += width)
    {
        const uint64_t cur = dd_load(s, width), x = cur ^ prev;
        if (x == 0)
            bw_write(&w, 1, 0);
        else
        {
            const unsigned lz = (unsigned)__builtin_clzll(x) - (64 - X), tz = (unsigned)__builtin_ctzll(x), db = X - lz - tz;
            if (p_db != 0 && p_lz <= lz && p_tz <= tz)
            {
                bw_write(&w, 2, 2);
                bw_write(&w, p_db, x >> p_tz);
            }
            else
            {
                bw_write(&w, 2, 3);
                bw_write(&w, LZL, lz);
                bw_write(&w, DBL, db);
                bw_write(&w, db, x >> tz);
                p_lz = lz; p_db = db; p_tz = tz;
            }
        }
        prev = cur;
    }
    bw_finish(&w);
    if (w.overflow) return -1;
    return (long)(w.cur - dst);
}

int cho_gorilla_decode(const uint8_t * src, size_t src_size, uint8_t * dst, size_t dst_size)
{
    if (src_size < 2) return -1;
    const unsigned width = src[0], skip = src[1];
    if (!(width == 1 || width == 2 || width == 4 || width == 8)) return -1;
    if (skip > dst_size || 2 + (size_t)skip > src_size) return -1;
    memcpy(dst, src + 2, skip);
    const unsigned X = 8 * width, DBL = gorilla_len_bits(width), LZL = DBL - 1;
    const uint8_t * s = src + 2 + skip, * s_end = src + src_size;
    uint8_t * d = dst + skip;
    const size_t room = dst_size - skip;
    if (s + 4 > s_end) return 0;
    uint32_t items;
    memcpy(&items, s, 4);
    s += 4;
    if (s + width > s_end || items < 1) return 0;
    if ((uint64_t)items * width > room) return -1;
    uint64_t prev = dd_load(s, width);
    dd_store(d, width, prev);
    s += width; d += width;
    cho_bitr r = {s, s_end, 0, 0};
    unsigned p_lz = 0, p_db = 0, p_tz = 0;
    const uint64_t M = dd_mask(width);
    for (uint32_t read = 1; read < items && !br_eof(&r); ++read)
    {
        uint64_t cur = prev;
        unsigned lz = p_lz, db = p_db, tz = p_tz;
        if (br_read(&r, 1) == 1)
        {
            if (br_read(&r, 1) == 1)
            {
                lz = (unsigned)br_read(&r, LZL);
                db = (unsigned)br_read(&r, DBL);
                tz = X - lz - db; /* (UInt8 arithmetic in the reference: a corrupted length wraps there too) */
                tz &= 0xFF;
            }
            if (lz == 0 && db == 0 && tz == 0) return -1;
            uint64_t x = db > 64 ? 0 : br_read(&r, db);
            x = tz >= 64 ? 0 : x << tz;
            cur = (prev ^ x) & M;
        }
        dd_store(d, width, cur);
        d += width;
        p_lz = lz; p_db = db; p_tz = tz;
        prev = cur;
    }
    return 0;
}
