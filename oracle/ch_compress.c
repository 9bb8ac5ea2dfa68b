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
