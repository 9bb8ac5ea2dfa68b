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
