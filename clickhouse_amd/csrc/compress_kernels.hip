// compress_kernels.hip — SURVEY §8(f) rank 3: the feed side.  A MergeTree column file / a compressed Native stream is a sequence of
// frames (src/Compression/CompressedReadBufferBase.cpp:175-222, CompressionInfo.h:10-51): 16-byte CityHash128 checksum, method byte,
// compressed size, decompressed size, payload.  The reference decompresses every frame on a CPU core
// (LZ4_decompress_faster.cpp, ~2-5 GB/s per core) into host memory which then has to cross PCIe uncompressed; here the compressed
// bytes cross PCIe and the frames are decoded in HBM, one wavefront per frame, straight into the column's buffer.
//
//   k_lz4_decode    LZ4 block format (token | literal length bytes | literals | 2-byte offset | match length bytes).  The wave parses
//                   the sequence headers uniformly (values broadcast with readfirstlane -> scalar branches) and all 64 lanes copy:
//                   literals byte-per-lane, matches from the output already written -- an overlapping match (offset < length) is
//                   the `offset` bytes before it repeated, so byte k reads position (k mod offset) of that window and no lane
//                   depends on another lane of the same copy.  Input window and recent output are kept in LDS (see the kernel).
//                   Every input and output position is bounds-checked: a malformed frame sets the error flag, never faults.
//                   method NONE (0x02): payload copied as is
//   k_delta_decode  the Delta stage of CODEC(Delta, LZ4): running sums of 1/2/4/8-byte elements, wave scan + uniform carry
// Algorithmic bytes: compressed size read + decompressed size written (match sources are re-reads of fresh output: L1/L2).
#include "chgpu_internal.h"

struct FrameJob
{
    u64 src_off;  // payload begin in the compressed buffer
    u64 dst_off;  // begin in the output buffer
    u32 src_size; // payload bytes
    u32 dst_size; // decompressed bytes
    u32 method;   // 0x82 LZ4, 0x02 NONE
    u32 post;     // 0: dst is the output buffer; 0x92 .. 0x95: dst is the stage buffer and a Delta / T64 / DoubleDelta / Gorilla stage follows
                  // (CODEC(Delta, LZ4), CODEC(DoubleDelta, ZSTD), ...: the general-purpose stage's output is that codec's block, header first)
    u64 out_off;  // post != 0: where the codec stage writes in the output buffer
    u32 out_size; // post != 0: final decompressed size
    u32 pad;
};

__device__ __forceinline__ u32 uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 uni64(u64 v) { return ((u64)uni((u32)(v >> 32)) << 32) | uni((u32)v); }

// Where codec M of a frame reads and writes.  stage == nullptr: the frames that ARE an application of M (CODEC(T64) alone).  Otherwise the
// frames whose general-purpose stage left M's block in the stage buffer (CompressionCodecMultiple.cpp:68-130: every stage carries its own
// 9-byte header -- method, compressed size incl. the header, decompressed size -- which must agree with what the outer frame promised).
struct CodecIo
{
    const u8 * in;
    u8 * out;
    u32 isz, osz;
    bool take, bad;
};
__device__ __forceinline__ CodecIo codec_io(const FrameJob & job, u32 M, const u8 * src, const u8 * stage, u8 * dst_out)
{
    CodecIo io{nullptr, nullptr, 0, 0, false, false};
    if (!stage)
    {
        if (job.method != M)
            return io;
        io.take = true;
        io.in = src + job.src_off;
        io.out = dst_out + job.dst_off;
        io.isz = job.src_size;
        io.osz = job.dst_size;
        return io;
    }
    if (job.post != M)
        return io;
    io.take = true;
    const u8 * h = stage + job.dst_off;
    const u32 ssz = job.dst_size;
    if (ssz < 9)
    {
        io.bad = true;
        return io;
    }
    const u32 csize = (u32)h[1] | ((u32)h[2] << 8) | ((u32)h[3] << 16) | ((u32)h[4] << 24), dsize = (u32)h[5] | ((u32)h[6] << 8) | ((u32)h[7] << 16) | ((u32)h[8] << 24);
    io.bad = h[0] != M || csize != ssz || dsize != job.out_size;
    io.in = h + 9;
    io.isz = ssz - 9;
    io.out = dst_out + job.out_off;
    io.osz = job.out_size;
    return io;
}

// Per wave: a window of the compressed input and a ring of the most recent output live in LDS, so the serial part of the format --
// token, length bytes, offset -- and the short-distance matches that dominate column data (zero bytes 8 back, the previous value 8
// back, runs) never wait for a global load: the first version read everything through global memory and spent ~1.7 us per
// sequence in three dependent round trips (64 KiB frames of C2's column: 29 GB/s for the whole GPU).
static constexpr u32 LZ_IN = 1024;    // input window bytes per wave
static constexpr u32 LZ_RING = 4096;  // output ring bytes per wave (matches up to LZ_RING / 2 back are served from it); 20 KiB of LDS per
                                      // workgroup keeps 8 workgroups = 32 frames per CU in flight (8 KiB rings: 3 workgroups, two rounds for 6 K frames)
static constexpr u32 LZ_CHUNK = LZ_RING / 2;

__global__ __launch_bounds__(256) void k_lz4_decode(const u8 * __restrict__ src, u8 * dst_out, u8 * dst_stage, const FrameJob * __restrict__ jobs, u32 n_jobs, u32 * __restrict__ err)
{
    __shared__ __attribute__((aligned(16))) u8 lds_in[4][LZ_IN + 16];
    __shared__ __attribute__((aligned(16))) u8 lds_ring[4][LZ_RING];
    const u32 lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u8 * lin = lds_in[w];
    u8 * ring = lds_ring[w];
    const u32 wave0 = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    for (u32 j = wave0; j < n_jobs; j += n_waves)
    {
        const FrameJob job = jobs[j];
        // the frame's fields are the same in every lane: telling the compiler so keeps the base pointers and sizes in scalar
        // registers (stores become saddr + 32-bit lane offset instead of a 64-bit vector add per access)
        const u8 * in = src + uni64(job.src_off);
        u8 * out = (uni(job.post) ? dst_stage : dst_out) + uni64(job.dst_off);
        const u32 isz = uni(job.src_size), osz = uni(job.dst_size);
        if (uni(job.method) == 0x93u || uni(job.method) == 0x94u || uni(job.method) == 0x95u)
            continue; // T64 / DoubleDelta / Gorilla frames: k_t64_decode / k_double_delta_decode / k_gorilla_decode
        if (uni(job.method) == 0x02u)
        {
            if (isz != osz)
            {
                if (lane == 0)
                    atomicOr(err, 1u);
                continue;
            }
            for (u32 k = lane; k < osz; k += 64)
                out[k] = in[k];
            continue;
        }
        u32 ip = 0, op = 0;
        u32 in_base = 0, in_len = 0; // lin[0 .. in_len) = in[in_base .. in_base + in_len)
        bool bad = false;
        // make in[pos .. pos + need) available in the window (need <= 16); false at the end of the frame
        auto window = [&](u32 pos, u32 need) -> bool {
            if (pos + need > isz)
                return false;
            if (pos < in_base || pos + need > in_base + in_len)
            {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                in_base = pos;
                in_len = isz - pos < LZ_IN ? isz - pos : LZ_IN;
                for (u32 k = lane * 8; k < in_len; k += 64 * 8)
                {
                    if (k + 8 <= in_len)
                    {
                        u64 v;
                        __builtin_memcpy(&v, in + pos + k, 8); // unaligned 8-byte global load
                        *(u64 *)(lin + k) = v;
                    }
                    else
                        for (u32 q = k; q < in_len; ++q) // never read past the frame (a wrapped buffer may end right there)
                            lin[q] = in[pos + q];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            return true;
        };
        while (true)
        {
            // Fast path -- the shape of almost every sequence of a numeric column: literal and match lengths inside the token
            // (< 15 literals, < 19 match bytes), the whole sequence header inside the LDS window, the match source inside the ring.
            // One LDS read gives every lane one byte of the next 64 input bytes; token and offset are picked out of it with
            // v_readlane (no further LDS round trip); lanes 0..lit-1 store the literals, lanes 0..ml-1 the match.  ~45
            // instructions against ~170 on the general path below -- the decoder is bound by instruction issue.
            // (the window lies inside the frame, so 64 readable window bytes cover token + 14 literals + offset; 32 free output
            //  bytes cover 14 literals + 18 match bytes: two checks instead of one per field -- the decoder is bound by the CU's
            //  scalar unit, one instruction per clock for all its waves)
            if (ip >= in_base && ip + 64 <= in_base + in_len && op + 32 <= osz)
            {
                const u32 rel = ip - in_base;
                const u32 hb = lin[rel + lane];
                const u32 token = uni(hb);
                const u32 lit = token >> 4, mlt = token & 15;
                if (lit != 15 && mlt != 15)
                {
                    const u32 offset = (u32)__builtin_amdgcn_readlane((int)hb, (int)(1 + lit)) | ((u32)__builtin_amdgcn_readlane((int)hb, (int)(2 + lit)) << 8);
                    const u32 ml = mlt + 4;
                    // 1 <= offset <= min(bytes written, what the ring serves).  A fast-path sequence writes at most 32 bytes, all after its
                    // reads, so the ring serves it from anywhere in the last LZ_RING - 64 bytes (the chunked general path: LZ_CHUNK) --
                    // C2's column has 15 % of its matches more than 2 KiB back and 5 % more than 4 KiB
                    if (offset - 1 < (op + lit < LZ_RING - 64 ? op + lit : LZ_RING - 64))
                    {
                        // the literals are bytes 1..lit of the window: lane l takes lane l+1's byte (DPP wave shift, no LDS read)
                        const u32 litb = (u32)__builtin_amdgcn_update_dpp(0, (int)hb, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
                        if (lane < lit)
                        {
                            out[op + lane] = (u8)litb;
                            ring[(op + lane) & (LZ_RING - 1)] = (u8)litb;
                        }
                        op += lit;
                        // byte k of the match = byte (k mod offset) of the window before it; uniform branches keep the integer
                        // modulo (~30 instructions) out of the common cases: no overlap, or a power-of-two period (runs, 8-byte values)
                        u32 k = lane;
                        if (offset < ml)
                        {
                            if ((offset & (offset - 1)) == 0)
                                k = lane & (offset - 1);
                            else
                                k = lane % offset;
                        }
                        if (lane < ml)
                        {
                            const u8 v = ring[(op - offset + k) & (LZ_RING - 1)]; // LDS operations of a wave complete in order
                            out[op + lane] = v;
                            ring[(op + lane) & (LZ_RING - 1)] = v;
                        }
                        op += ml;
                        ip += 1 + lit + 2;
                        continue;
                    }
                }
            }
            if (!window(ip, 1))
            {
                bad = true;
                break;
            }
            const u32 token = uni(lin[ip - in_base]);
            ++ip;
            u32 lit = token >> 4;
            if (lit == 15)
            {
                u32 b;
                do
                {
                    if (!window(ip, 1))
                    {
                        bad = true;
                        break;
                    }
                    b = uni(lin[ip - in_base]);
                    ++ip;
                    lit += b;
                } while (b == 255);
                if (bad)
                    break;
            }
            if (lit > isz - ip || lit > osz - op)
            {
                bad = true;
                break;
            }
            if (lit)
            {
                if (ip >= in_base && ip + lit <= in_base + in_len)
                {
                    for (u32 k = lane; k < lit; k += 64) // literals out of the LDS window
                    {
                        const u8 v = lin[ip - in_base + k];
                        out[op + k] = v;
                        ring[(op + k) & (LZ_RING - 1)] = v;
                    }
                }
                else
                {
                    for (u32 k = lane; k < lit; k += 64) // a long literal run: straight from the compressed buffer
                    {
                        const u8 v = in[ip + k];
                        out[op + k] = v;
                        if (lit - k <= LZ_RING) // only the last LZ_RING bytes matter (and each ring byte is written once)
                            ring[(op + k) & (LZ_RING - 1)] = v;
                    }
                }
                ip += lit;
                op += lit;
            }
            if (ip >= isz)
                break; // the last sequence ends after its literals
            if (!window(ip, 2))
            {
                bad = true;
                break;
            }
            const u32 offset = uni((u32)lin[ip - in_base] | ((u32)lin[ip - in_base + 1] << 8));
            ip += 2;
            u32 ml = token & 15;
            if (ml == 15)
            {
                u32 b;
                do
                {
                    if (!window(ip, 1))
                    {
                        bad = true;
                        break;
                    }
                    b = uni(lin[ip - in_base]);
                    ++ip;
                    ml += b;
                } while (b == 255);
                if (bad)
                    break;
            }
            ml += 4;
            if (offset == 0 || offset > op || ml > osz - op)
            {
                bad = true;
                break;
            }
            if (offset <= LZ_CHUNK)
            {
                // from the ring, at most LZ_CHUNK bytes at a time: a chunk never overwrites ring bytes it still has to read, and
                // because the output repeats with period `offset`, the window "offset bytes before the current position" is
                // the right source for every chunk
                while (ml)
                {
                    const u32 c = ml < LZ_CHUNK ? ml : LZ_CHUNK;
                    const u32 wbase = op - offset;
                    if (offset >= c)
                    {
                        for (u32 k = lane; k < c; k += 64)
                        {
                            const u8 v = ring[(wbase + k) & (LZ_RING - 1)];
                            out[op + k] = v;
                            ring[(op + k) & (LZ_RING - 1)] = v;
                        }
                    }
                    else
                    {
                        for (u32 k = lane; k < c; k += 64)
                        {
                            const u8 v = ring[(wbase + k % offset) & (LZ_RING - 1)];
                            out[op + k] = v;
                            ring[(op + k) & (LZ_RING - 1)] = v;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    op += c;
                    ml -= c;
                }
            }
            else
            {
                // a far match: its source left the ring; read the output buffer itself (a wave's memory operations execute in
                // order, so its own earlier stores are visible; the fences only pin the compiler)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const u8 * win = out + (op - offset);
                if (offset >= ml)
                {
                    for (u32 k = lane; k < ml; k += 64)
                    {
                        const u8 v = win[k];
                        out[op + k] = v;
                        if (ml - k <= LZ_RING)
                            ring[(op + k) & (LZ_RING - 1)] = v;
                    }
                }
                else
                {
                    for (u32 k = lane; k < ml; k += 64)
                    {
                        const u8 v = win[k % offset];
                        out[op + k] = v;
                        if (ml - k <= LZ_RING)
                            ring[(op + k) & (LZ_RING - 1)] = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                op += ml;
            }
        }
        if (!bad && op != osz)
            bad = true;
        if (bad && lane == 0)
            atomicOr(err, 2u);
    }
}

// CompressionCodecDelta::doDecompressData (src/Compression/CompressionCodecDelta.cpp:84-175) as the second stage of
// CODEC(Delta, LZ4) (CompressionCodecMultiple.cpp:68-130: the LZ4 stage yields the Delta stage's own 9-byte header + payload):
// payload = [element width][bytes_to_skip][skipped bytes][deltas]; the values are the running sums, wrap-around.  One wave per
// frame: 64 deltas per step, wave inclusive scan by shuffles, the carry travels in a uniform register.  Loads and stores are
// unaligned by construction (9 + 2 header bytes in front of the payload).
template <typename T>
__device__ __forceinline__ void delta_frame(const u8 * __restrict__ pay, u32 n_el, u8 * __restrict__ out, u32 lane)
{
    T carry = 0;
    for (u32 base = 0; base < n_el; base += 64)
    {
        const u32 i = base + lane;
        T v = 0;
        if (i < n_el)
            __builtin_memcpy(&v, pay + (size_t)i * sizeof(T), sizeof(T));
#pragma unroll
        for (int d = 1; d < 64; d <<= 1)
        {
            T o;
            if constexpr (sizeof(T) == 8)
                o = (T)(((u64)__shfl_up((u32)((u64)v >> 32), d, 64) << 32) | __shfl_up((u32)(u64)v, d, 64));
            else
                o = (T)__shfl_up((u32)v, d, 64);
            if (lane >= (u32)d)
                v += o;
        }
        v += carry;
        if (i < n_el)
            __builtin_memcpy(out + (size_t)i * sizeof(T), &v, sizeof(T));
        T last;
        if constexpr (sizeof(T) == 8)
            last = (T)(((u64)__shfl((u32)((u64)v >> 32), 63, 64) << 32) | __shfl((u32)(u64)v, 63, 64));
        else
            last = (T)__shfl((u32)v, 63, 64);
        carry = last; // lanes beyond n_el hold the running sum too (their delta is 0)
    }
}

__global__ __launch_bounds__(256) void k_delta_decode(const u8 * __restrict__ stage, u8 * __restrict__ dst_out, const FrameJob * __restrict__ jobs, u32 n_jobs,
                                                      u32 * __restrict__ err)
{
    const u32 lane = threadIdx.x & 63;
    const u32 wave0 = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    for (u32 j = wave0; j < n_jobs; j += n_waves)
    {
        const FrameJob job = jobs[j];
        if (job.post != 0x92u)
            continue;
        const u8 * in = stage + job.dst_off; // the LZ4 stage's output: [method][compressed size][decompressed size][payload]
        const u32 ssz = job.dst_size;
        bool bad = ssz < 11;
        u32 w = 0, skip = 0, n_bytes = 0;
        if (!bad)
        {
            u32 csize, dsize;
            __builtin_memcpy(&csize, in + 1, 4);
            __builtin_memcpy(&dsize, in + 5, 4);
            w = in[9], skip = in[10];
            bad = in[0] != 0x92 || csize != ssz || dsize != job.out_size || !(w == 1 || w == 2 || w == 4 || w == 8) || skip >= w || 11 + skip > ssz || skip > job.out_size;
            // (bytes_to_skip = size % width, CompressionCodecDelta.cpp:109-133: a header that claims more is malformed -- and the copy
            //  below moves one byte per lane, i.e. at most 63)
            if (!bad)
            {
                n_bytes = ssz - 11 - skip;
                bad = n_bytes != job.out_size - skip || n_bytes % w != 0;
            }
        }
        if (__builtin_amdgcn_readfirstlane((int)bad))
        {
            if (lane == 0)
                atomicOr(err, 4u);
            continue;
        }
        u8 * out = dst_out + job.out_off;
        if (lane < skip)
            out[lane] = in[11 + lane];
        const u8 * pay = in + 11 + skip;
        switch (w)
        {
            case 1: delta_frame<u8>(pay, n_bytes, out + skip, lane); break;
            case 2: delta_frame<u16>(pay, n_bytes / 2, out + skip, lane); break;
            case 4: delta_frame<u32>(pay, n_bytes / 4, out + skip, lane); break;
            default: delta_frame<u64>(pay, n_bytes / 8, out + skip, lane); break;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// CODEC(T64) (src/Compression/CompressionCodecT64.cpp:538-677): payload = [cookie: type magic | variant << 7][min 8 B][max 8 B] and, per 64
// values, num_bits UInt64 of a transposed matrix -- num_bits = the bits in which min and max differ; the 'byte' variant stores whole bytes
// of the values as 64-byte planes and bit-transposes only the last partial byte, the 'bit' variant bit-transposes everything.  A block of
// 64 values is exactly one wavefront: lane = value; a byte plane is one coalesced 64-byte load, a bit row is one 8-byte word the whole
// wave reads (its bit `lane` belongs to this lane).  Upper bits come from min (or max, for the non-negative values of a signed range that
// crosses zero: restoreUpperBits).  One workgroup per frame, its four waves take the frame's blocks in turn.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 t64_bits_u(u64 mn, u64 mx) { const u64 x = mn ^ mx; return x ? 64u - (u32)__builtin_clzll(x) : 0u; }
__device__ __forceinline__ u32 t64_bits_s(i64 mn, i64 mx)
{
    if (mn < 0 && mx >= 0)
        return (mn + mx >= 0) ? t64_bits_u(0, (u64)mx) + 1 : t64_bits_u(0, (u64)~mn) + 1;
    return t64_bits_u((u64)mn, (u64)mx);
}

__global__ __launch_bounds__(256) void k_t64_decode(const u8 * __restrict__ src, const u8 * __restrict__ stage, u8 * __restrict__ dst_out, const FrameJob * __restrict__ jobs, u32 n_jobs,
                                                    u32 * __restrict__ err)
{
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u32 j = blockIdx.x; j < n_jobs; j += gridDim.x)
    {
        const FrameJob job = jobs[j];
        const CodecIo io = codec_io(job, 0x93u, src, stage, dst_out);
        if (!uni(io.take))
            continue;
        const u8 * in = (const u8 *)uni64((u64)io.in);
        u8 * out = (u8 *)uni64((u64)io.out);
        const u32 isz = uni(io.isz), osz = uni(io.osz);
        bool bad = uni(io.bad) || isz < 17;
        u32 width = 0, sgn = 0, full = 0, num_bits = 0;
        u64 mn = 0, mx = 0;
        if (!bad)
        {
            const u32 cookie = in[0], magic = cookie & 0x7Fu;
            full = cookie >> 7;
            switch (magic) // MagicNumber -> base type (CompressionCodecT64.cpp:75-160)
            {
                case 1: width = 1; break;
                case 17: case 6: width = 1; sgn = 1; break;
                case 2: case 13: width = 2; break;
                case 7: case 18: width = 2; sgn = 1; break;
                case 3: case 14: case 21: width = 4; break;
                case 8: case 19: case 22: width = 4; sgn = 1; break;
                case 4: width = 8; break;
                case 9: case 15: case 20: width = 8; sgn = 1; break;
                default: bad = true; break;
            }
            __builtin_memcpy(&mn, in + 1, 8);
            __builtin_memcpy(&mx, in + 9, 8);
        }
        u64 n = 0, num_full = 0;
        u32 tail = 0;
        if (!bad)
        {
            bad = osz % width != 0;
            n = osz / width;
            num_bits = sgn ? t64_bits_s((i64)mn, (i64)mx) : t64_bits_u(mn, mx);
            if (!bad && num_bits)
            {
                const u32 body = isz - 17, shift = 8 * num_bits;
                bad = body == 0 || body % shift != 0;
                num_full = body / shift;
                tail = (u32)(n % 64);
                if (tail && num_full)
                    --num_full;
                bad = bad || num_full * 64 + tail != n;
            }
        }
        if (__builtin_amdgcn_readfirstlane((int)bad))
        {
            if (threadIdx.x == 0)
                atomicOr(err, 8u);
            continue;
        }
        const u64 M = width == 8 ? ~0ull : ((1ull << (8 * width)) - 1);
        if (!num_bits)
        {
            for (u64 i = threadIdx.x; i < n; i += 256)
                __builtin_memcpy(out + i * width, &mn, width); // every value equals min
            continue;
        }
        u64 upper_min = 0, upper_max = 0, sign_bit = 0;
        if (num_bits < 64)
            upper_min = (mn >> num_bits << num_bits) & M;
        if (sgn && (i64)mn < 0 && (i64)mx >= 0 && num_bits < 64)
        {
            sign_bit = (1ull << (num_bits - 1)) & M;
            upper_max = (mx >> num_bits << num_bits) & M;
        }
        const u32 full_bytes = num_bits / 8, part_bits = num_bits % 8;
        const u64 blocks = num_full + (tail ? 1 : 0);
        const u8 * body = in + 17;
        for (u64 b = wave; b < blocks; b += 4)
        {
            const u8 * blk = body + b * 8 * (u64)num_bits;
            const u32 cnt = b == num_full ? tail : 64u;
            u64 v = 0;
            u32 first_row = 0;
            if (!full)
            {
                for (u32 k = 0; k < full_bytes; ++k)
                    v |= (u64)blk[64 * k + lane] << (8 * k);
                first_row = 8 * full_bytes;
            }
            const u32 n_rows = full ? num_bits : part_bits;
            for (u32 r = 0; r < n_rows; ++r)
            {
                u64 row;
                __builtin_memcpy(&row, blk + 8 * (u64)(first_row + r), 8); // the same address in every lane: one broadcast load
                v |= ((row >> lane) & 1ull) << (first_row + r);
            }
            if (sign_bit)
                v |= (v & sign_bit) ? upper_min : upper_max;
            else
                v |= upper_min;
            if (lane < cnt)
                __builtin_memcpy(out + (b * 64 + lane) * width, &v, width);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// CODEC(DoubleDelta) (src/Compression/CompressionCodecDoubleDelta.cpp:367-447): [width][bytes_to_skip][skipped][items u32][first value]
// [first delta] and a bit stream of prefix-coded double deltas (0 | 10 s x6 | 110 s x8 | 1110 s x11 | 11110 s x31 | 11111 s x63, big-endian
// bit order, BitHelpers.h).  A value's position in the stream depends on every code before it and its value on the two values before it:
// the frame is decoded front to back, here by ONE LANE PER FRAME -- the parallelism is across frames (a column file of N rows has
// N / 8192 ... N / 131072 of them).  Functional first: the rate is reported in profiles/, not tuned.
// ---------------------------------------------------------------------------------------------
// one value of `width` bytes stored with ONE instruction (a memcpy of a run-time width is a byte loop)
__device__ __forceinline__ void codec_store(u8 * d, u64 v, u32 width)
{
    typedef u64 u64_a1 __attribute__((aligned(1)));
    typedef u32 u32_a1 __attribute__((aligned(1)));
    typedef u16 u16_a1 __attribute__((aligned(1)));
    if (width == 8)
        *(u64_a1 *)d = v;
    else if (width == 4)
        *(u32_a1 *)d = (u32)v;
    else if (width == 2)
        *(u16_a1 *)d = (u16)v;
    else
        *d = (u8)v;
}

struct DdReader
{
    // A lane is alone with its frame: every load is a full memory round trip with nothing else to hide it, so the reader keeps DEPTH
    // 8-byte words of the stream in flight (q[0] the next to be used; a word is requested DEPTH appends before it is needed) and a
    // 128-bit window {hi, lo} of `bits` valid bits in front of them.  Big-endian bit order (BitHelpers.h): the stream's first bit is
    // the top bit of hi.
    static constexpr int DEPTH = 6;
    const u8 * cur; // the next byte not yet requested
    const u8 * end;
    u64 hi, lo;
    u32 bits;
    u64 q[DEPTH];
    u32 qn; // words in the queue
    typedef u64 u64_a1 __attribute__((aligned(1)));
    // (every index into q[] is a compile-time constant: a run-time index would move the queue to scratch memory)
    __device__ __forceinline__ u64 request()
    {
        if (cur + 8 <= end)
        {
            const u64 w = *(const u64_a1 *)cur; // (byte-swapped when it is used: the swap would wait for the load)
            cur += 8;
            ++qn;
            return w;
        }
        return 0;
    }
    __device__ __forceinline__ void init(const u8 * s, const u8 * e)
    {
        cur = s, end = e, hi = lo = 0, bits = 0, qn = 0;
#pragma unroll
        for (int k = 0; k < DEPTH; ++k)
            q[k] = request(); // (whole words run out only at the end: the valid ones are always the first qn)
    }
    __device__ __forceinline__ void append(u64 w, u32 nb) // nb valid bits at the top of w behind the window (bits + nb <= 128)
    {
        if (bits == 0)
            hi = w, lo = 0;
        else if (bits < 64)
            hi |= w >> bits, lo = w << (64 - bits);
        else if (bits == 64)
            lo = w;
        else
            lo |= w >> (bits - 64);
        bits += nb;
    }
    __device__ __forceinline__ void fill() // at least 64 valid bits while the stream has them
    {
        if (bits > 64)
            return;
        if (qn)
        {
            append(__builtin_bswap64(q[0]), 64);
#pragma unroll
            for (int k = 0; k + 1 < DEPTH; ++k)
                q[k] = q[k + 1];
            --qn;
            q[DEPTH - 1] = request();
            return;
        }
        // the last < 8 bytes of the stream, one by one
        u64 w = 0;
        u32 nb = 0;
        while (cur < end && nb < 64)
        {
            w |= (u64)*cur++ << (56 - nb);
            nb += 8;
        }
        if (nb)
            append(w, nb);
    }
    __device__ __forceinline__ u64 read32(u32 n) // n <= 32
    {
        if (n == 0)
            return 0;
        if (bits < n)
            fill();
        const u64 v = hi >> (64 - n);
        hi = (hi << n) | (lo >> (64 - n));
        lo <<= n;
        bits = bits >= n ? bits - n : 0;
        return v;
    }
    __device__ __forceinline__ u64 read(u32 n) { return n > 32 ? (read32(n - 32) << 32) | read32(32) : read32(n); }
    __device__ __forceinline__ u32 peek5() // the next five bits (zeros past the end)
    {
        if (bits < 8)
            fill();
        return (u32)(hi >> 59);
    }
    __device__ __forceinline__ bool eof()
    {
        return bits == 0 && qn == 0 && cur >= end;
    }
};

__global__ __launch_bounds__(64) void k_double_delta_decode(const u8 * __restrict__ src, const u8 * __restrict__ stage, u8 * __restrict__ dst_out, const FrameJob * __restrict__ jobs,
                                                            u32 n_jobs, u32 * __restrict__ err)
{
    const u32 j = blockIdx.x * 64 + threadIdx.x;
    if (j >= n_jobs)
        return;
    const FrameJob job = jobs[j];
    const CodecIo io = codec_io(job, 0x94u, src, stage, dst_out);
    if (!io.take)
        return;
    const u8 * in = io.in;
    u8 * out = io.out;
    const u32 isz = io.isz, osz = io.osz;
    if (io.bad || isz < 2)
    {
        atomicOr(err, 16u);
        return;
    }
    const u32 width = in[0], skip = in[1];
    if (!(width == 1 || width == 2 || width == 4 || width == 8) || skip >= width || skip > osz || 2 + skip > isz)
    {
        atomicOr(err, 16u);
        return;
    }
    for (u32 k = 0; k < skip; ++k)
        out[k] = in[2 + k];
    const u8 * s = in + 2 + skip, * s_end = in + isz;
    u8 * d = out + skip, * d_end = out + osz;
    if (s + 4 > s_end)
        return;
    u32 items;
    __builtin_memcpy(&items, s, 4);
    s += 4;
    const u64 M = width == 8 ? ~0ull : ((1ull << (8 * width)) - 1);
    if (s + width > s_end || items < 1)
        return;
    u64 prev_value = 0, prev_delta = 0;
    __builtin_memcpy(&prev_value, s, width);
    if (d + width > d_end)
    {
        atomicOr(err, 16u);
        return;
    }
    __builtin_memcpy(d, &prev_value, width);
    s += width;
    d += width;
    if (s + width > s_end || items < 2)
        return;
    __builtin_memcpy(&prev_delta, s, width);
    prev_value = (prev_value + prev_delta) & M;
    if (d + width > d_end)
    {
        atomicOr(err, 16u);
        return;
    }
    __builtin_memcpy(d, &prev_value, width);
    s += width;
    d += width;
    DdReader r;
    r.init(s, s_end);
    for (u32 read = 2; read < items && !r.eof(); ++read)
    {
        const u32 top = r.peek5(); // the five bits peekByte() >> 3 looks at (WRITE_SPEC_LUT)
        u32 pbits, dbits;
        if (top < 16) { pbits = 1; dbits = 0; }
        else if (top < 24) { pbits = 2; dbits = 7; }
        else if (top < 28) { pbits = 3; dbits = 9; }
        else if (top < 30) { pbits = 4; dbits = 12; }
        else if (top == 30) { pbits = 5; dbits = 32; }
        else { pbits = 5; dbits = 64; }
        (void)r.read32(pbits);
        u64 dd = 0;
        if (dbits)
        {
            const u64 sign = r.read32(1);
            dd = (r.read(dbits - 1) + 1) & M;
            if (sign)
                dd = (0 - dd) & M;
        }
        const u64 delta = (dd + prev_delta) & M, cur = (prev_value + delta) & M;
        if (d + width > d_end)
        {
            atomicOr(err, 16u);
            return;
        }
        codec_store(d, cur, width);
        d += width;
        prev_delta = (cur - prev_value) & M;
        prev_value = cur;
    }
}

// CODEC(Gorilla) (src/Compression/CompressionCodecGorilla.cpp:268-330): [width][bytes_to_skip][skipped][items u32][first value] and a bit
// stream of XOR differences -- 0: the value repeats | 10: the meaningful bits, inside the previous window of leading / trailing zeros |
// 11: leading zeros (W - 1 bits), length (W bits), the meaningful bits; W = 4 / 5 / 6 / 7 for 1 / 2 / 4 / 8-byte values.  Like DoubleDelta
// a front-to-back code: one lane per frame.
__global__ __launch_bounds__(64) void k_gorilla_decode(const u8 * __restrict__ src, const u8 * __restrict__ stage, u8 * __restrict__ dst_out, const FrameJob * __restrict__ jobs, u32 n_jobs,
                                                       u32 * __restrict__ err)
{
    const u32 j = blockIdx.x * 64 + threadIdx.x;
    if (j >= n_jobs)
        return;
    const FrameJob job = jobs[j];
    const CodecIo io = codec_io(job, 0x95u, src, stage, dst_out);
    if (!io.take)
        return;
    const u8 * in = io.in;
    u8 * out = io.out;
    const u32 isz = io.isz, osz = io.osz;
    if (io.bad || isz < 2)
    {
        atomicOr(err, 32u);
        return;
    }
    const u32 width = in[0], skip = in[1];
    if (!(width == 1 || width == 2 || width == 4 || width == 8) || skip >= width || skip > osz || 2 + skip > isz)
    {
        atomicOr(err, 32u);
        return;
    }
    for (u32 k = 0; k < skip; ++k)
        out[k] = in[2 + k];
    const u32 X = 8 * width, DBL = width == 1 ? 4u : width == 2 ? 5u : width == 4 ? 6u : 7u, LZL = DBL - 1;
    const u8 * s = in + 2 + skip, * s_end = in + isz;
    u8 * d = out + skip;
    if (s + 4 > s_end)
        return;
    u32 items;
    __builtin_memcpy(&items, s, 4);
    s += 4;
    if (s + width > s_end || items < 1)
        return;
    if ((u64)items * width > osz - skip)
    {
        atomicOr(err, 32u);
        return;
    }
    const u64 M = width == 8 ? ~0ull : ((1ull << X) - 1);
    u64 prev = 0;
    __builtin_memcpy(&prev, s, width);
    __builtin_memcpy(d, &prev, width);
    s += width;
    d += width;
    DdReader r;
    r.init(s, s_end);
    u32 p_lz = 0, p_db = 0, p_tz = 0;
    for (u32 read = 1; read < items && !r.eof(); ++read)
    {
        u64 cur = prev;
        u32 lz = p_lz, db = p_db, tz = p_tz;
        if (r.read32(1) == 1)
        {
            if (r.read32(1) == 1)
            {
                lz = (u32)r.read32(LZL);
                db = (u32)r.read32(DBL);
                tz = (X - lz - db) & 0xFFu; // (UInt8 arithmetic in the reference: a corrupted length wraps there too)
            }
            if (lz == 0 && db == 0 && tz == 0)
            {
                atomicOr(err, 32u);
                return;
            }
            u64 x = db > 64 ? 0 : (db == 64 ? ((r.read32(32) << 32) | r.read32(32)) : r.read(db));
            x = tz >= 64 ? 0 : x << tz;
            cur = (prev ^ x) & M;
        }
        codec_store(d, cur, width);
        d += width;
        p_lz = lz, p_db = db, p_tz = tz;
        prev = cur;
    }
}

/* Decompress n_frames frames of `compressed_u8` into one new UInt8 column of sum(decompressed_sizes) bytes.  Host arrays describe the
   frames: payload offset / size inside compressed_u8 and method byte of the (last applied) general-purpose stage, its output size
   (stage_sizes; NULL = decompressed_sizes) and, for CODEC(Delta, LZ4), post_methods[f] = 0x92 (NULL / 0 = single stage).
   CANNOT_DECOMPRESS -> CHGPU_ERR_BAD_ARGUMENTS. */
extern "C" int chgpu_decompress_frames(chgpu_ctx * ctx, const chgpu_col * compressed_u8, uint32_t n_frames, const uint64_t * payload_offsets,
                                       const uint32_t * payload_sizes, const uint32_t * decompressed_sizes, const uint8_t * methods,
                                       const uint8_t * post_methods, const uint32_t * stage_sizes, chgpu_col ** out_u8)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && compressed_u8 && out_u8, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    CHGPU_REQUIRE(compressed_u8->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS, "compressed data is a UInt8 column");
    CHGPU_REQUIRE(n_frames == 0 || (payload_offsets && payload_sizes && decompressed_sizes && methods), CHGPU_ERR_BAD_ARGUMENTS, "NULL frame arrays");
    std::vector<FrameJob> jobs(n_frames);
    u64 total = 0, stage_total = 0;
    u32 n_t64 = 0, n_dd = 0, n_gor = 0, p_delta = 0, p_t64 = 0, p_dd = 0, p_gor = 0;
    for (u32 f = 0; f < n_frames; ++f)
    {
        CHGPU_REQUIRE(methods[f] == 0x82 || methods[f] == 0x02 || methods[f] == 0x93 || methods[f] == 0x94 || methods[f] == 0x95, CHGPU_ERR_NOT_IMPLEMENTED,
                      "compression method 0x%02x: CPU path", methods[f]);
        n_t64 += methods[f] == 0x93;
        n_dd += methods[f] == 0x94;
        n_gor += methods[f] == 0x95;
        CHGPU_REQUIRE(payload_offsets[f] + payload_sizes[f] <= compressed_u8->rows, CHGPU_ERR_BAD_ARGUMENTS, "frame %u lies outside the compressed buffer", f);
        const u32 post = post_methods ? post_methods[f] : 0;
        CHGPU_REQUIRE(post == 0 || (post >= 0x92 && post <= 0x95), CHGPU_ERR_NOT_IMPLEMENTED, "codec 0x%02x in front of the general-purpose stage: CPU path", post);
        CHGPU_REQUIRE(post == 0 || methods[f] == 0x82 || methods[f] == 0x02, CHGPU_ERR_NOT_IMPLEMENTED, "a codec stage behind method 0x%02x: CPU path", methods[f]);
        CHGPU_REQUIRE(post == 0 || stage_sizes, CHGPU_ERR_BAD_ARGUMENTS, "stage_sizes is NULL");
        p_delta += post == 0x92;
        p_t64 += post == 0x93;
        p_dd += post == 0x94;
        p_gor += post == 0x95;
        FrameJob jb{};
        jb.src_off = payload_offsets[f];
        jb.src_size = payload_sizes[f];
        jb.method = methods[f];
        jb.post = post;
        if (post)
        {
            jb.dst_off = stage_total;
            jb.dst_size = stage_sizes[f];
            jb.out_off = total;
            jb.out_size = decompressed_sizes[f];
            stage_total += (stage_sizes[f] + 15u) & ~15ull;
        }
        else
        {
            jb.dst_off = total;
            jb.dst_size = decompressed_sizes[f];
        }
        jobs[f] = jb;
        total += decompressed_sizes[f];
    }
    chgpu_col * res = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, CHGPU_U8, total, &res));
    if (n_frames && total)
    {
        void * stage = nullptr;
        size_t stage_class = 0;
        void * scratch = nullptr;
        int rc = chgpu_scratch(ctx, sizeof(FrameJob) * (size_t)n_frames + 256, &scratch);
        if (rc == CHGPU_OK && stage_total)
            rc = chgpu_pool_alloc(ctx, stage_total + 64, &stage, &stage_class);
        if (rc != CHGPU_OK)
        {
            chgpu_col_free(res);
            return rc;
        }
        u32 * err = (u32 *)scratch;
        FrameJob * jd = (FrameJob *)((char *)scratch + 256);
        hipError_t e = hipMemsetAsync(err, 0, 256, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(jd, jobs.data(), sizeof(FrameJob) * (size_t)n_frames, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream); // `jobs` is pageable host memory: it must outlive the copy
        if (e == hipSuccess)
        {
            const u32 grid = chgpu_grid_for(ctx, (u64)n_frames * 64, 256, 8);
            hipLaunchKernelGGL(k_lz4_decode, dim3(grid), dim3(256), 0, ctx->stream, (const u8 *)compressed_u8->data, (u8 *)res->data, (u8 *)stage, (const FrameJob *)jd,
                               n_frames, err);
            ctx->counters[6] += 1;
            if (n_t64)
            {
                hipLaunchKernelGGL(k_t64_decode, dim3(n_frames < 4096 ? n_frames : 4096), dim3(256), 0, ctx->stream, (const u8 *)compressed_u8->data, (const u8 *)nullptr, (u8 *)res->data,
                                   (const FrameJob *)jd, n_frames, err);
                ctx->counters[6] += 1;
            }
            if (n_dd)
            {
                hipLaunchKernelGGL(k_double_delta_decode, dim3((n_frames + 63) / 64), dim3(64), 0, ctx->stream, (const u8 *)compressed_u8->data, (const u8 *)nullptr, (u8 *)res->data,
                                   (const FrameJob *)jd, n_frames, err);
                ctx->counters[6] += 1;
            }
            if (n_gor)
            {
                hipLaunchKernelGGL(k_gorilla_decode, dim3((n_frames + 63) / 64), dim3(64), 0, ctx->stream, (const u8 *)compressed_u8->data, (const u8 *)nullptr, (u8 *)res->data,
                                   (const FrameJob *)jd, n_frames, err);
                ctx->counters[6] += 1;
            }
            // the codec stages behind a general-purpose stage read its output in the stage buffer
            if (p_delta)
            {
                hipLaunchKernelGGL(k_delta_decode, dim3(grid), dim3(256), 0, ctx->stream, (const u8 *)stage, (u8 *)res->data, (const FrameJob *)jd, n_frames, err);
                ctx->counters[6] += 1;
            }
            if (p_t64)
            {
                hipLaunchKernelGGL(k_t64_decode, dim3(n_frames < 4096 ? n_frames : 4096), dim3(256), 0, ctx->stream, (const u8 *)compressed_u8->data, (const u8 *)stage, (u8 *)res->data,
                                   (const FrameJob *)jd, n_frames, err);
                ctx->counters[6] += 1;
            }
            if (p_dd)
            {
                hipLaunchKernelGGL(k_double_delta_decode, dim3((n_frames + 63) / 64), dim3(64), 0, ctx->stream, (const u8 *)compressed_u8->data, (const u8 *)stage, (u8 *)res->data,
                                   (const FrameJob *)jd, n_frames, err);
                ctx->counters[6] += 1;
            }
            if (p_gor)
            {
                hipLaunchKernelGGL(k_gorilla_decode, dim3((n_frames + 63) / 64), dim3(64), 0, ctx->stream, (const u8 *)compressed_u8->data, (const u8 *)stage, (u8 *)res->data,
                                   (const FrameJob *)jd, n_frames, err);
                ctx->counters[6] += 1;
            }
            e = hipGetLastError();
        }
        u32 failed = 0;
        if (e == hipSuccess)
            rc = chgpu_read_back(ctx, err, &failed, sizeof(failed));
        if (stage)
            chgpu_pool_free(ctx, stage, stage_class);
        if (e != hipSuccess)
            rc = chgpu_set_error(CHGPU_ERR_DEVICE, "frame decode: %s", hipGetErrorString(e));
        else if (rc == CHGPU_OK && failed)
            rc = chgpu_set_error(CHGPU_ERR_BAD_ARGUMENTS, "Cannot decompress: malformed frame (CANNOT_DECOMPRESS)");
        if (rc != CHGPU_OK)
        {
            chgpu_col_free(res);
            return rc;
        }
    }
    *out_u8 = res;
    return CHGPU_OK;
}

/* typed view of `rows` elements starting at byte `byte_offset` of a UInt8 column, copied into a new aligned column (decompressed
   column files are plain little-endian arrays: SerializationNumber::deserializeBinaryBulk) */
__global__ __launch_bounds__(256) void k_bytes_copy(const u8 * __restrict__ src, u64 nbytes, u8 * __restrict__ dst)
{
    const u64 stride = (u64)gridDim.x * 256;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0)
    {
        typedef u32 v4u __attribute__((ext_vector_type(4)));
        const u64 nv = nbytes / 16;
        for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride)
            ((v4u *)dst)[i] = __builtin_nontemporal_load((const v4u *)src + i);
        for (u64 i = nv * 16 + (u64)blockIdx.x * 256 + threadIdx.x; i < nbytes; i += stride)
            dst[i] = src[i];
    }
    else
        for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nbytes; i += stride)
            dst[i] = src[i];
}

extern "C" int chgpu_col_from_bytes(chgpu_ctx * ctx, const chgpu_col * bytes_u8, uint64_t byte_offset, int type, uint64_t rows, chgpu_col ** out)
{
    ChgpuDeviceGuard _dev_guard(ctx);
    CHGPU_REQUIRE(ctx && bytes_u8 && out, CHGPU_ERR_BAD_ARGUMENTS, "NULL argument");
    const size_t es = chgpu_type_size(type);
    CHGPU_REQUIRE(es && bytes_u8->type == CHGPU_U8, CHGPU_ERR_BAD_ARGUMENTS, "bad type");
    CHGPU_REQUIRE(byte_offset + rows * es <= bytes_u8->rows, CHGPU_ERR_SIZES_MISMATCH, "Cannot read all data: %llu rows of %zu bytes at %llu, %llu available",
                  (unsigned long long)rows, es, (unsigned long long)byte_offset, (unsigned long long)bytes_u8->rows);
    chgpu_col * res = nullptr;
    CHGPU_TRY(chgpu_col_new(ctx, type, rows, &res));
    if (rows)
    {
        const u64 nbytes = rows * es;
        hipLaunchKernelGGL(k_bytes_copy, dim3(chgpu_grid_for(ctx, (nbytes + 15) / 16, 256, 8)), dim3(256), 0, ctx->stream,
                           (const u8 *)bytes_u8->data + byte_offset, nbytes, (u8 *)res->data);
        ctx->counters[6] += 1;
    }
    *out = res;
    return CHGPU_OK;
}
