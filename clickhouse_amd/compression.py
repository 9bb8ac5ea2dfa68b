"""The feed side: compressed column data decoded in HBM.

Mirror of CompressedReadBuffer (src/Compression/CompressedReadBufferBase.cpp:175-222) + SerializationNumber::deserializeBinaryBulk:
the host walks the frame headers, the compressed bytes cross PCIe once, every frame is decompressed by one wavefront
(csrc/compress_kernels.hip) and the result is viewed as a typed column.  No CPU decompression in this path.
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import _capi as K
from .columns import TAG_OF, Column, Context

CHECKSUM_SIZE, HEADER_SIZE = 16, 9  # CompressionInfo.h:10


METHOD_NONE, METHOD_LZ4, METHOD_MULTIPLE, METHOD_DELTA = 0x02, 0x82, 0x91, 0x92


def parse_frames(buf) -> list:
    """frame headers of a compressed buffer -> [(method, payload_offset, payload_size, decompressed_size, post_method, stage_size)]:
    the general-purpose stage to run on the device and, for CODEC(Delta, LZ4), the Delta stage behind it (post_method 0x92;
    stage_size = bytes the LZ4 stage yields).  Any other codec chain keeps its method byte: the device call answers NOT_IMPLEMENTED."""
    mv = memoryview(buf)
    frames, pos, n = [], 0, len(mv)
    while pos < n:
        if n - pos < CHECKSUM_SIZE + HEADER_SIZE:
            raise K.ChgpuError(K.ERR_BAD_ARGUMENTS, "Cannot read all data: truncated frame header")
        method, csize, dsize = struct.unpack_from("<BII", mv, pos + CHECKSUM_SIZE)
        if csize < HEADER_SIZE or pos + CHECKSUM_SIZE + csize > n:
            raise K.ChgpuError(K.ERR_BAD_ARGUMENTS, "Cannot decompress: frame size out of range")
        off, size = pos + CHECKSUM_SIZE + HEADER_SIZE, csize - HEADER_SIZE
        post, stage = 0, dsize
        if method == METHOD_MULTIPLE and size >= 1:  # CompressionCodecMultiple.cpp:68-130: [n][methods...][last stage: header + payload]
            k = mv[off]
            methods = bytes(mv[off + 1:off + 1 + k])
            if methods == bytes([METHOD_DELTA, METHOD_LZ4]) and size >= 1 + k + HEADER_SIZE:
                m2, c2, d2 = struct.unpack_from("<BII", mv, off + 1 + k)
                if m2 != METHOD_LZ4 or c2 < HEADER_SIZE or 1 + k + c2 > size:
                    raise K.ChgpuError(K.ERR_BAD_ARGUMENTS, "Cannot decompress: bad stage header in codec Multiple")
                method, post, stage = METHOD_LZ4, METHOD_DELTA, d2
                off, size = off + 1 + k + HEADER_SIZE, c2 - HEADER_SIZE
        frames.append((method, off, size, dsize, post, stage))
        pos += CHECKSUM_SIZE + csize
    return frames


def decompress_frames(ctx: Context, compressed: Column, frames) -> Column:
    n = len(frames)
    frames = [tuple(f) if len(f) >= 6 else tuple(f[:4]) + (0, f[3]) for f in frames]
    offs = (C.c_uint64 * max(1, n))(*[f[1] for f in frames])
    sizes = (C.c_uint32 * max(1, n))(*[f[2] for f in frames])
    dsizes = (C.c_uint32 * max(1, n))(*[f[3] for f in frames])
    methods = (C.c_uint8 * max(1, n))(*[f[0] for f in frames])
    posts = (C.c_uint8 * max(1, n))(*[f[4] for f in frames])
    stages = (C.c_uint32 * max(1, n))(*[f[5] for f in frames])
    h = C.c_void_p()
    K.check(K.lib().chgpu_decompress_frames(ctx._h, compressed._h, n, offs, sizes, dsizes, methods, posts, stages, C.byref(h)))
    return Column(ctx, h)


def column_from_bytes(data: Column, byte_offset: int, dtype, rows: int) -> Column:
    h = C.c_void_p()
    K.check(K.lib().chgpu_col_from_bytes(data.ctx._h, data._h, byte_offset, TAG_OF[np.dtype(dtype)], rows, C.byref(h)))
    return Column(data.ctx, h)


def read_column_file(ctx: Context, buf, dtype) -> Column:
    """a MergeTree `<column>.bin` of a numeric column (compressed frames of a plain little-endian array) -> Column in HBM"""
    frames = parse_frames(buf)
    compressed = ctx.upload(np.frombuffer(buf, dtype=np.uint8))
    raw = decompress_frames(ctx, compressed, frames)
    es = np.dtype(dtype).itemsize
    if raw.size() % es:
        raise K.ChgpuError(K.ERR_SIZES_MISMATCH, "Cannot read all data: size is not a multiple of the element size")
    return column_from_bytes(raw, 0, dtype, raw.size() // es)
