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


def parse_frames(buf, verify_checksums: bool = True) -> list:
    """frame headers of a compressed buffer -> [(method, payload_offset, payload_size, decompressed_size, post_method, stage_size)]:
    the general-purpose stage to run on the device and, for CODEC(Delta, LZ4), the Delta stage behind it (post_method 0x92;
    stage_size = bytes the LZ4 stage yields).  Any other codec chain keeps its method byte: the device call answers NOT_IMPLEMENTED.
    The walk is chgpu_compressed_walk_frames: every frame's CityHash128 checksum is verified (verify_checksums=False = the reference's
    disable_checksum), sizes above 1 GiB are refused (CompressedReadBufferBase.cpp:49-127,163-172)."""
    raw = bytes(buf) if not isinstance(buf, (bytes, bytearray)) else buf
    arr = (C.c_uint8 * len(raw)).from_buffer_copy(raw) if len(raw) else None
    n = C.c_uint32(0)
    K.check(K.lib().chgpu_compressed_walk_frames(arr, len(raw), int(verify_checksums), 0, C.byref(n), None, None, None, None, None, None))
    cap = max(1, n.value)
    offs, sizes, dsizes = (C.c_uint64 * cap)(), (C.c_uint32 * cap)(), (C.c_uint32 * cap)()
    methods, posts, stages = (C.c_uint8 * cap)(), (C.c_uint8 * cap)(), (C.c_uint32 * cap)()
    K.check(K.lib().chgpu_compressed_walk_frames(arr, len(raw), 0, cap, C.byref(n), offs, sizes, dsizes, methods, posts, stages))
    return [(methods[f], offs[f], sizes[f], dsizes[f], posts[f], stages[f]) for f in range(n.value)]


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


def read_column_file(ctx: Context, buf, dtype, verify_checksums: bool = True) -> Column:
    """a MergeTree `<column>.bin` of a numeric column (compressed frames of a plain little-endian array) -> Column in HBM
    (chgpu_read_compressed_column: frame walk + checksum verification on the host, decode on the device)"""
    h = C.c_void_p()
    if isinstance(buf, np.ndarray):  # the file as it lies in host memory (no copy: a column file is hundreds of megabytes)
        view = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
        K.check(K.lib().chgpu_read_compressed_column(ctx._live(), C.c_void_p(view.ctypes.data), view.shape[0], TAG_OF[np.dtype(dtype)], int(verify_checksums), C.byref(h)))
        return Column(ctx, h)
    raw = bytes(buf) if not isinstance(buf, (bytes, bytearray)) else buf
    arr = (C.c_uint8 * len(raw)).from_buffer_copy(raw) if len(raw) else None
    K.check(K.lib().chgpu_read_compressed_column(ctx._live(), arr, len(raw), TAG_OF[np.dtype(dtype)], int(verify_checksums), C.byref(h)))
    return Column(ctx, h)


def city_hash128(data: bytes):
    """CityHash_v1_0_2::CityHash128 -> (low64, high64): the checksum a writer puts in front of a frame"""
    out = (C.c_uint64 * 2)()
    K.check(K.lib().chgpu_city_hash128(bytes(data), len(data), out))
    return int(out[0]), int(out[1])


class NativeColumn:
    """one column of a Native block in HBM.  kind "numeric": `values`; "fixed_string": `values` = rows x N bytes (UInt8) and `fixed_n`;
    "string": `offsets`, `chars` (ColumnString); "lc_string": `indexes` (UInt8/16/32/64) + the dictionary `offsets`, `chars` (key 0 is NULL when
    `nullable`).  `null_map`: the UInt8 map of a Nullable column (None otherwise)."""

    def __init__(self, name, type_name, kind):
        self.name, self.type_name, self.kind = name, type_name, kind
        self.values = self.offsets = self.chars = self.indexes = self.null_map = None
        self.fixed_n, self.nullable, self.num_keys = 0, False, 0

    def strings(self):
        """the values as a list of bytes (None for NULL): host-side, for tests and small results"""
        if self.kind == "numeric":
            raise TypeError("numeric column")
        if self.kind == "fixed_string":
            raw = self.values.numpy().tobytes()
            vals = [raw[i:i + self.fixed_n] for i in range(0, len(raw), self.fixed_n)]
        else:
            offs = self.offsets.numpy() if self.offsets.size() else np.zeros(0, dtype=np.uint64)
            chars = self.chars.numpy().tobytes() if self.chars.size() else b""
            keys = [chars[(int(offs[i - 1]) if i else 0):int(offs[i]) - 1] for i in range(offs.shape[0])]
            if self.kind == "string":
                vals = keys
            else:
                if self.nullable and keys:
                    keys[0] = None
                vals = [keys[int(i)] for i in self.indexes.numpy()]
        if self.null_map is not None:
            nm = self.null_map.numpy()
            vals = [None if nm[i] else v for i, v in enumerate(vals)]
        return vals


class _NativeColumnStruct(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("type_name", C.c_char * 64), ("type", C.c_int32), ("kind", C.c_int32), ("fixed_n", C.c_uint32), ("is_nullable", C.c_int32),
                ("data_offset", C.c_uint64), ("data_bytes", C.c_uint64), ("null_map_offset", C.c_uint64), ("chars_bytes", C.c_uint64),
                ("lc_num_keys", C.c_uint64), ("lc_keys_offset", C.c_uint64), ("lc_keys_bytes", C.c_uint64), ("lc_keys_chars_bytes", C.c_uint64)]


NATIVE_KINDS = {0: "numeric", 1: "string", 2: "fixed_string", 3: "lc_string"}


def _read_strings(ctx: Context, view: bytes, offset: int, nbytes: int, rows: int):
    arr = (C.c_uint8 * max(1, nbytes)).from_buffer_copy(view[offset:offset + nbytes] or b"\0")
    oh, ch_ = C.c_void_p(), C.c_void_p()
    K.check(K.lib().chgpu_native_read_strings(ctx._live(), arr, nbytes, rows, C.byref(oh), C.byref(ch_)))
    return Column(ctx, oh), Column(ctx, ch_)


def read_native_block(ctx: Context, buf, pos: int = 0, server_revision: int = 0, described: bool = False):
    """NativeReader::read for one block: -> (dict(rows, bucket_num, is_overflows, next_pos), [(name, Column)]) for blocks of plain numeric
    columns, or with described=True [(name, NativeColumn)] for every type the walk carries (numbers, String, FixedString(N), Nullable of
    them, LowCardinality(String)).  The header walk is chgpu_native_walk_block; the values go to HBM as they lie in the buffer, Strings
    through chgpu_native_read_strings."""
    from .columns import NP_OF
    raw = bytes(buf)
    view = raw[pos:]
    arr = (C.c_uint8 * len(view)).from_buffer_copy(view) if len(view) else None
    ncols, nrows, used = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
    bucket, over = C.c_int32(-1), C.c_int(0)
    K.check(K.lib().chgpu_native_walk_block(arr, len(view), server_revision, 0, None, C.byref(ncols), C.byref(nrows), C.byref(bucket), C.byref(over), C.byref(used)))
    cols = (_NativeColumnStruct * max(1, ncols.value))()
    K.check(K.lib().chgpu_native_walk_block(arr, len(view), server_revision, ncols.value, cols, C.byref(ncols), C.byref(nrows), C.byref(bucket), C.byref(over), C.byref(used)))
    out = []
    rows = int(nrows.value)
    for c in range(ncols.value):
        d = cols[c]
        nc = NativeColumn(d.name.decode(), d.type_name.decode(), NATIVE_KINDS[d.kind])
        nc.nullable, nc.fixed_n, nc.num_keys = bool(d.is_nullable), int(d.fixed_n), int(d.lc_num_keys)
        if nc.kind == "numeric":
            nc.values = ctx.upload(np.frombuffer(view, dtype=np.dtype(NP_OF[d.type]), count=rows, offset=d.data_offset))
        elif nc.kind == "fixed_string":
            nc.values = ctx.upload(np.frombuffer(view, dtype=np.uint8, count=rows * nc.fixed_n, offset=d.data_offset))
        elif nc.kind == "string":
            nc.offsets, nc.chars = _read_strings(ctx, view, d.data_offset, d.data_bytes, rows)
        else:
            nc.offsets, nc.chars = _read_strings(ctx, view, d.lc_keys_offset, d.lc_keys_bytes, nc.num_keys)
            nc.indexes = ctx.upload(np.frombuffer(view, dtype=np.dtype(NP_OF[d.type]), count=rows, offset=d.data_offset))
        if nc.nullable and nc.kind != "lc_string" and rows:
            nc.null_map = ctx.upload(np.frombuffer(view, dtype=np.uint8, count=rows, offset=d.null_map_offset))
        if not described and (nc.kind != "numeric" or nc.nullable):
            raise K.ChgpuError(K.ERR_NOT_IMPLEMENTED, f"column {nc.name} has type {nc.type_name}: read_native_block(described=True)")
        out.append((nc.name, nc if described else nc.values))
    return dict(rows=rows, bucket_num=int(bucket.value), is_overflows=bool(over.value), next_pos=pos + int(used.value)), out
