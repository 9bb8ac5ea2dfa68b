"""TEST INFRASTRUCTURE ONLY — the reference's compressed-frame layout and codecs restated for the feed path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Frame (src/Compression/CompressedReadBufferBase.cpp:175-222, CompressionInfo.h:10-51, ICompressionCodec.cpp compress()):
  16 bytes  CityHash128 (cityhash 1.0.2) of everything that follows, as {low64, high64}: written by write_frames, verified by read_frames
  1 byte    method (0x82 LZ4, 0x02 NONE, 0x92 Delta, ...)
  4 bytes   compressed size, little endian, INCLUDING this 9-byte header
  4 bytes   decompressed size
  payload
Codecs: ch_compress.c (LZ4 block format, Delta).  `write_frames` builds frames from raw bytes with Arrow's liblz4 as the
compressor — an implementation independent of both decoders under test.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
METHOD_NONE, METHOD_LZ4, METHOD_DELTA, METHOD_MULTIPLE = 0x02, 0x82, 0x92, 0x91
METHOD_T64, METHOD_DOUBLE_DELTA, METHOD_GORILLA = 0x93, 0x94, 0x95  # CompressionInfo.h:40-51
METHOD_ZSTD = 0x90
DELTA_LZ4 = "delta+lz4"  # CODEC(Delta(w), LZ4)
HEADER = 9
CHECKSUM = 16
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libchcompress.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "ch_compress.c")):
            subprocess.check_call(["make", "-C", _HERE, "libchcompress.so"])
        L = C.CDLL(so)
        for name in ("cho_lz4_decompress", "cho_delta_decode", "cho_double_delta_decode", "cho_t64_decode", "cho_gorilla_decode"):
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.cho_double_delta_encode.restype = C.c_long
        L.cho_double_delta_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_size_t]
        L.cho_gorilla_encode.restype = C.c_long
        L.cho_gorilla_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_size_t]
        L.cho_t64_encode.restype = C.c_long
        L.cho_t64_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_int, C.c_uint, C.c_int, C.c_void_p, C.c_size_t]
        L.cho_city_hash128.restype = None
        L.cho_city_hash128.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def lz4_decompress(payload: bytes, dst_size: int) -> bytes:
    src = np.frombuffer(payload, dtype=np.uint8)
    dst = np.zeros(dst_size, dtype=np.uint8)
    if lib().cho_lz4_decompress(src.ctypes.data, src.shape[0], dst.ctypes.data, dst_size) != 0:
        raise ValueError("CANNOT_DECOMPRESS")
    return dst.tobytes()


def delta_decode(payload: bytes, dst_size: int) -> bytes:
    src = np.frombuffer(payload, dtype=np.uint8)
    dst = np.zeros(dst_size, dtype=np.uint8)
    if lib().cho_delta_decode(src.ctypes.data, src.shape[0], dst.ctypes.data, dst_size) != 0:
        raise ValueError("CANNOT_DECOMPRESS")
    return dst.tobytes()


def double_delta_decode(payload: bytes, dst_size: int) -> bytes:
    """CompressionCodecDoubleDelta::doDecompressData over the codec payload ([width][bytes_to_skip] ...)"""
    src = np.frombuffer(payload, dtype=np.uint8)
    dst = np.zeros(dst_size, dtype=np.uint8)
    if lib().cho_double_delta_decode(src.ctypes.data, src.shape[0], dst.ctypes.data, dst_size) != 0:
        raise ValueError("CANNOT_DECOMPRESS")
    return dst.tobytes()


def double_delta_encode(raw: bytes, width: int) -> bytes:
    """CompressionCodecDoubleDelta::doCompressData -> the codec payload"""
    src = np.frombuffer(raw, dtype=np.uint8)
    cap = 2 + width + 4 + 2 * width + (len(raw) // width) * 9 + 64
    dst = np.zeros(cap, dtype=np.uint8)
    n = lib().cho_double_delta_encode(src.ctypes.data, src.shape[0], width, dst.ctypes.data, cap)
    if n < 0:
        raise ValueError("CANNOT_COMPRESS")
    return dst[:n].tobytes()


def gorilla_decode(payload: bytes, dst_size: int) -> bytes:
    src = np.frombuffer(payload, dtype=np.uint8)
    dst = np.zeros(dst_size, dtype=np.uint8)
    if lib().cho_gorilla_decode(src.ctypes.data, src.shape[0], dst.ctypes.data, dst_size) != 0:
        raise ValueError("CANNOT_DECOMPRESS")
    return dst.tobytes()


def gorilla_encode(raw: bytes, width: int) -> bytes:
    """CompressionCodecGorilla::doCompressData -> the codec payload"""
    src = np.frombuffer(raw, dtype=np.uint8)
    cap = 2 + width + 4 + width + (len(raw) // width) * (2 + 13 + 8 * width) // 8 + 64
    dst = np.zeros(cap, dtype=np.uint8)
    n = lib().cho_gorilla_encode(src.ctypes.data, src.shape[0], width, dst.ctypes.data, cap)
    if n < 0:
        raise ValueError("CANNOT_COMPRESS")
    return dst[:n].tobytes()


T64_MAGIC = {"uint8": 1, "uint16": 2, "uint32": 3, "uint64": 4, "int8": 6, "int16": 7, "int32": 8, "int64": 9}  # CompressionCodecT64.cpp:75-94


def t64_encode(values: np.ndarray, variant_bit: bool = False) -> bytes:
    """CompressionCodecT64::doCompressData of a numeric array -> the codec payload ([cookie][min][max][transposed blocks])"""
    values = np.ascontiguousarray(values)
    raw = values.view(np.uint8).reshape(-1)
    cap = 17 + (values.shape[0] + 64) * 8 + 1024
    dst = np.zeros(cap, dtype=np.uint8)
    n = lib().cho_t64_encode(raw.ctypes.data, raw.shape[0], values.dtype.itemsize, int(values.dtype.kind == "i"), T64_MAGIC[values.dtype.name],
                             int(variant_bit), dst.ctypes.data, cap)
    if n < 0:
        raise ValueError("CANNOT_COMPRESS")
    return dst[:n].tobytes()


def t64_decode(payload: bytes, dst_size: int) -> bytes:
    src = np.frombuffer(payload, dtype=np.uint8)
    dst = np.zeros(dst_size, dtype=np.uint8)
    if lib().cho_t64_decode(src.ctypes.data, src.shape[0], dst.ctypes.data, dst_size) != 0:
        raise ValueError("CANNOT_DECOMPRESS")
    return dst.tobytes()


def city_hash128(data: bytes):
    """CityHash_v1_0_2::CityHash128 -> (low64, high64)"""
    out = (C.c_uint64 * 2)()
    lib().cho_city_hash128(bytes(data), len(data), out)
    return int(out[0]), int(out[1])


def _framed(stage: bytes) -> bytes:
    """a frame = the checksum of (header + payload) in front of them (CompressedWriteBuffer::nextImpl)"""
    lo, hi = city_hash128(stage)
    return struct.pack("<QQ", lo, hi) + stage


def _stage(method: int, payload: bytes, decompressed_size: int) -> bytes:
    """one codec application as ICompressionCodec::compress lays it out: 9-byte header + payload"""
    return struct.pack("<BII", method, HEADER + len(payload), decompressed_size) + payload


def delta_encode(raw: bytes, width: int) -> bytes:
    """CompressionCodecDelta::doCompressData (CompressionCodecDelta.cpp:109-133): [width][bytes_to_skip][skipped][deltas]"""
    skip = len(raw) % width
    x = np.frombuffer(raw[skip:], dtype={1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[width])
    d = np.diff(np.concatenate((np.zeros(1, dtype=x.dtype), x))) if x.size else x
    return bytes([width, skip]) + raw[:skip] + d.astype(x.dtype).tobytes()


def write_codec_frames(values: np.ndarray, method: int, block_rows: int = 8192, t64_bit: bool = False) -> bytes:
    """a column file whose frames are single applications of DoubleDelta / T64 (CODEC(DoubleDelta), CODEC(T64)): one frame per block_rows values"""
    values = np.ascontiguousarray(values)
    out = bytearray()
    for lo in range(0, values.shape[0], block_rows):
        chunk = values[lo:lo + block_rows]
        raw = chunk.tobytes()
        payload = (double_delta_encode(raw, values.dtype.itemsize) if method == METHOD_DOUBLE_DELTA else gorilla_encode(raw, values.dtype.itemsize)
                   if method == METHOD_GORILLA else t64_encode(chunk, t64_bit))
        out += _framed(_stage(method, payload, len(raw)))
    return bytes(out)


def zstd_compress(raw: bytes, level: int = 1) -> bytes:
    """CompressionCodecZSTD::doCompressData (CompressionCodecZSTD.cpp:40-57): the payload is one zstd frame (libzstd through pyarrow here)"""
    import pyarrow as pa
    return pa.Codec("zstd", compression_level=level).compress(raw, asbytes=True)


def zstd_decompress(payload: bytes, dst_size: int) -> bytes:
    """CompressionCodecZSTD::doDecompressData (:60-66): ZSTD_decompress into a buffer of the size the frame header promises"""
    import pyarrow as pa
    out = pa.decompress(payload, decompressed_size=dst_size, codec="zstd", asbytes=True)
    if len(out) != dst_size:
        raise ValueError("CANNOT_DECOMPRESS: wrong decompressed size")
    return out


def _codec_encode(chunk: np.ndarray, method: int, t64_bit: bool = False) -> bytes:
    raw = chunk.tobytes()
    w = chunk.dtype.itemsize
    if method == METHOD_DOUBLE_DELTA:
        return double_delta_encode(raw, w)
    if method == METHOD_GORILLA:
        return gorilla_encode(raw, w)
    if method == METHOD_T64:
        return t64_encode(chunk, t64_bit)
    if method == METHOD_DELTA:
        return delta_encode(raw, w)
    raise NotImplementedError(hex(method))


def _general_encode(raw: bytes, method: int) -> bytes:
    import pyarrow as pa
    if method == METHOD_LZ4:
        return pa.compress(raw, codec="lz4_raw", asbytes=True)
    if method == METHOD_ZSTD:
        return zstd_compress(raw)
    if method == METHOD_NONE:
        return raw
    raise NotImplementedError(hex(method))


def write_multiple_frames(values: np.ndarray, codec: int, general: int, block_rows: int = 8192, t64_bit: bool = False) -> bytes:
    """CODEC(<column codec>, <general-purpose codec>) -- e.g. CODEC(DoubleDelta, ZSTD), CODEC(T64, LZ4): a Multiple frame per block_rows
    values (CompressionCodecMultiple.cpp:40-66: the method list, then the stages applied in order, each with its own 9-byte header)"""
    values = np.ascontiguousarray(values)
    out = bytearray()
    for lo in range(0, values.shape[0], block_rows):
        chunk = values[lo:lo + block_rows]
        st1 = _stage(codec, _codec_encode(chunk, codec, t64_bit), chunk.nbytes)
        st2 = _stage(general, _general_encode(st1, general), len(st1))
        out += _framed(_stage(METHOD_MULTIPLE, bytes([2, codec, general]) + st2, chunk.nbytes))
    return bytes(out)


def write_frames(raw: bytes, block_size: int = 65536, method=METHOD_LZ4, delta_width: int = 8) -> bytes:
    """CompressedWriteBuffer: one frame per `block_size` bytes of input, each with its CityHash128 checksum.
    method DELTA_LZ4 = CODEC(Delta(delta_width), LZ4): a Multiple frame (CompressionCodecMultiple.cpp:40-66) -- the method list, then
    the stages applied in order, each with its own header."""
    import pyarrow as pa
    out = bytearray()
    for lo in range(0, len(raw), block_size):
        chunk = raw[lo:lo + block_size]
        if method == DELTA_LZ4:
            st1 = _stage(METHOD_DELTA, delta_encode(chunk, delta_width), len(chunk))
            st2 = _stage(METHOD_LZ4, pa.compress(st1, codec="lz4_raw", asbytes=True), len(st1))
            out += _framed(_stage(METHOD_MULTIPLE, bytes([2, METHOD_DELTA, METHOD_LZ4]) + st2, len(chunk)))
            continue
        payload = _general_encode(chunk, method)
        out += _framed(_stage(method, payload, len(chunk)))
    return bytes(out)


def parse_frames(buf: bytes, verify_checksums: bool = True):
    """-> list of (method, payload_offset, payload_size, decompressed_size)"""
    frames, pos = [], 0
    while pos < len(buf):
        if len(buf) - pos < CHECKSUM + HEADER:
            raise ValueError("CANNOT_READ_ALL_DATA")
        method, csize, dsize = struct.unpack_from("<BII", buf, pos + CHECKSUM)
        if csize < HEADER or pos + CHECKSUM + csize > len(buf):
            raise ValueError("CANNOT_DECOMPRESS: bad frame size")
        if verify_checksums and struct.unpack_from("<QQ", buf, pos) != city_hash128(buf[pos + CHECKSUM:pos + CHECKSUM + csize]):
            raise ValueError("CHECKSUM_DOESNT_MATCH")
        frames.append((method, pos + CHECKSUM + HEADER, csize - HEADER, dsize))
        pos += CHECKSUM + csize
    return frames


def read_frames(buf: bytes) -> bytes:
    """CompressedReadBuffer over the whole buffer"""
    out = bytearray()
    for method, off, size, dsize in parse_frames(buf):
        payload = buf[off:off + size]
        if method == METHOD_LZ4:
            out += lz4_decompress(payload, dsize)
        elif method == METHOD_NONE:
            out += payload
        elif method == METHOD_DELTA:
            out += delta_decode(payload, dsize)
        elif method == METHOD_DOUBLE_DELTA:
            out += double_delta_decode(payload, dsize)
        elif method == METHOD_T64:
            out += t64_decode(payload, dsize)
        elif method == METHOD_GORILLA:
            out += gorilla_decode(payload, dsize)
        elif method == METHOD_MULTIPLE:
            out += _multiple_decode(payload, dsize)
        elif method == METHOD_ZSTD:
            out += zstd_decompress(payload, dsize)
        else:
            raise NotImplementedError(hex(method))
    return bytes(out)


def _decode_stage(buf: bytes) -> bytes:
    method, csize, dsize = struct.unpack_from("<BII", buf, 0)
    payload = buf[HEADER:csize]
    if method == METHOD_LZ4:
        return lz4_decompress(payload, dsize)
    if method == METHOD_DELTA:
        return delta_decode(payload, dsize)
    if method == METHOD_NONE:
        return payload
    if method == METHOD_ZSTD:
        return zstd_decompress(payload, dsize)
    if method == METHOD_DOUBLE_DELTA:
        return double_delta_decode(payload, dsize)
    if method == METHOD_GORILLA:
        return gorilla_decode(payload, dsize)
    if method == METHOD_T64:
        return t64_decode(payload, dsize)
    raise NotImplementedError(hex(method))


def _multiple_decode(payload: bytes, dsize: int) -> bytes:
    """CompressionCodecMultiple::doDecompressData (CompressionCodecMultiple.cpp:68-130): undo the stages from the last to the first"""
    n = payload[0]
    if n == 0:
        raise ValueError("Wrong compression methods list")
    buf = payload[1 + n:]
    for _ in range(n):
        buf = _decode_stage(buf)
    if len(buf) != dsize:
        raise ValueError("Wrong final decompressed size in codec Multiple")
    return buf
