"""TEST INFRASTRUCTURE ONLY — the reference's compressed-frame layout and codecs restated for the feed path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Frame (src/Compression/CompressedReadBufferBase.cpp:175-222, CompressionInfo.h:10-51, ICompressionCodec.cpp compress()):
  16 bytes  CityHash128 of everything that follows (NOT verified here nor by the product: cityhash102 is not restated)
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
METHOD_NONE, METHOD_LZ4, METHOD_DELTA = 0x02, 0x82, 0x92
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
        for name in ("cho_lz4_decompress", "cho_delta_decode"):
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
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


def write_frames(raw: bytes, block_size: int = 65536, method: int = METHOD_LZ4) -> bytes:
    """CompressedWriteBuffer: one frame per `block_size` bytes of input (checksum bytes are left zero: unverified)"""
    import pyarrow as pa
    out = bytearray()
    for lo in range(0, len(raw), block_size):
        chunk = raw[lo:lo + block_size]
        payload = pa.compress(chunk, codec="lz4_raw", asbytes=True) if method == METHOD_LZ4 else chunk
        out += bytes(CHECKSUM) + struct.pack("<BII", method, HEADER + len(payload), len(chunk)) + payload
    return bytes(out)


def parse_frames(buf: bytes):
    """-> list of (method, payload_offset, payload_size, decompressed_size)"""
    frames, pos = [], 0
    while pos < len(buf):
        if len(buf) - pos < CHECKSUM + HEADER:
            raise ValueError("CANNOT_READ_ALL_DATA")
        method, csize, dsize = struct.unpack_from("<BII", buf, pos + CHECKSUM)
        if csize < HEADER or pos + CHECKSUM + csize > len(buf):
            raise ValueError("CANNOT_DECOMPRESS: bad frame size")
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
        else:
            raise NotImplementedError(hex(method))
    return bytes(out)
