"""compressed frames (LZ4 / NONE) decoded in HBM.  Compressor = Apache Arrow's liblz4 (independent of both
decoders); CPU: the C restatement returns the original bytes; GPU: the wave-per-frame decoder does, for real column data and for
adversarial streams (long overlapping matches, 255-run length bytes, incompressible data), and rejects malformed frames."""
import struct

import numpy as np
import pytest

from oracle import compression as OC


def _datasets(rng):
    yield "monotone int64", np.cumsum(rng.integers(0, 5, size=400_000)).astype(np.int64).tobytes()
    yield "low-cardinality u8", rng.integers(0, 11, size=700_001).astype(np.uint8).tobytes()
    yield "zeros (offset-1 runs, 255-length bytes)", bytes(300_000)
    yield "period 3", (b"abc" * 100_000)[:299_999]
    yield "period 70 (offset > 64 < length)", (bytes(range(70)) * 5000)
    yield "random (incompressible)", rng.integers(0, 256, size=200_003, dtype=np.uint8).tobytes()
    yield "dates u16", (8000 + np.sort(rng.integers(0, 2500, size=500_000))).astype(np.uint16).tobytes()
    yield "tiny", b"x"
    yield "float64 prices", np.round(rng.random(100_000) * 1000, 2).tobytes()


def test_oracle_lz4_against_arrow_compressor():
    rng = np.random.Generator(np.random.PCG64(8))
    for name, raw in _datasets(rng):
        for bs in (65536, 1 << 20):
            assert OC.read_frames(OC.write_frames(raw, bs)) == raw, name
    assert OC.read_frames(OC.write_frames(b"hello", method=OC.METHOD_NONE)) == b"hello"
    # hand-made block: 4 literals "abcd", match offset 4 length 4+15+255+1 (two length bytes), final literal "Z"
    blk = bytes([0x4F]) + b"abcd" + struct.pack("<H", 4) + bytes([255, 1]) + bytes([0x10]) + b"Z"
    assert OC.lz4_decompress(blk, 4 + 275 + 1) == (b"abcd" * 70)[:279] + b"Z"
    with pytest.raises(ValueError):
        OC.lz4_decompress(blk[:-1], 280)          # truncated
    with pytest.raises(ValueError):
        OC.lz4_decompress(bytes([0x10, 65, 9, 0]) + b"\x00", 100)  # offset beyond the output start


def test_oracle_delta_codec():
    x = np.array([10, 12, 11, 2**63, 5], dtype=np.uint64)
    d = np.diff(np.concatenate(([np.uint64(0)], x))).astype(np.uint64)
    assert OC.delta_decode(bytes([8, 0]) + d.tobytes(), 40) == x.tobytes()
    assert OC.delta_encode(x.tobytes(), 8) == bytes([8, 0]) + d.tobytes()
    rng = np.random.Generator(np.random.PCG64(2))
    for w, raw in ((8, np.cumsum(rng.integers(0, 1000, size=70_000)).astype(np.int64).tobytes()), (4, rng.integers(0, 2**32, size=33_333, dtype=np.uint32).tobytes()),
                   (2, b"xyz" + np.arange(50_001, dtype=np.uint16).tobytes()), (1, bytes(range(256)) * 300)):
        buf = OC.write_frames(raw, 65536, OC.DELTA_LZ4, w)     # CODEC(Delta(w), LZ4): Multiple frames
        assert OC.read_frames(buf) == raw


@pytest.mark.gpu
def test_gpu_frames_decode_to_original_bytes():
    import clickhouse_amd as ch
    from clickhouse_amd import compression as CC
    ctx = ch.Context()
    rng = np.random.Generator(np.random.PCG64(8))
    for name, raw in _datasets(rng):
        for bs, method in ((65536, OC.METHOD_LZ4), (1 << 20, OC.METHOD_LZ4), (4096, OC.METHOD_LZ4), (65536, OC.METHOD_NONE)):
            buf = OC.write_frames(raw, bs, method)
            frames = CC.parse_frames(buf)
            assert [f[:4] for f in frames] == OC.parse_frames(buf)
            out = CC.decompress_frames(ctx, ctx.upload(np.frombuffer(buf, dtype=np.uint8)), frames).numpy().tobytes()
            assert out == raw, (name, bs, method)


@pytest.mark.gpu
def test_gpu_delta_lz4_frames_decode_to_original_bytes():
    """CODEC(Delta(w), LZ4): LZ4 stage into a stage buffer, Delta stage (running sums) into the column"""
    import clickhouse_amd as ch
    from clickhouse_amd import compression as CC
    ctx = ch.Context()
    rng = np.random.Generator(np.random.PCG64(2))
    cases = [(8, np.cumsum(rng.integers(0, 1000, size=300_000)).astype(np.int64).tobytes()),
             (8, rng.integers(-2**63, 2**63 - 1, size=20_001, dtype=np.int64).tobytes()),          # wrap-around sums
             (4, (1_700_000_000 + np.cumsum(rng.integers(0, 3, size=400_003))).astype(np.uint32).tobytes()),  # timestamps
             (2, b"xyz" + np.arange(50_001, dtype=np.uint16).tobytes()),                          # bytes_to_skip = 1
             (1, bytes(range(256)) * 300), (8, np.arange(3, dtype=np.int64).tobytes())]
    for w, raw in cases:
        for bs in (65536, 1 << 20, 4096):
            buf = OC.write_frames(raw, bs, OC.DELTA_LZ4, w)
            frames = CC.parse_frames(buf)
            assert all(f[0] == 0x82 and f[4] == 0x92 for f in frames)
            out = CC.decompress_frames(ctx, ctx.upload(np.frombuffer(buf, dtype=np.uint8)), frames).numpy().tobytes()
            assert out == raw, (w, bs)
    # a corrupted Delta header inside the LZ4 stage is an error, not a fault
    raw = np.arange(10_000, dtype=np.int64).tobytes()
    st1 = bytearray(OC._stage(OC.METHOD_DELTA, OC.delta_encode(raw, 8), len(raw)))
    st1[9] = 3  # element width 3
    import pyarrow as pa
    st2 = OC._stage(OC.METHOD_LZ4, pa.compress(bytes(st1), codec="lz4_raw", asbytes=True), len(st1))
    buf = bytes(16) + OC._stage(OC.METHOD_MULTIPLE, bytes([2, 0x92, 0x82]) + st2, len(raw))
    with pytest.raises(ch.ChgpuError) as ei:
        CC.decompress_frames(ctx, ctx.upload(np.frombuffer(buf, dtype=np.uint8)), CC.parse_frames(buf))
    assert ei.value.code == ch._capi.ERR_BAD_ARGUMENTS


@pytest.mark.gpu
def test_gpu_read_column_file_then_filter_sum():
    """compressed column file -> HBM column -> the hot path, no CPU decompression"""
    import clickhouse_amd as ch
    from clickhouse_amd import compression as CC
    ctx = ch.Context()
    rng = np.random.Generator(np.random.PCG64(1))
    a = rng.integers(0, 2**31, size=1_000_003, dtype=np.int64)
    a[::7] = 5  # some compressibility
    col = CC.read_column_file(ctx, OC.write_frames(a.tobytes()), np.int64)
    assert col.size() == a.shape[0] and np.array_equal(col.numpy(), a)
    s, c = ch.filter_sum(col, ch.LT, 214748365)
    m = a < 214748365
    assert (int(s), c) == (int(a[m].sum()), int(m.sum()))
    d = (8000 + rng.integers(0, 2500, size=77_777)).astype(np.uint16)
    assert np.array_equal(CC.read_column_file(ctx, OC.write_frames(d.tobytes(), 8192), np.uint16).numpy(), d)


@pytest.mark.gpu
def test_gpu_malformed_frames_are_errors_not_faults():
    import clickhouse_amd as ch
    from clickhouse_amd import compression as CC
    ctx = ch.Context()
    good = OC.write_frames(np.arange(50_000, dtype=np.int64).tobytes())
    frames = CC.parse_frames(good)
    up = ctx.upload(np.frombuffer(good, dtype=np.uint8))
    m, off, size, dsize = frames[0][:4]
    for bad in ([(m, off, size - 3, dsize, 0, dsize)] + frames[1:],         # truncated payload
                [(m, off, size, dsize + 100, 0, dsize + 100)] + frames[1:],  # claims more output than the block yields
                [(m, off + 1, size - 1, dsize, 0, dsize)] + frames[1:]):    # starts inside the block: garbage tokens / offsets
        with pytest.raises(ch.ChgpuError) as ei:
            CC.decompress_frames(ctx, up, bad)
        assert ei.value.code == ch._capi.ERR_BAD_ARGUMENTS
    with pytest.raises(ch.ChgpuError) as ei:
        CC.decompress_frames(ctx, up, [(0x90, off, size, dsize, 0, dsize)])  # ZSTD: CPU path
    assert ei.value.code == ch._capi.ERR_NOT_IMPLEMENTED
    with pytest.raises(ch.ChgpuError):
        CC.parse_frames(good[:-5])
    # the context still works after the rejected calls
    assert CC.decompress_frames(ctx, up, frames).numpy().tobytes() == np.arange(50_000, dtype=np.int64).tobytes()
