"""compressed frames (LZ4 / NONE) decoded in HBM.  Compressor = Apache Arrow's liblz4 (independent of both
decoders); CPU: the C restatement returns the original bytes; GPU: the wave-per-frame decoder does, for real column data and for
adversarial streams (long overlapping matches, 255-run length bytes, incompressible data), and rejects malformed frames."""
import json
import os
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
    buf = OC._framed(OC._stage(OC.METHOD_MULTIPLE, bytes([2, 0x92, 0x82]) + st2, len(raw)))  # the checksum is right, the Delta header is not
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


# ---- frame integrity: host-side, no GPU needed (the walk runs on the host in the product too) -------------------------------------------
def test_city_hash128_product_and_oracle_match_reference_vectors():
    """CityHash128 1.0.2: the product's implementation (feed.hip) and the oracle's (ch_compress.c) against vectors produced by the reference's
    own contrib/cityhash102 compiled in place (tests/golden/round2_kat.json, generator tests/golden/make_golden.py)"""
    import json
    import os
    from clickhouse_amd import compression as CC
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "round2_kat.json")))["city_hash128"]
    assert len(kat) >= 40
    for v in kat:
        data = bytes.fromhex(v["hex"])
        want = (int(v["low64"]), int(v["high64"]))
        assert CC.city_hash128(data) == want and OC.city_hash128(data) == want, len(data)


def test_frame_walk_verifies_checksums_and_caps_sizes():
    import clickhouse_amd as ch
    from clickhouse_amd import compression as CC
    raw = np.arange(30_000, dtype=np.int64).tobytes()
    good = OC.write_frames(raw, 65536)
    frames = CC.parse_frames(good)
    assert len(frames) == 4 and [f[:4] for f in frames] == OC.parse_frames(good)
    for pos in (0, 15, 16, 20, 24, 25, 1000, len(good) - 1):                 # checksum bytes, header bytes, payload bytes
        bad = bytearray(good)
        bad[pos] ^= 0x04
        with pytest.raises(ch.ChgpuError) as ei:
            CC.parse_frames(bytes(bad))
        assert ei.value.code == ch._capi.ERR_BAD_ARGUMENTS
    bad = bytearray(good)
    bad[1000] ^= 0x04
    assert len(CC.parse_frames(bytes(bad), verify_checksums=False)) == 4      # disable_checksum: the walk alone
    import struct
    huge = bytearray(good[:16 + 9])
    struct.pack_into("<I", huge, 16 + 5, 0x40000001)                          # decompressed size above DBMS_MAX_COMPRESSED_SIZE
    with pytest.raises(ch.ChgpuError) as ei:
        CC.parse_frames(bytes(huge), verify_checksums=False)
    assert "TOO_LARGE_SIZE_COMPRESSED" in str(ei.value)


def test_native_block_header_walk():
    """NativeReader::read's header walk over a block written the way NativeWriter lays it out"""
    import ctypes as C
    import clickhouse_amd as ch
    from clickhouse_amd import compression as CC
    K = ch._capi

    def varuint(x):
        out = bytearray()
        while x >= 0x80:
            out.append((x & 0x7F) | 0x80)
            x >>= 7
        out.append(x)
        return bytes(out)

    def string(s):
        return varuint(len(s)) + s.encode()

    a = np.arange(1000, dtype=np.uint64) * 3
    b = (np.arange(1000) % 7).astype(np.int32)
    for rev in (0, 54453, 54454):
        blk = bytearray()
        if rev > 0:
            blk += varuint(1) + bytes([0]) + varuint(2) + (17).to_bytes(4, "little", signed=True) + varuint(0)   # BlockInfo: is_overflows, bucket_num
        blk += varuint(2) + varuint(1000)
        for name, tname, arr in (("k", "UInt64", a), ("v", "Int32", b)):
            blk += string(name) + string(tname)
            if rev >= 54454:
                blk += bytes([0])
            blk += arr.tobytes()
        raw = bytes(blk) + b"next block"
        arr_t = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        cols = (CC._NativeColumnStruct * 2)()
        ncols, nrows, used = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        bucket, over = C.c_int32(-1), C.c_int(0)
        K.check(K.lib().chgpu_native_walk_block(arr_t, len(raw), rev, 2, cols, C.byref(ncols), C.byref(nrows), C.byref(bucket), C.byref(over), C.byref(used)))
        assert (ncols.value, nrows.value, used.value) == (2, 1000, len(blk)) and bucket.value == (17 if rev > 0 else -1)
        assert (cols[0].name, cols[0].type_name, cols[0].type, cols[0].data_bytes) == (b"k", b"UInt64", K.U64, 8000)
        assert raw[cols[1].data_offset:cols[1].data_offset + 4000] == b.tobytes()
    rc = K.lib().chgpu_native_walk_block(arr_t, 30, 54454, 2, cols, C.byref(ncols), C.byref(nrows), C.byref(bucket), C.byref(over), C.byref(used))
    assert rc == K.ERR_BAD_ARGUMENTS                                                                   # truncated
    s = bytes(varuint(1) + varuint(3) + string("s") + string("Array(UInt8)") + b"\x01\x02\x03abc")
    arr_s = (C.c_uint8 * len(s)).from_buffer_copy(s)
    assert K.lib().chgpu_native_walk_block(arr_s, len(s), 0, 2, cols, C.byref(ncols), C.byref(nrows), None, None, C.byref(used)) == K.ERR_NOT_IMPLEMENTED


def _synthetic_lz4_block(rng, target, small):
    """a valid LZ4 block built sequence by sequence together with the bytes it decodes to: literal and match lengths on both sides of the
    token's 4-bit fields, offsets 1.. (overlapping matches of every period), near and far match sources -- shapes a real compressor
    rarely emits but the format allows"""
    out, blk = bytearray(), bytearray()
    while len(out) < target:
        lit = int(rng.choice([0, 1, 3, 4, 8, 13, 14, 15, 16, 40, 300])) if not small else int(rng.integers(0, 14))
        if not out and lit == 0:
            lit = 1
        ml = int(rng.choice([4, 5, 8, 15, 16, 17, 18, 19, 20, 60, 274, 700, 3000])) if not small else int(rng.integers(4, 17))
        lits = rng.integers(0, 256, size=lit, dtype=np.uint8).tobytes()
        have = len(out) + lit
        far = have if rng.random() < 0.15 else min(have, int(rng.choice([1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 31, 64, 100, 2047, 2048, 2049, 5000])))
        offset = int(rng.integers(1, min(far, 65535) + 1))
        token = (min(lit, 15) << 4) | min(ml - 4, 15)
        blk.append(token)
        if lit >= 15:
            r = lit - 15
            while r >= 255:
                blk.append(255)
                r -= 255
            blk.append(r)
        blk += lits
        out += lits
        blk += struct.pack("<H", offset)
        if ml - 4 >= 15:
            r = ml - 19
            while r >= 255:
                blk.append(255)
                r -= 255
            blk.append(r)
        start = len(out) - offset
        for k in range(ml):
            out.append(out[start + k])
    tail = rng.integers(0, 256, size=int(rng.integers(12, 40)), dtype=np.uint8).tobytes()   # the last sequence: literals only
    blk.append(min(len(tail), 15) << 4)
    if len(tail) >= 15:
        blk.append(len(tail) - 15)
    blk += tail
    out += tail
    return bytes(blk), bytes(out)


def _synthetic_frames(rng, n_frames):
    bufs, raws = [], []
    for i in range(n_frames):
        blk, raw = _synthetic_lz4_block(rng, int(rng.choice([50, 700, 5000, 30_000])), small=bool(i % 2))
        bufs.append(OC._framed(OC._stage(OC.METHOD_LZ4, blk, len(raw))))
        raws.append(raw)
    return b"".join(bufs), b"".join(raws)


def test_oracle_decodes_synthetic_sequences():
    rng = np.random.Generator(np.random.PCG64(21))
    buf, raw = _synthetic_frames(rng, 23)
    assert OC.read_frames(buf) == raw


@pytest.mark.gpu
def test_gpu_decodes_synthetic_sequences():
    """frame counts around the grid's stride, short and long sequences side by side, every small overlapping period on the fast path,
    matches on both sides of what the LDS ring serves (offsets 2047 / 2048 / 2049 / far)"""
    import clickhouse_amd as ch
    from clickhouse_amd import compression as CC
    ctx = ch.Context()
    rng = np.random.Generator(np.random.PCG64(22))
    for n_frames in (1, 2, 3, 5, 64, 257):
        buf, raw = _synthetic_frames(rng, n_frames)
        frames = CC.parse_frames(buf)
        assert len(frames) == n_frames
        out = CC.decompress_frames(ctx, ctx.upload(np.frombuffer(buf, dtype=np.uint8)), frames).numpy().tobytes()
        assert out == raw, n_frames


# ---- DoubleDelta / T64 (round 3) ---------------------------------------------------------------------------------------------------
NP_OF_NAME = {"Int8": np.int8, "UInt8": np.uint8, "Int16": np.int16, "UInt16": np.uint16, "Int32": np.int32, "UInt32": np.uint32, "Int64": np.int64, "UInt64": np.uint64}


def dd_compat_sequence(dtype):
    """DDCompatibilityTestSequence<T> (src/Compression/tests/gtest_compressionCodec.cpp:1139-1166) restated: three times 42, then for every
    corner point p of the encoding whose magnitude fits T the six values a generator of double deltas p-4 .. p+1 yields, starting afresh
    (previous value and delta 0) at every corner point -- the lambda is passed by value.  Arithmetic in Int64, cast to T by truncation."""
    info = np.iinfo(dtype)
    vals = [42, 42, 42]
    for p in (-63, 64, -255, 256, -2047, 2048, -2**31, 2**31 - 1):
        if abs(p) > info.max:
            break
        prev = prev_delta = 0
        for dd in range(p - 4, p + 2):
            cur = dd + prev + prev_delta
            prev, prev_delta = cur, dd + prev_delta
            vals.append(cur)
    return np.array([v & 0xFFFFFFFFFFFFFFFF for v in vals], dtype=np.uint64).astype(dtype)   # static_cast<T>: modular


def _codec_kat():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "codec_kat.json")))


def test_oracle_double_delta_pinned_by_reference_compatibility_vectors():
    """both directions against the reference's own (sequence, bytes) pairs: decode(bytes) == sequence and encode(sequence) == bytes"""
    kat = _codec_kat()
    for v in kat["double_delta_frames"]:
        dt = NP_OF_NAME[v["type"]]
        frame = bytes.fromhex(v["frame_hex"])
        method, csize, dsize = struct.unpack_from("<BII", frame, 0)
        assert method == OC.METHOD_DOUBLE_DELTA and csize == len(frame)
        seq = dd_compat_sequence(dt)
        assert dsize == seq.nbytes, (v["type"], dsize, seq.nbytes)
        assert OC.double_delta_decode(frame[9:], dsize) == seq.tobytes(), v["type"]
        assert OC.double_delta_encode(seq.tobytes(), seq.dtype.itemsize) == frame[9:], v["type"]
    for ex in kat["double_delta_doc_examples"]:
        seq = np.array(ex["values"], dtype=NP_OF_NAME[ex["type"]])
        w = seq.dtype.itemsize
        payload = bytes([w, 0]) + bytes.fromhex(ex["payload_hex"])
        assert OC.double_delta_encode(seq.tobytes(), w) == payload
        assert OC.double_delta_decode(payload, seq.nbytes) == seq.tobytes()


def t64_scenarios():
    """the value ranges the reference's T64 tests insert and read back (tests/queries/0_stateless/00870_t64_codec.sql, 00871_t64_codec_signed.sql,
    00872_t64_bit_codec.sql): runs of 1, 2, 4 rows, then blocks that straddle every power of two 2^8 .. 2^56 with 10 / 11 / 64 / 65 ... rows,
    for every width; the signed test mirrors them around zero.  The tests assert `column == T64 column` for every row."""
    out = []
    for dt in (np.uint8, np.uint16, np.uint32, np.uint64, np.int8, np.int16, np.int32, np.int64):
        signed = np.dtype(dt).kind == "i"
        seqs = [np.arange(1), np.arange(2), np.full(4, 42), np.arange(2**8), np.arange(2**9)]
        for e, counts in ((16, (10, 11, 64, 65)), (24, (10, 11, 128, 129)), (32, (10, 20, 256, 257)), (40, (10, 20, 512, 513)), (48, (10, 20, 1024, 1025)),
                          (56, (10, 20, 2048, 2049))):
            for c in counts:
                seqs.append(2**e - 10 + np.arange(c, dtype=np.int64))
                seqs.append(2**e - 64 + np.arange(c, dtype=np.int64))
            seqs.append(2**e - 1 + np.arange(counts[-1], dtype=np.int64))
        for s_ in seqs:
            s_ = np.asarray(s_, dtype=np.int64)
            out.append(s_.astype(np.uint64).astype(dt))               # toUInt / toInt of the reference: truncation
            if signed:
                out.append((-s_).astype(np.uint64).astype(dt))
                out.append((s_ - s_.shape[0] // 2).astype(np.uint64).astype(dt))   # a block that crosses zero
    return out


def test_oracle_t64_round_trips_the_reference_test_ranges():
    for seq in t64_scenarios():
        for bit in (False, True):
            payload = OC.t64_encode(seq, bit)
            assert OC.t64_decode(payload, seq.nbytes) == seq.tobytes(), (seq.dtype, seq[:4], bit)
    # a payload whose size is not a whole number of transposed blocks, and an unknown type cookie, are refused
    p = OC.t64_encode(np.arange(100, dtype=np.uint32))
    with pytest.raises(ValueError):
        OC.t64_decode(p[:-3], 400)
    with pytest.raises(ValueError):
        OC.t64_decode(bytes([5]) + p[1:], 400)


@pytest.mark.gpu
def test_gpu_double_delta_frames_pinned_by_reference_vectors_and_equal_to_the_oracle():
    import clickhouse_amd as ch
    ctx = ch.Context(0)
    kat = _codec_kat()
    for v in kat["double_delta_frames"]:
        dt = NP_OF_NAME[v["type"]]
        frame = bytes.fromhex(v["frame_hex"])
        lo, hi = OC.city_hash128(frame)
        col = ch.compression.read_column_file(ctx, struct.pack("<QQ", lo, hi) + frame, dt)
        assert np.array_equal(col.numpy(), dd_compat_sequence(dt)), v["type"]
    rng = np.random.Generator(np.random.PCG64(5))
    for dt in (np.uint8, np.int16, np.uint32, np.int64, np.uint64):
        info = np.iinfo(dt)
        n = 100_003
        cases = [np.cumsum(rng.integers(0, 50, size=n)).astype(np.uint64).astype(dt),                       # near-constant stride: 1-9 bit codes
                 rng.integers(info.min, info.max, size=n, dtype=np.int64 if info.min < 0 else np.uint64).astype(dt) if dt != np.uint64
                 else rng.integers(0, 2**64, size=n, dtype=np.uint64),                                      # every double delta at full width
                 np.full(n, 7).astype(dt), np.arange(3).astype(dt), np.zeros(0, dtype=dt), np.array([info.max, info.min, info.max], dtype=dt)]
        for seq in cases:
            buf = OC.write_codec_frames(seq, OC.METHOD_DOUBLE_DELTA, block_rows=8192)
            assert OC.read_frames(buf) == seq.tobytes()
            if seq.shape[0] == 0:
                continue
            got = ch.compression.read_column_file(ctx, buf, dt).numpy()
            assert np.array_equal(got, seq), (dt, seq[:5])


@pytest.mark.gpu
def test_gpu_t64_frames_round_trip_the_reference_test_ranges():
    import clickhouse_amd as ch
    ctx = ch.Context(0)
    rng = np.random.Generator(np.random.PCG64(6))
    extra = [rng.integers(1000, 5000, size=100_001).astype(np.uint32), rng.integers(-300, 300, size=70_000).astype(np.int64),
             rng.integers(0, 2**64, size=9_999, dtype=np.uint64), np.full(12345, -5, dtype=np.int16)]
    for seq in t64_scenarios() + extra:
        for bit in (False, True):
            buf = OC.write_codec_frames(seq, OC.METHOD_T64, block_rows=4096 + 37, t64_bit=bit)
            assert OC.read_frames(buf) == seq.tobytes()
            got = ch.compression.read_column_file(ctx, buf, seq.dtype.type).numpy()
            assert np.array_equal(got, seq), (seq.dtype, seq[:4], bit)


@pytest.mark.gpu
def test_gpu_malformed_codec_frames_are_errors_not_faults():
    import clickhouse_amd as ch
    ctx = ch.Context(0)
    seq = np.arange(1000, dtype=np.uint32) * 3
    for method in (OC.METHOD_T64, OC.METHOD_DOUBLE_DELTA):
        good = OC.write_codec_frames(seq, method, block_rows=1000)
        frame = bytearray(good[16:])
        for pos, val in ((9, 0x7f), (9, 3), (10, 0xff)):       # the cookie / width byte, bytes_to_skip
            bad = bytearray(frame)
            bad[pos] = val
            lo, hi = OC.city_hash128(bytes(bad))
            try:
                got = ch.compression.read_column_file(ctx, struct.pack("<QQ", lo, hi) + bytes(bad), np.uint32).numpy()
                assert got.shape[0] == 1000   # a changed byte that still parses must not fault either
            except ch.ChgpuError as e:
                assert e.code in (ch._capi.ERR_BAD_ARGUMENTS, ch._capi.ERR_NOT_IMPLEMENTED)


def test_oracle_gorilla_pinned_by_the_reference_worked_example_and_round_trips():
    kat = _codec_kat()
    for ex in kat["gorilla_doc_examples"]:
        seq = np.array(ex["values"], dtype=np.float32)
        payload = bytes([4, 0]) + bytes.fromhex(ex["payload_hex"])
        assert OC.gorilla_encode(seq.tobytes(), 4) == payload
        assert OC.gorilla_decode(payload, seq.nbytes) == seq.tobytes()
    rng = np.random.Generator(np.random.PCG64(8))
    for dt in (np.float32, np.float64, np.uint8, np.uint16, np.uint32, np.uint64):
        for seq in _gorilla_cases(rng, dt):
            assert OC.gorilla_decode(OC.gorilla_encode(seq.tobytes(), seq.dtype.itemsize), seq.nbytes) == seq.tobytes()


def _gorilla_cases(rng, dt):
    n = 50_001
    if np.dtype(dt).kind == "f":
        return [np.cumsum(rng.random(n)).astype(dt), (rng.random(n) * 1e-3 + 20.0).astype(dt), np.full(n, 1.5).astype(dt),
                rng.standard_normal(n).astype(dt) * dt(1e30), np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0], dtype=dt)]
    info = np.iinfo(dt)
    return [rng.integers(0, 200, size=n).astype(dt), rng.integers(0, info.max, size=n, dtype=np.uint64, endpoint=True).astype(dt), np.array([info.max, 0, info.max, 1], dtype=dt)]


@pytest.mark.gpu
def test_gpu_gorilla_frames_equal_the_oracle_and_the_reference_example():
    import clickhouse_amd as ch
    ctx = ch.Context(0)
    kat = _codec_kat()
    ex = kat["gorilla_doc_examples"][0]
    seq = np.array(ex["values"], dtype=np.float32)
    frame = struct.pack("<BII", OC.METHOD_GORILLA, 9 + 2 + len(bytes.fromhex(ex["payload_hex"])), seq.nbytes) + bytes([4, 0]) + bytes.fromhex(ex["payload_hex"])
    lo, hi = OC.city_hash128(frame)
    got = ch.compression.read_column_file(ctx, struct.pack("<QQ", lo, hi) + frame, np.float32).numpy()
    assert got.tobytes() == seq.tobytes()
    rng = np.random.Generator(np.random.PCG64(9))
    for dt in (np.float32, np.float64, np.uint8, np.uint16, np.uint32, np.uint64):
        for seq in _gorilla_cases(rng, dt):
            buf = OC.write_codec_frames(seq, OC.METHOD_GORILLA, block_rows=4096)
            assert OC.read_frames(buf) == seq.tobytes()
            got = ch.compression.read_column_file(ctx, buf, dt).numpy()
            assert got.tobytes() == seq.tobytes(), (dt, seq[:4])


# ---- round 3: ZSTD and the usual pairs CODEC(<column codec>, <general-purpose codec>) ------------------------------------------------------
_PAIR_CODECS = ["DELTA", "T64", "DOUBLE_DELTA", "GORILLA"]
_PAIR_GENERALS = ["LZ4", "ZSTD", "NONE"]


def _pair_values(rng, dt, n):
    if np.dtype(dt).kind == "f":
        return (np.cumsum(rng.normal(size=n)) * 3.5).astype(dt)
    info = np.iinfo(dt)
    walk = np.cumsum(rng.integers(-3, 40, size=n))
    return (walk % (int(info.max) - int(info.min) + 1) + int(info.min)).astype(dt) if np.dtype(dt).itemsize < 8 else walk.astype(dt)


def test_oracle_multiple_frames_round_trip():
    """Multiple{codec, general} as CompressionCodecMultiple lays it out (every stage with its own header), and plain ZSTD frames; the
    codecs of gtest_compressionCodec.cpp:805-812 ("DoubleDelta, ZSTD", "Gorilla, ZSTD", ...)"""
    rng = np.random.Generator(np.random.PCG64(12))
    for codec in _PAIR_CODECS:
        for general in _PAIR_GENERALS:
            v = _pair_values(rng, np.int64 if codec != "GORILLA" else np.float64, 20_001)
            buf = OC.write_multiple_frames(v, getattr(OC, "METHOD_" + codec), getattr(OC, "METHOD_" + general), block_rows=4096)
            assert OC.read_frames(buf) == v.tobytes()
            assert buf[16] == OC.METHOD_MULTIPLE and buf[25:28] == bytes([2, getattr(OC, "METHOD_" + codec), getattr(OC, "METHOD_" + general)])
    raw = _pair_values(rng, np.uint32, 50_000).tobytes()
    assert OC.read_frames(OC.write_frames(raw, 65536, OC.METHOD_ZSTD)) == raw


@pytest.mark.gpu
@pytest.mark.parametrize("general", _PAIR_GENERALS)
@pytest.mark.parametrize("codec", _PAIR_CODECS)
def test_gpu_codec_pairs_decode_to_the_original_column(codec, general):
    """the column codec runs on the device behind the general-purpose stage (LZ4 decoded on the device; ZSTD undone on the host by libzstd,
    the library the reference links -- CompressionCodecZSTD.cpp:60-66 -- before the bytes cross PCIe)"""
    import clickhouse_amd as ch
    ctx = ch.Context(0)
    rng = np.random.Generator(np.random.PCG64(13))
    dts = (np.float64, np.float32, np.uint32) if codec == "GORILLA" else (np.int64, np.uint32, np.int16, np.uint8) if codec != "T64" else (np.int64, np.uint32, np.int16, np.int8)
    for dt in dts:
        for n in (1, 4097, 30_000):
            v = _pair_values(rng, dt, n)
            buf = OC.write_multiple_frames(v, getattr(OC, "METHOD_" + codec), getattr(OC, "METHOD_" + general), block_rows=4096)
            assert OC.read_frames(buf) == v.tobytes()
            got = ch.compression.read_column_file(ctx, buf, dt).numpy()
            assert got.tobytes() == v.tobytes(), (codec, general, dt, n)


@pytest.mark.gpu
def test_gpu_zstd_frames_and_mixed_files():
    import clickhouse_amd as ch
    ctx = ch.Context(0)
    rng = np.random.Generator(np.random.PCG64(14))
    v = _pair_values(rng, np.int64, 200_000)
    raw = v.tobytes()
    z = OC.write_frames(raw, 65536, OC.METHOD_ZSTD)
    assert ch.compression.read_column_file(ctx, z, np.int64).numpy().tobytes() == raw
    # a file whose parts were written under different codecs (ALTER ... MODIFY CODEC leaves such files): frames are independent
    mixed = OC.write_frames(raw[:400_000], 65536, OC.METHOD_LZ4) + OC.write_frames(raw[400_000:800_000], 65536, OC.METHOD_ZSTD) + \
        OC.write_multiple_frames(v[100_000:150_000], OC.METHOD_DOUBLE_DELTA, OC.METHOD_ZSTD) + OC.write_frames(raw[1_200_000:], 1 << 20, OC.METHOD_NONE)
    assert ch.compression.read_column_file(ctx, mixed, np.int64).numpy().tobytes() == raw
    # a damaged zstd payload is CANNOT_DECOMPRESS, and so is a stage header that disagrees with the outer frame
    bad = bytearray(OC.write_frames(raw[:65536], 65536, OC.METHOD_ZSTD))
    bad[16 + 9 + 6] ^= 0xFF
    with pytest.raises(ch.ChgpuError) as e:
        ch.compression.read_column_file(ctx, bytes(bad), np.int64, verify_checksums=False)
    assert e.value.code == ch._capi.ERR_BAD_ARGUMENTS
    m = bytearray(OC.write_multiple_frames(v[:4096], OC.METHOD_T64, OC.METHOD_NONE))
    inner = 16 + 9 + 3 + 9                                           # outer checksum + header, method list, the NONE stage's header: T64's own header
    m[inner + 5] ^= 0x01                                             # ... whose decompressed size no longer matches
    with pytest.raises(ch.ChgpuError) as e:
        ch.compression.read_column_file(ctx, bytes(m), np.int64, verify_checksums=False)
    assert e.value.code == ch._capi.ERR_BAD_ARGUMENTS


# ---- round 3: Native blocks with String / FixedString / Nullable / LowCardinality(String) columns ------------------------------------------
def _varuint(x):
    out = bytearray()
    while x >= 0x80:
        out.append((x & 0x7F) | 0x80)
        x >>= 7
    out.append(x)
    return bytes(out)


def _nstring(b):
    return _varuint(len(b)) + b


def _walk(raw, rev, capacity=8):
    import ctypes as C
    import clickhouse_amd as ch
    from clickhouse_amd import compression as CC
    K = ch._capi
    arr = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
    cols = (CC._NativeColumnStruct * capacity)()
    ncols, nrows, used = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
    bucket, over = C.c_int32(-1), C.c_int(0)
    K.check(K.lib().chgpu_native_walk_block(arr, len(raw), rev, capacity, cols, C.byref(ncols), C.byref(nrows), C.byref(bucket), C.byref(over), C.byref(used)))
    return cols, ncols.value, nrows.value, used.value


def test_native_lowcardinality_blocks_pinned_by_the_reference_test():
    """the four hand-written LowCardinality(String) blocks of 02010_lc_native.python: the valid one is described, the three malformed ones
    are refused with the message 02010_lc_native.reference records"""
    import clickhouse_amd as ch
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "native_lc_blocks.json")))
    for case in kat["cases"]:
        raw = bytes.fromhex(case["block_hex"])
        if case["error"] is None:
            cols, ncols, nrows, used = _walk(raw, kat["server_revision"])
            d = cols[0]
            assert (ncols, nrows, used) == (1, 1, len(raw)) and (d.name, d.type_name, d.kind, d.type, d.lc_num_keys) == (b"x", b"LowCardinality(String)", 3, ch._capi.U64, 1)
            assert raw[d.lc_keys_offset:d.lc_keys_offset + d.lc_keys_bytes] == b"\x05hello" and raw[d.data_offset:d.data_offset + d.data_bytes] == bytes(8)
        else:
            with pytest.raises(ch.ChgpuError) as e:
                _walk(raw, kat["server_revision"])
            assert e.value.code == ch._capi.ERR_BAD_ARGUMENTS and case["error"] in str(e.value), (case["name"], str(e.value))


def _native_block(rev, rows, columns):
    """NativeWriter::write restated: [BlockInfo] columns rows, then per column name, type, [custom-serialization flag], values"""
    blk = bytearray()
    if rev > 0:
        blk += _varuint(1) + bytes([0]) + _varuint(2) + (-1).to_bytes(4, "little", signed=True) + _varuint(0)
    blk += _varuint(len(columns)) + _varuint(rows)
    for name, tname, payload in columns:
        blk += _nstring(name.encode()) + _nstring(tname.encode())
        if rev >= 54454:
            blk += bytes([0])
        blk += payload
    return bytes(blk)


def _lc_payload(keys, indexes, index_dtype):
    code = {1: 0, 2: 1, 4: 2, 8: 3}[np.dtype(index_dtype).itemsize]
    return (1).to_bytes(8, "little") + (code | 0x200).to_bytes(8, "little") + len(keys).to_bytes(8, "little") + b"".join(_nstring(k) for k in keys) + \
        len(indexes).to_bytes(8, "little") + np.asarray(indexes, dtype=index_dtype).tobytes()


def _mixed_block(rng, rows, rev):
    words = [b"", b"a", b"hello", b"x" * 200, "żółw".encode(), b"with\0zero", b"tail "]
    s_vals = [words[i] for i in rng.integers(0, len(words), size=rows)]
    ns_vals = [None if rng.random() < 0.2 else words[i] for i in rng.integers(0, len(words), size=rows)]
    nums = rng.integers(0, 2**63, size=rows, dtype=np.uint64)
    ni = rng.integers(-100, 100, size=rows).astype(np.int32)
    ni_null = (rng.random(rows) < 0.3).astype(np.uint8)
    fs = rng.integers(0, 256, size=(rows, 3), dtype=np.uint8)
    keys = [b"", b"DE", b"FR", b"a longer dictionary value"]            # (the dictionary of a Native block starts with the default value)
    lc_ix = rng.integers(0, len(keys), size=rows)
    nkeys = [b"", b"", b"red", b"green"]                                  # LowCardinality(Nullable(String)): key 0 = NULL, key 1 = the default ''
    nlc_ix = rng.integers(0, len(nkeys), size=rows)
    cols = [
        ("n", "UInt64", nums.tobytes()),
        ("s", "String", b"".join(_nstring(v) for v in s_vals)),
        ("ni", "Nullable(Int32)", ni_null.tobytes() + ni.tobytes()),
        ("fs", "FixedString(3)", fs.tobytes()),
        ("lc", "LowCardinality(String)", _lc_payload(keys, lc_ix, np.uint8) if rows else b""),
        ("ns", "Nullable(String)", bytes(1 if v is None else 0 for v in ns_vals) + b"".join(_nstring(v or b"") for v in ns_vals)),
        ("nlc", "LowCardinality(Nullable(String))", _lc_payload(nkeys, nlc_ix, np.uint16) if rows else b""),
    ]
    want = dict(n=nums, s=s_vals, ni=[None if ni_null[i] else int(ni[i]) for i in range(rows)], fs=[fs[i].tobytes() for i in range(rows)],
                lc=[keys[i] for i in lc_ix], ns=ns_vals, nlc=[None if i == 0 else nkeys[i] for i in nlc_ix])
    return _native_block(rev, rows, cols), want


def test_native_walk_describes_non_numeric_columns():
    rng = np.random.Generator(np.random.PCG64(21))
    for rev in (0, 54454):
        for rows in (0, 1, 500):
            raw, _ = _mixed_block(rng, rows, rev)
            cols, ncols, nrows, used = _walk(raw + b"next", rev)
            assert (ncols, nrows, used) == (7, rows, len(raw))
            assert [cols[c].kind for c in range(7)] == [0, 1, 0, 2, 3, 1, 3] and [cols[c].is_nullable for c in range(7)] == [0, 0, 1, 0, 0, 1, 1]
            assert cols[3].fixed_n == 3 and cols[3].data_bytes == 3 * rows and cols[0].data_bytes == 8 * rows
            if rows:
                assert cols[2].data_offset == cols[2].null_map_offset + rows and cols[4].lc_num_keys == 4 and cols[6].lc_num_keys == 4
            # every truncation of the block is CANNOT_READ_ALL_DATA, never a read past the end
            import clickhouse_amd as ch
            for cut in sorted(set(int(x) for x in rng.integers(1, len(raw), size=12))) if rows else []:
                with pytest.raises(ch.ChgpuError) as e:
                    _walk(raw[:cut], rev)
                assert e.value.code == ch._capi.ERR_BAD_ARGUMENTS


@pytest.mark.gpu
def test_gpu_native_block_with_strings_and_lowcardinality():
    import clickhouse_amd as ch
    from clickhouse_amd import compression as CC
    ctx = ch.Context(0)
    rng = np.random.Generator(np.random.PCG64(22))
    for rows in (0, 1, 3000):
        raw, want = _mixed_block(rng, rows, 54454)
        info, cols = CC.read_native_block(ctx, raw + raw, 0, 54454, described=True)
        assert info["rows"] == rows and info["next_pos"] == len(raw)
        got = dict(cols)
        assert np.array_equal(got["n"].values.numpy() if rows else np.zeros(0, dtype=np.uint64), want["n"])
        if not rows:
            continue
        assert got["s"].strings() == want["s"] and got["fs"].strings() == want["fs"] and got["lc"].strings() == want["lc"]
        assert got["ns"].strings() == want["ns"] and got["nlc"].strings() == want["nlc"]
        nm = got["ni"].null_map.numpy()
        assert [None if nm[i] else int(v) for i, v in enumerate(got["ni"].values.numpy())] == want["ni"]
        # the LowCardinality column is a key column as it stands: GROUP BY its indexes, name the groups through the dictionary
        A = ch.Aggregator(np.uint8, [(ch.AGG_COUNT, None)], ctx=ctx)
        A.execute_on_block(got["lc"].indexes, [None])
        gk, (gc,) = A.convert_to_block()
        keys = [b"", b"DE", b"FR", b"a longer dictionary value"]
        assert {keys[int(k)]: int(c) for k, c in zip(gk, gc)} == {k: want["lc"].count(k) for k in set(want["lc"])}
        # the String column goes through the device dictionary encoder: same groups as the host's
        from clickhouse_amd.lowcardinality import ColumnString
        lc = ColumnString(got["s"].offsets, got["s"].chars).dictionary_encode()
        assert sorted(lc.dictionary) == sorted(set(want["s"])) and [lc.dictionary[int(i)] for i in lc.indexes.numpy()] == want["s"]
