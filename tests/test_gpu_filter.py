"""GPU parity, filter path: HIP kernels through the C ABI vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ch():
    import clickhouse_amd
    return clickhouse_amd


@pytest.fixture(scope="module")
def ctx(ch):
    c = ch.Context(0)
    yield c
    c.close()


DTYPES = [np.int64, np.uint64, np.uint32, np.int32, np.float64, np.uint8, np.uint16, np.int16, np.int8, np.float32]


def _rand(rng, dtype, n, lo=0, hi=255):
    if np.dtype(dtype).kind == "f":
        return (rng.random(n) * (hi - lo) + lo).astype(dtype)
    return rng.integers(lo, hi, size=n).astype(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_filter_property_like_gtest_column_vector(ch, ctx, oracle_mod, dtype):
    rng = np.random.Generator(np.random.PCG64(11))
    for ratio in (1, 2, 5, 11, 32, 64, 100, 1000):
        for _ in range(3):
            rows = int(rng.integers(1, 10000))
            data = _rand(rng, dtype, rows)
            filt = ((rng.integers(0, ratio, size=rows) == 0) * rng.integers(1, 255, size=rows)).astype(np.uint8)
            got = ctx.upload(data).filter(ctx.upload(filt)).numpy()
            want = oracle_mod.filter_column(data, filt)
            assert got.dtype == want.dtype and np.array_equal(got, want)
            assert ch.count_bytes_in_filter(ctx.upload(filt)) == oracle_mod.count_bytes_in_filter(filt)


@pytest.mark.parametrize("rows", [0, 1, 63, 64, 65, 1023, 1024, 1025, 65409, 1_000_003])
def test_filter_ragged_sizes_and_unaligned_views(ch, ctx, oracle_mod, rows):
    rng = np.random.Generator(np.random.PCG64(rows + 1))
    data = rng.integers(-2**62, 2**62, size=rows + 3, dtype=np.int64)
    filt = (rng.integers(0, 10, size=rows + 3) == 0).astype(np.uint8)
    col, m = ctx.upload(data), ctx.upload(filt)
    for start in (0, 1, 3):  # views that are 8-B but not 16-B aligned
        n = rows
        got = col.cut(start, n).filter(m.cut(start, n)).numpy()
        assert np.array_equal(got, oracle_mod.filter_column(data[start:start + n], filt[start:start + n]))


@pytest.mark.parametrize("n_u32,n_8", [(2, 0), (5, 2), (7, 5), (1, 3), (4, 4)])
def test_filter_columns_same_width_columns_share_one_mask_pass(ch, ctx, oracle_mod, n_u32, n_8):
    """columns of one element width are compacted four (three, two) at a time by one kernel that reads the mask once; every batch
    size, a ragged last chunk, views whose first row is not 16-byte aligned, and masks that keep nothing / everything"""
    rng = np.random.Generator(np.random.PCG64(100 * n_u32 + n_8))
    n = 2_500_013
    cols_np = [rng.integers(0, 2**32, size=n, dtype=np.uint32) for _ in range(n_u32)]
    cols_np += [rng.integers(-2**62, 2**62, size=n, dtype=np.int64) if i % 2 == 0 else rng.random(n) for i in range(n_8)]
    cols_np.insert(1, rng.integers(0, 2**16, size=n).astype(np.uint16))        # an odd width between them
    for filt in ((rng.integers(0, 5, size=n) == 0).astype(np.uint8), np.zeros(n, dtype=np.uint8), np.full(n, 7, dtype=np.uint8)):
        dev = [ctx.upload(c) for c in cols_np]
        m = ctx.upload(filt)
        outs = ch.filter_columns(dev, m)
        for got, c in zip(outs, cols_np):
            assert np.array_equal(got.numpy(), c[filt != 0])
        if filt[0] == 0 and filt.any():
            for start in (1, 3):                                                  # unaligned views
                outs = ch.filter_columns([d.cut(start, n - start) for d in dev], m.cut(start, n - start))
                for got, c in zip(outs, cols_np):
                    assert np.array_equal(got.numpy(), c[start:][filt[start:] != 0])


def test_filter_columns_of_a_block_with_one_mask(ch, ctx, oracle_mod):
    rng = np.random.Generator(np.random.PCG64(19))
    for n in (0, 1, 1025, 300_007):
        cols_np = [rng.integers(-2**62, 2**62, size=n, dtype=np.int64), rng.integers(0, 2**32, size=n, dtype=np.uint32),
                   rng.integers(0, 256, size=n).astype(np.uint8), rng.random(n)]
        filt = (rng.integers(0, 4, size=n) == 0).astype(np.uint8)
        outs = ch.filter_columns([ctx.upload(c) for c in cols_np], ctx.upload(filt))
        assert len(outs) == 4
        for got, c in zip(outs, cols_np):
            want = oracle_mod.filter_column(c, filt)
            assert got.numpy().dtype == want.dtype and np.array_equal(got.numpy(), want)
    assert ch.filter_columns([], ctx.upload(np.ones(5, dtype=np.uint8))) == []
    with pytest.raises(ch.ChgpuError) as e:
        ch.filter_columns([ctx.upload(np.arange(4, dtype=np.int64)), ctx.upload(np.arange(5, dtype=np.int64))], ctx.upload(np.ones(4, dtype=np.uint8)))
    assert e.value.code == ch._capi.ERR_SIZES_MISMATCH


def test_filter_size_mismatch_is_an_error(ch, ctx):
    col = ctx.upload(np.arange(10, dtype=np.int64))
    m = ctx.upload(np.ones(9, dtype=np.uint8))
    with pytest.raises(ch.ChgpuError) as e:
        col.filter(m)
    assert e.value.code == ch._capi.ERR_SIZES_MISMATCH and "doesn't match" in str(e.value)


def test_cmp_const_all_ops_and_mixed_signedness(ch, ctx, oracle_mod):
    O = oracle_mod
    rng = np.random.Generator(np.random.PCG64(5))
    ops = [ch.EQ, ch.NE, ch.LT, ch.GT, ch.LE, ch.GE]
    a64 = np.concatenate([rng.integers(-50, 50, size=5000), np.array([-2**63, 2**63 - 1, 0, -1, 1])]).astype(np.int64)
    u64 = np.concatenate([rng.integers(0, 100, size=5000).astype(np.uint64), np.array([0, 2**64 - 1, 2**63], dtype=np.uint64)])
    u32 = rng.integers(0, 2**32, size=5001, dtype=np.uint32)
    i32 = rng.integers(-2**31, 2**31, size=5003).astype(np.int32)
    cases = [
        (a64, O.I64, [0, -7, 49, -2**63, 2**63 - 1]), (a64, O.U64, [0, 7, 2**63, 2**64 - 1]),
        (u64, O.U64, [0, 50, 2**63, 2**64 - 1]), (u64, O.I64, [-1, 0, 50, -2**63, 2**63 - 1]),
        (u32, O.U64, [0, 2**31, 2**32 - 1, 2**40]), (u32, O.I64, [-5, 2**31]),
        (i32, O.I64, [-2**31, 0, 2**31 - 1, -2**40, 2**40]), (i32, O.U64, [0, 5, 2**63]),
    ]
    u16 = rng.integers(0, 2**16, size=5001).astype(np.uint16)
    i16 = rng.integers(-2**15, 2**15, size=5003).astype(np.int16)
    i8 = rng.integers(-128, 128, size=5005).astype(np.int8)
    cases += [(u16, O.U16, [0, 1, 40000, 65535]), (u16, O.I64, [-1, 65535, 65536]), (i16, O.I16, [-32768, -1, 0, 32767]), (i16, O.U64, [0, 32767, 2**63]),
              (i8, O.I8, [-128, -1, 0, 127]), (i8, O.I64, [-129, 128, 5]), (u16, O.F64, [0.5, 65535.5, -1.0, float("nan")]), (i8, O.F64, [-128.5, 126.5])]
    for arr, stag, scalars in cases:
        col = ctx.upload(arr)
        for s in scalars:
            for op in ops:
                got = ch.cmp_const(col, op, s, stag).numpy()
                want = O.cmp_const(arr, op, s, stag)
                assert np.array_equal(got, want), (arr.dtype, stag, s, op)
    f = np.concatenate([rng.standard_normal(4099), np.array([np.nan, np.inf, -np.inf, 0.0, -0.0])])
    fc = ctx.upload(f)
    for s in (0.0, 0.5, np.nan, np.inf):
        for op in ops:
            assert np.array_equal(ch.cmp_const(fc, op, s).numpy(), O.cmp_const(f, op, s)), (s, op)
    # integer column vs Float64 constant and Float64 column vs integer constant: the mathematical comparison of
    # accurate::lessOp / equalsOp (AccurateComparison.h:20-130), incl. constants no integer / no double can equal
    fconsts = [0.0, -0.0, 1.5, -0.5, -7.0, 49.0, 49.000001, 2.0**31, 2.0**53, 2.0**53 + 2, 2.0**63, -2.0**63, 2.0**64, 1e19, -1e19,
               1.8446744073709552e19, 3e38, -3e38, np.inf, -np.inf, np.nan]
    big = np.array([2**53, 2**53 + 1, 2**53 + 2, 2**63 - 1, -2**63, -2**53 - 1], dtype=np.int64)
    ubig = np.array([2**53 + 1, 2**63, 2**64 - 1, 2**64 - 1025], dtype=np.uint64)
    for arr in (a64, np.concatenate([a64, big]), np.concatenate([u64, ubig]), u32, i32):
        col = ctx.upload(arr)
        for c in fconsts:
            for op in ops:
                got = ch.cmp_const(col, op, c, ch.F64).numpy()
                want = O.cmp_const(arr, op, c, O.F64)
                assert np.array_equal(got, want), (arr.dtype, c, op)
    fx = np.concatenate([f, np.array([2.0**53, 2.0**53 + 2, 2.0**63, -2.0**63, 2.0**64, 1.8446744073709552e19, 9007199254740993.0, 1e300, -1e300])])
    fxc = ctx.upload(fx)
    for stag, scalars in ((O.I64, [0, -1, 7, 2**53, 2**53 + 1, 2**63 - 1, -2**63, -2**53 - 1]), (O.U64, [0, 5, 2**53 + 1, 2**63, 2**64 - 1]),
                          (O.U32, [0, 2**32 - 1]), (O.I32, [-2**31, 3])):
        for sc in scalars:
            for op in ops:
                got = ch.cmp_const(fxc, op, sc, stag).numpy()
                want = O.cmp_const(fx, op, sc, stag)
                assert np.array_equal(got, want), (stag, sc, op)


@pytest.mark.parametrize("dtype", [np.int64, np.uint64, np.uint32, np.int32, np.uint8, np.uint16, np.int16, np.int8])
def test_sum_integers_bit_exact_with_wraparound(ch, ctx, oracle_mod, dtype):
    rng = np.random.Generator(np.random.PCG64(6))
    info = np.iinfo(dtype)
    for n in (0, 1, 7, 1000, 65409, 300_001):
        a = rng.integers(info.min, info.max, size=n, dtype=dtype, endpoint=True)
        col = ctx.upload(a)
        got = ch.sum_add_many(col)
        want = oracle_mod.sum_add_many(a)
        assert got.dtype == want.dtype and got[0] == want[0]
        if n > 10:
            st = np.array([123], dtype=got.dtype)
            assert ch.sum_add_many(col, 3, n - 2, st)[0] == oracle_mod.sum_add_many(a, 3, n - 2, np.array([123], dtype=got.dtype))[0]
            cond = (rng.integers(0, 3, size=n) == 0).astype(np.uint8) * 7
            assert ch.sum_add_many_conditional(col, ctx.upload(cond))[0] == oracle_mod.sum_add_many_conditional(a, cond)[0]


def test_sum_float64_within_1e6_relative(ch, ctx, oracle_mod):
    rng = np.random.Generator(np.random.PCG64(4))
    for n in (1, 1000, 65409, 2_000_003):
        a = rng.random(n)
        got = float(ch.sum_add_many(ctx.upload(a))[0])
        want = float(oracle_mod.sum_add_many(a)[0])
        assert abs(got - want) <= 1e-6 * abs(want)  # BASELINE.json north_star tolerance for sum/avg(Float64)
    # run-to-run reproducible (fixed grid, fixed fold order)
    col = ctx.upload(rng.standard_normal(1_000_000) * 1e6)
    assert len({float(ch.sum_add_many(col)[0]) for _ in range(3)}) == 1
    a = np.array([1.0, np.nan, 2.0])
    assert np.isnan(ch.sum_add_many(ctx.upload(a))[0])


def test_float32_columns_compare_sum_and_avg(ch, ctx, oracle_mod):
    # Float32: comparisons after the exact widening to double (against Float32, Float64 and integer constants), sums and
    # averages accumulated in Float64 (SumSimple: NearestFieldType<Float32>)
    O = oracle_mod
    rng = np.random.Generator(np.random.PCG64(32))
    f = np.concatenate([(rng.standard_normal(300_001) * 100).astype(np.float32), np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 16777216.0, 16777218.0], dtype=np.float32)])
    fc = ctx.upload(f)
    ops = [ch.EQ, ch.NE, ch.LT, ch.GT, ch.LE, ch.GE]
    for sc, tag in ((0.5, ch.F32), (0.1, ch.F64), (float(np.float32(0.1)), ch.F64), (np.nan, ch.F32), (16777217, ch.I64), (-3, ch.I32), (2**63, ch.U64), (np.inf, ch.F64)):
        for op in ops:
            assert np.array_equal(ch.cmp_const(fc, op, sc, tag).numpy(), O.cmp_const(f, op, sc, tag)), (sc, tag, op)
    pos = np.abs(f[np.isfinite(f)]).astype(np.float32)
    pc = ctx.upload(pos)
    s = ch.sum_add_many(pc)[0]
    so = O.sum_add_many(pos)[0]
    assert s.dtype == np.float64 and abs(float(s) - float(so)) <= 1e-6 * abs(float(so))
    s2, c2 = ch.filter_sum(pc, ch.LT, 50.0, scalar_tag=ch.F64)
    m = pos.astype(np.float64) < 50.0
    assert c2 == int(m.sum()) and abs(float(s2) - pos[m].astype(np.float64).sum()) <= 1e-6 * pos[m].astype(np.float64).sum()
    k = rng.integers(0, 50, size=pos.shape[0]).astype(np.uint32)
    g = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.float32), (ch.AGG_AVG, np.float32), (ch.AGG_COUNT, None)], ctx=ctx)
    g.execute_on_block(k, [pos, pos, None])
    o = O.Aggregator(np.uint32, [(O.AGG_SUM, np.float32), (O.AGG_AVG, np.float32), (O.AGG_COUNT, None)])
    o.execute_on_block(k, [pos, pos, None])
    gk, gr = g.convert_to_block()
    ok, orr = o.convert_to_block()
    gi, oi = np.argsort(gk), np.argsort(ok)
    assert np.array_equal(gk[gi], ok[oi]) and np.array_equal(gr[2][gi], orr[2][oi])
    assert gr[0].dtype == np.float64 and np.allclose(gr[0][gi], orr[0][oi], rtol=1e-6) and np.allclose(gr[1][gi], orr[1][oi], rtol=1e-6)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 1000, 65409, 10_000_000])
def test_fused_filter_sum_matches_block_pipeline_c1(ch, ctx, oracle_mod, n):
    # BASELINE.json configs[0] shape: Int64 uniform [0,2^31), seed 1, a < 214748365 (~10 %)
    rng = np.random.Generator(np.random.PCG64(1))
    a = rng.integers(0, 2**31, size=n, dtype=np.int64)
    s, c = ch.filter_sum(ctx.upload(a), ch.LT, 214748365)
    so, co, _, _ = oracle_mod.filter_sum_pipeline(a, oracle_mod.LT, 214748365)
    assert (int(s), c) == (int(so), co)


def test_fused_filter_sum_two_columns_types_and_views(ch, ctx, oracle_mod):
    rng = np.random.Generator(np.random.PCG64(2))
    n = 1_000_001
    for dtype in (np.int64, np.uint64, np.uint32, np.int32):
        info = np.iinfo(dtype)
        b = rng.integers(info.min, info.max, size=n, dtype=dtype, endpoint=True)
        a = rng.integers(info.min, info.max, size=n, dtype=dtype, endpoint=True)
        thr = int(np.quantile(b.astype(np.float64), 0.1))
        bc, ac = ctx.upload(b), ctx.upload(a)
        for op in (ch.LT, ch.GE, ch.EQ, ch.NE):
            s, c = ch.filter_sum(bc, op, thr, ac)
            so, co, _, _ = oracle_mod.filter_sum_pipeline(b, op, thr, a)
            assert (int(s), c) == (int(so), co), (dtype, op)
        # 8-byte-but-not-16-byte aligned views take the scalar-load kernel; same answer
        s, c = ch.filter_sum(bc.cut(1, n - 2), ch.LT, thr, ac.cut(1, n - 2))
        so, co, _, _ = oracle_mod.filter_sum_pipeline(b[1:n - 1], oracle_mod.LT, thr, a[1:n - 1])
        assert (int(s), c) == (int(so), co)
    f = rng.random(n)
    s, c = ch.filter_sum(ctx.upload(f), ch.LT, 0.1)
    so, co, _, _ = oracle_mod.filter_sum_pipeline(f, oracle_mod.LT, 0.1)
    assert c == co and abs(float(s) - float(so)) <= 1e-6 * abs(float(so))
    with pytest.raises(ch.ChgpuError) as e:
        ch.filter_sum(ctx.upload(a[:10]), ch.LT, 1, ctx.upload(a[:9]))
    assert e.value.code == ch._capi.ERR_SIZES_MISMATCH


def test_fused_filter_sum_predicate_and_value_of_different_types(ch, ctx, oracle_mod):
    # WHERE over one type, sum over another: mask + conditional sum on the device (addManyConditional semantics);
    # also a Float64 threshold against an integer predicate column
    O = oracle_mod
    rng = np.random.Generator(np.random.PCG64(21))
    n = 700_003
    preds = {np.int32: rng.integers(-1000, 1000, size=n).astype(np.int32), np.uint8: rng.integers(0, 256, size=n).astype(np.uint8),
             np.float64: rng.standard_normal(n), np.uint64: rng.integers(0, 2**63, size=n, dtype=np.uint64)}
    vals = {np.int64: rng.integers(-2**62, 2**62, size=n, dtype=np.int64), np.uint32: rng.integers(0, 2**32, size=n, dtype=np.uint32),
            np.float64: rng.random(n)}
    for pdt, p in preds.items():
        for vdt, v in vals.items():
            if pdt == vdt:
                continue
            for op, thr, tag in ((ch.LT, np.quantile(p.astype(np.float64), 0.3), None), (ch.GE, 0.5, ch.F64)):
                scalar = float(thr) if (tag == ch.F64 or pdt == np.float64) else int(thr)
                s, c = ch.filter_sum(ctx.upload(p), op, scalar, ctx.upload(v), scalar_tag=tag)
                mask = O.cmp_const(p, op, scalar, tag).astype(bool)
                assert c == int(mask.sum()), (pdt, vdt, op)
                if vdt == np.float64:
                    want = float(v[mask].sum())
                    assert abs(float(s) - want) <= 1e-6 * abs(want) + 1e-12
                else:
                    want = int(v[mask].astype(np.uint64).sum(dtype=np.uint64)) if vdt == np.uint32 else int(v[mask].view(np.uint64).sum(dtype=np.uint64))
                    assert int(np.array(s).astype(np.uint64)) == want % 2**64, (pdt, vdt, op)   # modulo 2^64
    r = ctx.upload(np.zeros(2, dtype=np.uint64))
    ch.filter_sum_async(ctx.upload(preds[np.int32]), ch.LT, 0, ctx.upload(vals[np.int64]), r)
    ctx.synchronize()
    got = r.numpy()
    m = preds[np.int32] < 0
    assert int(got[1]) == int(m.sum()) and int(got[0]) == int(vals[np.int64][m].view(np.uint64).sum(dtype=np.uint64))


def test_filter_description_nullable(ch, ctx, oracle_mod):
    rng = np.random.Generator(np.random.PCG64(3))
    d = rng.integers(0, 3, size=10007).astype(np.uint8)
    nm = rng.integers(0, 2, size=10007).astype(np.uint8)
    got = ch.filter_description_nullable(ctx.upload(d), ctx.upload(nm)).numpy()
    want = np.zeros_like(d)
    oracle_mod.lib().cho_filter_description_nullable(d.ctypes.data, nm.ctypes.data, d.shape[0], want.ctypes.data)
    assert np.array_equal(got, want)


def test_index_and_replicate(ch, ctx, oracle_mod):
    rng = np.random.Generator(np.random.PCG64(9))
    for dtype in (np.int64, np.uint32, np.uint8, np.float64):
        data = _rand(rng, dtype, 10000)
        col = ctx.upload(data)
        idx = rng.integers(0, 10000, size=5000, dtype=np.uint64)
        assert np.array_equal(col.index(ctx.upload(idx)).numpy(), oracle_mod.index_column(data, idx))
        assert np.array_equal(col.index(ctx.upload(idx.astype(np.uint32)), limit=100).numpy(), oracle_mod.index_column(data, idx, 100))
        cnt = rng.integers(0, 4, size=10000)
        off = np.cumsum(cnt).astype(np.uint64)
        assert np.array_equal(col.replicate(ctx.upload(off)).numpy(), oracle_mod.replicate(data, off))
    miss = np.array([0, 2**64 - 1, 5], dtype=np.uint64)
    got = ctx.upload(np.arange(1, 11, dtype=np.int64)).index(ctx.upload(miss), default_for_missing=True).numpy()
    assert got.tolist() == [1, 0, 6]


def test_size_independent_properties_at_scale(ch, ctx):
    # 2^27 rows: linearity of sum over a split, count(p) + count(!p) == n, filter keeps order and sums match
    import torch
    n = 1 << 27
    g = torch.Generator(device="cuda").manual_seed(1)
    t = torch.randint(0, 2**31, (n,), dtype=torch.int64, device="cuda", generator=g)
    torch.cuda.synchronize()
    col = ctx.wrap(t.data_ptr(), np.int64, n, keepalive=t)
    thr = 214748365
    s_lt, c_lt = ch.filter_sum(col, ch.LT, thr)
    s_ge, c_ge = ch.filter_sum(col, ch.GE, thr)
    total = ch.sum_add_many(col)[0]
    assert c_lt + c_ge == n and int(s_lt) + int(s_ge) == int(total) == int(t.sum().item())
    half = n // 2 + 1
    assert int(ch.sum_add_many(col, 0, half)[0]) + int(ch.sum_add_many(col, half, n)[0]) == int(total)
    mask = ch.cmp_const(col, ch.LT, thr)
    assert ch.count_bytes_in_filter(mask) == c_lt
    kept = col.filter(mask)
    assert kept.size() == c_lt and int(ch.sum_add_many(kept)[0]) == int(s_lt)
    # order preserved: equals torch's boolean-mask selection
    assert np.array_equal(kept.numpy(), t[t < thr].cpu().numpy())


@pytest.mark.gpu
def test_replicate_columns_matches_numpy_repeat():
    """joinBlock's replicate over every column of a Block in one call: columns of one width share a kernel (2, 3, 4 and more of them),
    the other widths go one by one; zero-length and long runs, an empty Block, a size mismatch"""
    import clickhouse_amd as ch
    ctx = ch.Context()
    rng = np.random.Generator(np.random.PCG64(77))
    n = 200_003
    counts = rng.choice([0, 0, 0, 1, 1, 2, 7], size=n).astype(np.uint64)
    counts[5] = 1000
    off = np.cumsum(counts).astype(np.uint64)
    dts = [np.uint32, np.int64, np.uint32, np.uint8, np.float64, np.uint32, np.uint16, np.int32, np.uint32, np.uint32, np.uint64]
    cols = [rng.integers(0, 200, size=n).astype(dt) for dt in dts]
    for k in (1, 2, 3, 5, len(cols)):
        outs = ch.replicate_columns([ctx.upload(c) for c in cols[:k]], ctx.upload(off))
        for c, o in zip(cols[:k], outs):
            assert np.array_equal(o.numpy(), np.repeat(c, counts.astype(np.int64)))
    outs = ch.replicate_columns([ctx.upload(cols[0][:0]), ctx.upload(cols[1][:0])], ctx.upload(off[:0]))
    assert [o.size() for o in outs] == [0, 0]
    outs = ch.replicate_columns([ctx.upload(cols[0]), ctx.upload(cols[2])], ctx.upload(np.zeros(n, dtype=np.uint64)))
    assert [o.size() for o in outs] == [0, 0]
    with pytest.raises(ch.ChgpuError) as e:
        ch.replicate_columns([ctx.upload(cols[0]), ctx.upload(cols[1][:10])], ctx.upload(off))
    assert e.value.code == ch._capi.ERR_SIZES_MISMATCH
