"""and / multiply / plus / minus and the fused SSB Q1.1-style filter+sum — oracle vs numpy on the CPU,
HIP kernels vs oracle on the GPU."""
import numpy as np
import pytest


def _q11_columns(n, seed=1, dtype=np.uint32):
    rng = np.random.Generator(np.random.PCG64(seed))
    od = (19920101 + rng.integers(0, 70000, size=n)).astype(dtype)       # lo_orderdate-like
    disc = rng.integers(0, 11, size=n).astype(dtype)                     # lo_discount
    qty = rng.integers(1, 51, size=n).astype(dtype)                      # lo_quantity
    price = rng.integers(90_000, 10_000_000, size=n).astype(dtype)       # lo_extendedprice
    return od, disc, qty, price


def _q11_preds(M):
    # WHERE lo_orderdate BETWEEN 19930101 AND 19931231 AND lo_discount BETWEEN 1 AND 3 AND lo_quantity < 25
    return [(0, M.GE, 19930101), (0, M.LE, 19931231), (1, M.GE, 1), (1, M.LE, 3), (2, M.LT, 25)]


def test_oracle_arith_and_pipeline_against_numpy(oracle_mod):
    O = oracle_mod
    od, disc, qty, price = _q11_columns(400_003)
    m = (od >= 19930101) & (od <= 19931231) & (disc >= 1) & (disc <= 3) & (qty < 25)
    s, c = O.expr_filter_sum_pipeline([od, disc, qty, price], _q11_preds(O), O.VAL_MUL, 3, 1)
    assert s.dtype == np.uint64 and int(s) == int((price[m].astype(np.uint64) * disc[m]).sum()) and c == int(m.sum())
    s4, c4 = O.expr_filter_sum_pipeline([od, disc, qty, price], _q11_preds(O), O.VAL_MUL, 3, 1, threads=4)
    assert (int(s4), c4) == (int(s), c)
    # result types (NumberTraits.h:73-87): UInt32 * UInt32 -> UInt64, UInt32 - UInt32 -> Int64, Int64 * Int64 wraps
    a = np.array([1, 5, 4_000_000_000], dtype=np.uint32)
    b = np.array([3, 2, 4_000_000_000], dtype=np.uint32)
    assert O.arith(O.VAL_MINUS, a, b).tolist() == [-2, 3, 0] and O.arith(O.VAL_MINUS, a, b).dtype == np.int64
    assert O.arith(O.VAL_MUL, a, b).tolist() == [3, 10, 16_000_000_000_000_000_000] and O.arith(O.VAL_MUL, a, b).dtype == np.uint64
    big = np.array([2**62, -3], dtype=np.int64)
    assert O.arith(O.VAL_MUL, big, np.array([4, 5], dtype=np.int64)).tolist() == [0, -15]
    assert O.and_u8(np.array([1, 1, 0, 0], dtype=np.uint8), np.array([1, 0, 1, 0], dtype=np.uint8)).tolist() == [1, 0, 0, 0]


@pytest.mark.gpu
def test_gpu_and_arith_columns(oracle_mod):
    import clickhouse_amd as ch
    O = oracle_mod
    ctx = ch.Context(0)
    rng = np.random.Generator(np.random.PCG64(2))
    n = 100_003
    for dtype in (np.uint32, np.int32, np.int64, np.uint64):
        info = np.iinfo(dtype)
        a = rng.integers(info.min, info.max, size=n, dtype=dtype, endpoint=True)
        b = rng.integers(info.min, info.max, size=n, dtype=dtype, endpoint=True)
        for op in (ch.VAL_MUL, ch.VAL_PLUS, ch.VAL_MINUS):
            got = ch.arith(op, ctx.upload(a), ctx.upload(b)).numpy()
            want = O.arith(op, a, b)
            assert got.dtype == want.dtype and np.array_equal(got, want), (dtype, op)
    m1 = (rng.integers(0, 2, size=n)).astype(np.uint8)
    m2 = (rng.integers(0, 2, size=n)).astype(np.uint8)
    assert np.array_equal(ch.and_(ctx.upload(m1), ctx.upload(m2)).numpy(), O.and_u8(m1, m2))
    with pytest.raises(ch.ChgpuError) as e:
        ch.arith(ch.VAL_MUL, ctx.upload(rng.random(4)), ctx.upload(rng.random(4)))
    assert e.value.code == ch._capi.ERR_NOT_IMPLEMENTED


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 3, 4099, 2_000_003])
def test_gpu_fused_expression_over_columns_of_different_widths(oracle_mod, n):
    # the real SSB lineorder shape: UInt32 orderdate / extendedprice next to UInt8 discount / quantity (10 B/row), plus
    # signed and 8-byte columns; every predicate is folded in the width of the column it tests
    import clickhouse_amd as ch
    O = oracle_mod
    ctx = ch.Context(0)
    rng = np.random.Generator(np.random.PCG64(n + 11))
    orderdate = rng.integers(19920101, 19981231, size=n).astype(np.uint32)
    discount = rng.integers(0, 11, size=n).astype(np.uint8)
    quantity = rng.integers(1, 51, size=n).astype(np.uint8)
    price = rng.integers(90_000, 10_500_000, size=n).astype(np.uint32)
    cols_np = [orderdate, discount, quantity, price]
    cols = [ctx.upload(c) for c in cols_np]
    preds = lambda M: [(0, M.GE, 19930101), (0, M.LE, 19931231), (1, M.GE, 1), (1, M.LE, 3), (2, M.LT, 25)]
    for vop, va, vb in ((ch.VAL_MUL, 3, 1), (ch.VAL_COL, 3, 0), (ch.VAL_PLUS, 1, 2), (ch.VAL_MINUS, 1, 2), (ch.VAL_MUL, 1, 2)):
        s, c = ch.expr_filter_sum(cols, preds(ch), vop, va, vb)
        so, co = O.expr_filter_sum_pipeline(cols_np, preds(O), vop, va, vb)
        assert s.dtype == so.dtype and (int(s), c) == (int(so), co), (vop, va, vb)
    # signed / 8-byte mix, constants outside a column's range, a Float64 constant against an integer column
    a = rng.integers(-2**31, 2**31, size=n).astype(np.int32)
    b = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)
    u = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    mixed_np = [a, b, u, discount]
    mixed = [ctx.upload(c) for c in mixed_np]
    p2 = lambda M: [(0, M.GT, -2**40, M.I64), (1, M.LT, 2**61), (2, M.GE, 2**63, M.U64), (3, M.LT, 7.5, M.F64), (0, M.NE, 5)]
    for vop, va, vb in ((ch.VAL_MUL, 0, 3), (ch.VAL_MINUS, 0, 1), (ch.VAL_PLUS, 2, 3), (ch.VAL_COL, 1, 0)):
        s, c = ch.expr_filter_sum(mixed, p2(ch), vop, va, vb)
        so, co = O.expr_filter_sum_pipeline(mixed_np, p2(O), vop, va, vb)
        assert s.dtype == so.dtype and (int(s), c) == (int(so), co), (vop, va, vb)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 1023, 65409, 3_000_001])
def test_gpu_fused_q11_matches_oracle_pipeline_and_unfused_path(oracle_mod, n):
    import clickhouse_amd as ch
    O = oracle_mod
    ctx = ch.Context(0)
    cols_np = _q11_columns(n, seed=n + 5)
    cols = [ctx.upload(c) for c in cols_np]
    s, c = ch.expr_filter_sum(cols, _q11_preds(ch), ch.VAL_MUL, 3, 1)
    so, co = O.expr_filter_sum_pipeline(list(cols_np), _q11_preds(O), O.VAL_MUL, 3, 1)
    assert s.dtype == so.dtype and (int(s), c) == (int(so), co)
    if n:
        # the same query through the materialising operators (cmp -> and -> filter -> multiply -> sum) gives the same answer
        mask = None
        for ci, op, sc in _q11_preds(ch):
            m = ch.cmp_const(cols[ci], op, sc)
            mask = m if mask is None else ch.and_(mask, m)
        prod = ch.arith(ch.VAL_MUL, cols[3].filter(mask), cols[1].filter(mask))
        assert prod.size() == co and int(ch.sum_add_many(prod)[0]) == int(so)
    # other value expressions / types
    for vop in (ch.VAL_COL, ch.VAL_PLUS, ch.VAL_MINUS):
        s, c = ch.expr_filter_sum(cols, _q11_preds(ch)[2:], vop, 3, 2)
        so, co = O.expr_filter_sum_pipeline(list(cols_np), _q11_preds(O)[2:], vop, 3, 2)
        assert s.dtype == so.dtype and (int(s), c) == (int(so), co), vop
    i64 = [ctx.upload(cn.astype(np.int64) - 5_000_000) for cn in cols_np[:2]]
    i64_np = [cn.astype(np.int64) - 5_000_000 for cn in cols_np[:2]]
    s, c = ch.expr_filter_sum(i64, [(1, ch.LT, -4_999_995), (0, ch.GT, 0)], ch.VAL_MUL, 0, 1)
    so, co = O.expr_filter_sum_pipeline(i64_np, [(1, O.LT, -4_999_995), (0, O.GT, 0)], O.VAL_MUL, 0, 1)
    assert (int(s), c) == (int(so), co)
