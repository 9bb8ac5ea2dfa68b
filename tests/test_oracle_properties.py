"""Oracle vs the properties the reference's gtests check (restated; the gtests themselves cannot be compiled here)."""
import numpy as np
import pytest


@pytest.mark.parametrize("dtype", [np.int64, np.uint64, np.uint32, np.float64, np.uint8, np.int32])
def test_filter_property_like_gtest_column_vector(oracle_mod, dtype):
    # src/Columns/tests/gtest_column_vector.cpp:41-104: filtered == in-order subsequence where mask != 0
    rng = np.random.Generator(np.random.PCG64(1))
    for ratio in (1, 2, 5, 11, 32, 64, 100, 1000):
        for _ in range(6):
            rows = int(rng.integers(1, 10000))
            data = rng.integers(0, 255, size=rows).astype(dtype)
            filt = (rng.integers(0, ratio, size=rows) == 0).astype(np.uint8) * rng.integers(1, 255, size=rows).astype(np.uint8)
            got = oracle_mod.filter_column(data, filt)
            assert np.array_equal(got, data[filt != 0])
            assert oracle_mod.count_bytes_in_filter(filt) == int((filt != 0).sum())


def test_filter_prefix_suffix_fast_paths_and_errors(oracle_mod):
    data = np.arange(64 * 5 + 7, dtype=np.int64)
    for filt in (np.ones_like(data, dtype=np.uint8), np.zeros_like(data, dtype=np.uint8)):
        assert np.array_equal(oracle_mod.filter_column(data, filt), data[filt != 0])
    f = np.zeros(data.shape[0], dtype=np.uint8)
    f[:17] = 1          # prefix block
    f[64 + 40:128] = 1  # suffix block
    f[130] = f[190] = 7  # sparse block, non-0/1 bytes count as set (ColumnVector.cpp:716)
    assert np.array_equal(oracle_mod.filter_column(data, f), data[f != 0])
    with pytest.raises(ValueError):
        oracle_mod.filter_column(data, f[:-1])
    assert oracle_mod.filter_column(data[:0], f[:0]).shape[0] == 0


def test_bytes64_mask(oracle_mod):
    rng = np.random.Generator(np.random.PCG64(2))
    for _ in range(200):
        b = (rng.integers(0, 3, size=64) * rng.integers(0, 255, size=64)).astype(np.uint8)
        want = sum((1 << i) for i in range(64) if b[i] != 0)
        assert oracle_mod.lib().cho_bytes64MaskToBits64Mask(b.ctypes.data) == want


def test_hash_table_scenarios_like_gtest_hash_table(oracle_mod):
    # src/Common/tests/gtest_hash_table.cpp:50-140: insert/emplace/find incl. zero key; iteration covers all
    m = oracle_mod.HashMap()
    assert len(m) == 0 and m.find(1) is None and m.find(0) is None
    assert m.emplace(1, 10) and not m.emplace(1, 99)
    assert m.find(1) == 10
    assert m.emplace(0, 5) and m.has_zero and m.find(0) == 5 and not m.emplace(0, 6)
    assert len(m) == 2
    keys, vals = m.dump()
    assert keys[0] == 0  # zero key first in iteration order (HashTable.h:830-845)
    # growth: 256 cells, max fill 128, x4 until 2^23 (HashTable.h:246-249)
    m2 = oracle_mod.HashMap()
    assert m2.buf_size == 256
    for k in range(1, 129):
        m2.emplace(k, k)
    assert m2.buf_size == 256
    m2.emplace(1000, 1)
    assert m2.buf_size == 1024
    for k in range(2000, 2000 + 600):
        m2.emplace(k, k)
    assert m2.buf_size == 4096
    for k in range(1, 129):
        assert m2.find(k) == k
    assert m2.find(77777) is None
    ks, vs = m2.dump()
    assert len(ks) == len(m2) == 129 + 600 and len(set(ks.tolist())) == len(ks)
    m3 = oracle_mod.HashMap()
    m3.reserve(1000)  # Grower::set: log2(999)+2 = 11
    assert m3.buf_size == 2048


def test_sum_semantics(oracle_mod):
    # Int64 wraps modulo 2^64 (AggregateFunctionSum.h:36-39)
    a = np.array([2**62, 2**62, 2**62, 2**62, 5], dtype=np.int64)
    assert int(oracle_mod.sum_add_many(a)[0]) == 5
    u = np.array([2**32 - 1] * 3, dtype=np.uint32)
    s = oracle_mod.sum_add_many(u)
    assert s.dtype == np.uint64 and int(s[0]) == 3 * (2**32 - 1)
    # Float64: 16 lanes then tail, exactly
    rng = np.random.Generator(np.random.PCG64(3))
    x = rng.random(1000)
    lanes = np.zeros(16)
    body = x[: 1000 // 16 * 16].reshape(-1, 16)
    for row in body:
        lanes += row
    want = 0.0
    for v in lanes:
        want += v
    tail = 0.0
    for v in x[1000 // 16 * 16:]:
        tail += v
    want += tail
    assert float(oracle_mod.sum_add_many(x)[0]) == want
    cond = (rng.integers(0, 2, size=1000)).astype(np.uint8)
    got = float(oracle_mod.sum_add_many_conditional(x, cond)[0])
    assert abs(got - x[cond != 0].sum()) < 1e-9
    ai = rng.integers(-2**40, 2**40, size=1000)
    assert int(oracle_mod.sum_add_many_conditional(ai, cond)[0]) == int(ai[cond != 0].sum())


def test_comparison_semantics(oracle_mod):
    O = oracle_mod
    a = np.array([-1, 0, 1, 2**62], dtype=np.int64)
    # mixed signedness compared mathematically (AccurateComparison.h:36-45)
    assert O.cmp_const(a, O.LT, 1, O.U64).tolist() == [1, 1, 0, 0]
    assert O.cmp_const(a, O.EQ, 2**62, O.U64).tolist() == [0, 0, 0, 1]
    u = np.array([0, 5, 2**63 + 5], dtype=np.uint64)
    assert O.cmp_const(u, O.GT, -1, O.I64).tolist() == [1, 1, 1]
    f = np.array([np.nan, 1.0, -np.inf], dtype=np.float64)
    for op, want in ((O.LT, [0, 1, 1]), (O.GT, [0, 0, 0]), (O.LE, [0, 1, 1]), (O.GE, [0, 0, 0]), (O.EQ, [0, 0, 0]), (O.NE, [1, 1, 1])):
        assert O.cmp_const(f, op, 2.0).tolist() == want
    # Int64 vs Float64 exact (DecomposedFloat): 2^53+1 is not representable
    big = np.array([2**53, 2**53 + 1], dtype=np.int64)
    assert O.cmp_const(big, O.GT, float(2**53), O.F64).tolist() == [0, 1]
    assert O.cmp_const(big, O.EQ, float(2**53), O.F64).tolist() == [1, 0]


def test_pipeline_semantics_c1_shape(oracle_mod):
    # FilterTransform drops empty chunks, passes all-true chunks through; result == numpy
    rng = np.random.Generator(np.random.PCG64(1))
    a = rng.integers(0, 2**31, size=300000, dtype=np.int64)
    s, c, dropped, passed = oracle_mod.filter_sum_pipeline(a, oracle_mod.LT, 214748365, block_rows=65409)
    sel = a[a < 214748365]
    assert int(s) == int(sel.sum()) and c == sel.shape[0] and dropped == 0 and passed == 0
    s4, c4, _, _ = oracle_mod.filter_sum_pipeline(a, oracle_mod.LT, 214748365, threads=4)
    assert (int(s4), c4) == (int(s), c)
    z = np.zeros(200000, dtype=np.int64)
    s, c, dropped, passed = oracle_mod.filter_sum_pipeline(z, oracle_mod.LT, 0)
    assert (int(s), c, dropped) == (0, 0, 4)
    s, c, dropped, passed = oracle_mod.filter_sum_pipeline(z, oracle_mod.EQ, 0)
    assert (c, passed) == (200000, 4)
    # no-key aggregation over an empty input still yields one row (sum 0, count 0)
    s, c, _, _ = oracle_mod.filter_sum_pipeline(z[:0], oracle_mod.EQ, 0)
    assert (int(s), c) == (0, 0)


def test_aggregator_two_level_and_merge(oracle_mod):
    O = oracle_mod
    rng = np.random.Generator(np.random.PCG64(2))
    k = rng.integers(0, 150000, size=400000, dtype=np.uint32)
    k[:10] = 0  # zero key is a legal key
    v = rng.integers(-2**31, 2**31, size=400000, dtype=np.int64)
    parts = []
    for t in range(3):
        a = O.Aggregator(np.uint32, [(O.AGG_SUM, np.int64), (O.AGG_COUNT, None), (O.AGG_AVG, np.int64)],
                         two_level_threshold=50000 if t != 1 else 0)
        lo, hi = t * 130000, min(400000, (t + 1) * 130000 + 10000)
        for b in range(lo, hi, 65409):
            a.execute_on_block(k[b:min(hi, b + 65409)], [v[b:min(hi, b + 65409)], None, v[b:min(hi, b + 65409)]])
        parts.append((a, lo, hi))
    assert parts[0][0].is_two_level and not parts[1][0].is_two_level
    dst = parts[0][0]
    dst.merge(parts[1][0])
    dst.merge(parts[2][0])
    keys, (s, c, avg) = dst.convert_to_block()
    idx = np.concatenate([np.arange(lo, hi) for _, lo, hi in parts])
    kk, vv = k[idx], v[idx]
    uk, inv = np.unique(kk, return_inverse=True)
    want_s = np.zeros(uk.shape[0], dtype=np.int64)
    np.add.at(want_s, inv, vv)
    want_c = np.bincount(inv).astype(np.uint64)
    order = np.argsort(keys)
    assert np.array_equal(keys[order], uk)
    assert np.array_equal(s[order], want_s) and np.array_equal(c[order], want_c)
    assert np.allclose(avg[order], want_s / want_c, rtol=1e-12)


def test_join_strictness_matrix_and_max_rows(oracle_mod):
    O = oracle_mod
    right1 = np.array([5, 7, 5, 0, 9, 5], dtype=np.uint64)
    right2 = np.array([7, 11, 0], dtype=np.uint64)
    left = np.array([5, 1, 0, 7, 7, 12, 9], dtype=np.uint64)

    def run(kind, strict, **kw):
        j = O.HashJoin(kind, strict, **kw)
        j.add_block(right1)
        j.add_block(right2)
        return j, j.joined_pairs(left)

    j, (l, b, r, c) = run(O.JOIN_INNER, O.STRICT_ALL)
    got = sorted(zip(l.tolist(), b.tolist(), r.tolist()))
    want = sorted([(0, 0, 0), (0, 0, 2), (0, 0, 5), (2, 0, 3), (2, 1, 2), (3, 0, 1), (3, 1, 0), (4, 0, 1), (4, 1, 0), (6, 0, 4)])
    assert got == want and c == 7
    # first element for a left row is the first-inserted right row (RowRefs.h:66-108)
    assert (l[0], b[0], r[0]) == (0, 0, 0)
    j, (l, b, r, c) = run(O.JOIN_LEFT, O.STRICT_ALL)
    assert sorted(zip(l.tolist(), b.tolist(), r.tolist())) == sorted(want + [(1, -1, -1), (5, -1, -1)])
    j, (l, b, r, c) = run(O.JOIN_LEFT, O.STRICT_ANY)
    assert list(zip(l.tolist(), b.tolist(), r.tolist())) == [(0, 0, 0), (1, -1, -1), (2, 0, 3), (3, 0, 1), (4, 0, 1), (5, -1, -1), (6, 0, 4)]
    j, (l, b, r, c) = run(O.JOIN_LEFT, O.STRICT_ANY, any_take_last_row=True)
    assert list(zip(l.tolist(), b.tolist(), r.tolist())) == [(0, 0, 5), (1, -1, -1), (2, 1, 2), (3, 1, 0), (4, 1, 0), (5, -1, -1), (6, 0, 4)]
    j, (l, b, r, c) = run(O.JOIN_INNER, O.STRICT_ANY)  # each right key joins its first left row only
    assert list(zip(l.tolist(), b.tolist(), r.tolist())) == [(0, 0, 0), (2, 0, 3), (3, 0, 1), (6, 0, 4)]
    j, (l, b, r, c) = run(O.JOIN_LEFT, O.STRICT_SEMI)
    assert list(zip(l.tolist(), b.tolist(), r.tolist())) == [(0, 0, 0), (2, 0, 3), (3, 0, 1), (4, 0, 1), (6, 0, 4)]
    j, (l, b, r, c) = run(O.JOIN_LEFT, O.STRICT_ANTI)
    assert list(zip(l.tolist(), b.tolist(), r.tolist())) == [(1, -1, -1), (5, -1, -1)]
    # max_joined_block_rows: stop BEFORE row i once current_offset >= max (HashJoinMethodsImpl.h:436-444)
    j = O.HashJoin(O.JOIN_INNER, O.STRICT_ALL)
    j.add_block(right1)
    j.add_block(right2)
    l, b, r, c = j.joined_pairs(left, max_joined_block_rows=4)
    assert c == 3 and l.shape[0] == 5  # rows 0..2 consumed: 3 + 0 + 2 matches
    # null keys are neither inserted nor matched
    j = O.HashJoin(O.JOIN_INNER, O.STRICT_ALL)
    j.add_block(right1, null_map=np.array([1, 0, 0, 0, 0, 0], dtype=np.uint8))
    l, b, r, c = j.joined_pairs(left, null_map=np.array([0, 0, 1, 0, 0, 0, 0], dtype=np.uint8))
    assert sorted(zip(l.tolist(), r.tolist())) == [(0, 2), (0, 5), (3, 1), (4, 1), (6, 4)]


def test_rowreflist_batch_order(oracle_mod):
    # 1 root + 7-slot batches: order = root, newest batch 0..size-1, older batches (RowRefs.h:33-108,129-138)
    O = oracle_mod
    j = O.HashJoin(O.JOIN_INNER, O.STRICT_ALL)
    j.add_block(np.full(17, 3, dtype=np.uint64))
    l, b, r, c = j.joined_pairs(np.array([3], dtype=np.uint64))
    assert r.tolist() == [0, 15, 16, 8, 9, 10, 11, 12, 13, 14, 1, 2, 3, 4, 5, 6, 7]


def test_selector_and_scatter_index_replicate(oracle_mod):
    O = oracle_mod
    rng = np.random.Generator(np.random.PCG64(5))
    keys = rng.integers(0, 2**64, size=10000, dtype=np.uint64)
    for shards in (1, 2, 8, 256):
        sel = O.hash_to_selector(keys, shards)
        want = ((O.hash_crc32(keys) >> np.uint64(24)) & np.uint64(0xFF)) & np.uint64(shards - 1)
        assert np.array_equal(sel, want)
    sel = O.hash_to_selector(keys, 8)
    parts = O.scatter(keys, sel, 8)
    for s in range(8):
        assert np.array_equal(parts[s], keys[sel == s])
    k32 = keys.astype(np.uint32)
    assert np.array_equal(O.hash_to_selector(k32, 8), O.hash_to_selector(k32.astype(np.uint64), 8))
    idx = rng.integers(0, 10000, size=5000, dtype=np.uint64)
    assert np.array_equal(O.index_column(keys, idx), keys[idx])
    cnt = rng.integers(0, 4, size=1000)
    off = np.cumsum(cnt).astype(np.uint64)
    assert np.array_equal(O.replicate(keys[:1000], off), np.repeat(keys[:1000], cnt))
    wh = O.weak_hash32(keys)
    assert np.array_equal(wh.astype(np.uint64), O.hash_crc32(keys))
