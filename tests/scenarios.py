"""Restatements (as input builders) of the reference's stateless SQL tests that pin the hot path's semantics.

Each scenario takes an `engine` exposing the mirror API shared by the CPU oracle (oracle.HashJoin / oracle.Aggregator)
and the GPU host classes (clickhouse_amd.HashJoin / clickhouse_amd.Aggregator), and returns rows formatted like the
reference's .reference files so they can be diffed against tests/golden/sql_reference_rows.json.
"""
import numpy as np

I64, U32, U64, F64, U8, I32 = 0, 1, 2, 3, 4, 5
AGG_COUNT, AGG_SUM, AGG_AVG, AGG_MIN, AGG_MAX, AGG_ANY = 0, 1, 2, 3, 4, 5
JOIN_INNER, JOIN_LEFT = 0, 1
STRICT_ANY, STRICT_ALL, STRICT_SEMI, STRICT_ANTI = 0, 1, 2, 3


def numbers(n, dtype=np.uint64):
    return np.arange(n, dtype=dtype)


def _gather_default(col, rrow):
    """Right payload by row with type default 0 for missing rows (addNotFoundRow -> insertDefault)."""
    out = np.zeros(rrow.shape[0], dtype=col.dtype)
    ok = rrow >= 0
    out[ok] = col[rrow[ok]]
    return out


def _fmt(*cols):
    return [[str(int(v)) for v in row] for row in zip(*cols)]


def q00049(engine):
    # numbers ANY LEFT JOIN (number*2 AS number, number*10+1 AS joined LIMIT 10) USING number LIMIT 10
    n = numbers(10)
    j = engine.HashJoin(JOIN_LEFT, STRICT_ANY)
    j.add_block(n * 2)
    joined = (n * 2) * 10 + 1  # `number` in the 2nd expression resolves to the alias number*2 (expected row 2 -> 21)
    left, _, rrow, _ = j.joined_pairs(n)
    rows = _fmt(n[left], _gather_default(joined, rrow))
    return sorted(rows, key=lambda r: [int(x) for x in r])


def q00050(engine):
    n = numbers(10)
    j = engine.HashJoin(JOIN_LEFT, STRICT_ANY)
    j.add_block(n * 2)
    left, _, rrow, _ = j.joined_pairs(n)
    bk = _gather_default(n * 2, rrow)
    rows = _fmt(n[left], bk, _gather_default(n, rrow))
    return sorted(rows, key=lambda r: int(r[0]))


def q00051(engine):
    n = numbers(10)
    j = engine.HashJoin(JOIN_INNER, STRICT_ANY)
    j.add_block(n * 2)
    left, _, rrow, _ = j.joined_pairs(n)
    rows = _fmt(n[left], _gather_default(n * 2, rrow), _gather_default(n, rrow))
    return sorted(rows, key=lambda r: [int(x) for x in r])


def q00052(engine):
    n = numbers(10)
    j = engine.HashJoin(JOIN_LEFT, STRICT_ALL)
    j.add_block(n // 2)
    left, _, rrow, _ = j.joined_pairs(n)
    rows = _fmt(n[left], _gather_default(n, rrow))
    return sorted(rows, key=lambda r: [int(x) for x in r])


def q00053(engine):
    n = numbers(10)
    j = engine.HashJoin(JOIN_INNER, STRICT_ALL)
    j.add_block(n // 2)
    left, _, rrow, _ = j.joined_pairs(n)
    rows = _fmt(n[left], _gather_default(n // 2, rrow), _gather_default(n, rrow))
    return sorted(rows, key=lambda r: (int(r[0]), int(r[2])))


def q00055(engine):
    # two UInt8 keys packed into one fixed-width key (keys16-style packing); equality semantics identical
    n = numbers(10)
    lk = (n % 4) | ((n % 3) << 8)
    rk = (n % 2) | ((n % 6) << 8)
    j = engine.HashJoin(JOIN_LEFT, STRICT_ALL)
    j.add_block(rk)
    left, _, rrow, _ = j.joined_pairs(lk)
    rows = _fmt(n[left], _gather_default(n, rrow))
    return sorted(rows, key=lambda r: [int(x) for x in r])


def q00120(engine, sql_intHash64, sql_intHash32):
    # (number, intHash64(number)) ANY LEFT JOIN (number, intHash32(number)) USING number
    # GROUP BY value1, value2 -> sum(number)
    n = numbers(10)
    v1 = np.array([sql_intHash64(int(x)) for x in n], dtype=np.uint64)
    v2 = np.array([sql_intHash32(int(x)) for x in n], dtype=np.uint32)
    j = engine.HashJoin(JOIN_LEFT, STRICT_ANY)
    j.add_block(n)
    left, _, rrow, _ = j.joined_pairs(n)
    value1 = v1[left]
    value2 = _gather_default(v2, rrow)
    # group by (value1, value2): value1 is already unique per row; use it as the 64-bit key and carry value2
    a = engine.Aggregator(np.uint64, [(AGG_SUM, np.uint64)], two_level_threshold=100000)
    a.execute_on_block(value1, [n[left]])
    keys, (sums,) = a.convert_to_block()
    v2_of = {int(k): int(v) for k, v in zip(value1, value2)}
    rows = [[str(int(k)), str(v2_of[int(k)]), str(int(s))] for k, s in zip(keys, sums)]
    return sorted(rows, key=lambda r: (int(r[0]), int(r[1])))


def q00041(engine):
    # SELECT number, count() FROM numbers LIMIT 200000 GROUP BY number ORDER BY count(), number LIMIT 10
    n = numbers(200000)
    a = engine.Aggregator(np.uint64, [(AGG_COUNT, None)])
    bs = 65409
    for b in range(0, n.shape[0], bs):
        a.execute_on_block(n[b:b + bs], [None])
    keys, (cnt,) = a.convert_to_block()
    assert keys.shape[0] == 200000
    order = np.lexsort((keys, cnt))[:10]
    return _fmt(keys[order], cnt[order])


def q00266(engine):
    n = numbers(110000)
    a = engine.Aggregator(np.uint64, [(AGG_COUNT, None)])
    bs = 65409
    for b in range(0, n.shape[0], bs):
        a.execute_on_block(n[b:b + bs], [None])
    keys, _ = a.convert_to_block()
    assert keys.shape[0] == 110000
    return [[str(int(k))] for k in np.sort(keys)[:10]]


def q02144(engine):
    # avg(-8000000000000000000) over numbers(65535*2): (a) GROUP BY constant key 1, (b) without key
    rows = 65535 * 2
    v = np.full(rows, -8000000000000000000, dtype=np.int64)
    out = []
    a = engine.Aggregator(np.uint32, [(AGG_AVG, np.int64)])
    bs = 65409
    k = np.ones(rows, dtype=np.uint32)
    for b in range(0, rows, bs):
        a.execute_on_block(k[b:b + bs], [v[b:b + bs]])
    _, (avg,) = a.convert_to_block()
    out.append(float(avg[0]))
    a = engine.Aggregator(None, [(AGG_AVG, np.int64)])
    for b in range(0, rows, bs):
        a.execute_on_block(None, [v[b:b + bs]])
    _, (avg,) = a.convert_to_block()
    out.append(float(avg[0]))
    return out


def q01091(engine):
    n = numbers(1000000)
    a = engine.Aggregator(None, [(AGG_SUM, np.uint64)])
    bs = 65409
    for b in range(0, n.shape[0], bs):
        a.execute_on_block(None, [n[b:b + bs]])
    _, (s,) = a.convert_to_block()
    return [[str(int(s[0]))]]


def q01300_max(engine, rows=10000000, block=65505):
    # round(max(log(2) * number), 6) FROM numbers(1e7) GROUP BY number % 2, number % 3, (number % 2 + number % 3) % 2  ORDER BY k
    # (the third key is a function of the first two: six groups; the keys are packed the way keys16 would be)
    n = numbers(rows)
    val = np.log(2.0) * n.astype(np.float64)
    key = ((n % 2) | ((n % 3) << np.uint64(8))).astype(np.uint32)
    a = engine.Aggregator(np.uint32, [(AGG_MAX, np.float64)])
    for b in range(0, rows, block):
        a.execute_on_block(key[b:b + block], [val[b:b + block]])
    _, (mx,) = a.convert_to_block()
    return sorted(round(float(x), 6) for x in mx)


def q01321_min_max(engine, rows=10000000, block=65409):
    # SELECT min(number % 2) AS a, max(number % 3) AS b FROM numbers(1e7) GROUP BY number % 2, number % 3 ORDER BY a, b
    n = numbers(rows)
    k2, k3 = (n % 2).astype(np.uint8), (n % 3).astype(np.uint8)
    key = (k2.astype(np.uint32) | (k3.astype(np.uint32) << 8))
    a = engine.Aggregator(np.uint32, [(AGG_MIN, np.uint8), (AGG_MAX, np.uint8)])
    for b in range(0, rows, block):
        a.execute_on_block(key[b:b + block], [k2[b:b + block], k3[b:b + block]])
    _, (mn, mx) = a.convert_to_block()
    return sorted([str(int(x)), str(int(y))] for x, y in zip(mn, mx))


def q01321_any(engine, rows=10000000, block=65409):
    # SELECT any(number % 2) AS a, anyLast(number % 3) AS b FROM numbers(1e7) GROUP BY number % 2, number % 3 ORDER BY a, b
    # (both arguments are the group's own key: first = last, so anyLast is answered by any)
    n = numbers(rows)
    k2, k3 = (n % 2).astype(np.uint8), (n % 3).astype(np.uint8)
    key = (k2.astype(np.uint32) | (k3.astype(np.uint32) << 8))
    a = engine.Aggregator(np.uint32, [(AGG_ANY, np.uint8), (AGG_ANY, np.uint8)])
    for b in range(0, rows, block):
        a.execute_on_block(key[b:b + block], [k2[b:b + block], k3[b:b + block]])
    _, (x, y) = a.convert_to_block()
    return sorted([str(int(p)), str(int(q))] for p, q in zip(x, y))


def q01321_max_product(engine, rows=10000000, block=65409):
    # SELECT max((number % 5) * (number % 7)) AS a FROM numbers(1e7) GROUP BY number % 7, number % 5 ORDER BY a
    n = numbers(rows)
    k5, k7 = (n % 5), (n % 7)
    prod = (k5 * k7).astype(np.uint16)   # UInt8 * UInt8 -> UInt16 (NumberTraits)
    key = (k7 | (k5 << np.uint64(8))).astype(np.uint32)
    a = engine.Aggregator(np.uint32, [(AGG_MAX, np.uint16)])
    for b in range(0, rows, block):
        a.execute_on_block(key[b:b + block], [prod[b:b + block]])
    _, (mx,) = a.convert_to_block()
    return sorted(int(x) for x in mx)


def q01300(engine, rows=10000000, block=65505):
    # round(avg(log(2) * number), 6) FROM numbers(1e7) GROUP BY number % 5  ORDER BY k  (max_block_size = 65505)
    n = numbers(rows)
    val = np.log(2.0) * n.astype(np.float64)
    key = (n % 5).astype(np.uint32)  # UInt8 in the reference; widened key, same grouping
    a = engine.Aggregator(np.uint32, [(AGG_AVG, np.float64)])
    for b in range(0, rows, block):
        a.execute_on_block(key[b:b + block], [val[b:b + block]])
    _, (avg,) = a.convert_to_block()
    return sorted(float(x) for x in avg)


# ---- ASOF joins: the inputs of the reference's 00927 tests and the rows they print ----------------------------------------------------------
def _utc(t):
    return "1970-01-01 00:00:%02d" % int(t)


def _num(x):
    x = float(x)
    return str(int(x)) if x == int(x) else repr(x)


def asof_noninclusive(join_pairs):
    """00927_asof_join_noninclusive: A(k, t, a) ASOF LEFT / INNER JOIN B(k, t, b) on k, A.t >= B.t.  join_pairs(build_blocks, left_k, left_t,
    left_join) -> [(left_row, block, row)] ((-1, -1) = default row).  -> the three result sets in the file's order, ORDER BY (A.k, A.t)"""
    ak = np.repeat(np.array([1, 2, 3], dtype=np.uint32), 5)
    at = np.tile(np.arange(1, 6, dtype=np.uint32), 3)
    aa = at.astype(np.float64)
    b_blocks = [(np.array([1, 1], dtype=np.uint32), np.array([2, 4], dtype=np.uint32), np.array([2.0, 4.0])),       # the two INSERTs into B
                (np.array([2], dtype=np.uint32), np.array([3], dtype=np.uint32), np.array([3.0]))]
    build = [(k, t, None, None) for k, t, _ in b_blocks]
    out = []
    for left_join in (True, False, False):
        rows = []
        for i, blk, r in join_pairs(build, ak, at, left_join):
            bk, bt, bb = (0, 0, 0.0) if blk < 0 else (b_blocks[blk][0][r], b_blocks[blk][1][r], b_blocks[blk][2][r])
            rows.append([str(int(ak[i])), _utc(at[i]), _num(aa[i]), _num(bb), _utc(bt), str(int(bk))])
        out += rows                                                  # (A is already in (k, t) order)
    return out


def asof_joins_left(join_pairs):
    """00927_asof_joins: tv ASOF LEFT JOIN md USING(key, t): md inserted out of time order"""
    md = [(np.array([1, 1, 1, 1], dtype=np.uint32), np.array([20, 5, 10, 15], dtype=np.uint32), np.array([7.0, 1, 11, 5]), np.array([8.0, 2, 12, 6])),
          (np.array([2, 2, 2, 2], dtype=np.uint32), np.array([20, 5, 10, 15], dtype=np.uint32), np.array([17.0, 11, 21, 5]), np.array([18.0, 12, 22, 6]))]
    tk = np.repeat(np.array([1, 2], dtype=np.uint32), 7)
    tt = np.tile(np.array([5, 6, 10, 11, 15, 16, 20], dtype=np.uint32), 2)
    tv = np.array([1.5, 1.51, 11.5, 11.51, 5.5, 5.6, 7.5, 2.5, 2.51, 12.5, 12.51, 6.5, 5.6, 8.5])
    rows = []
    for i, blk, r in join_pairs([(k, t, None, None) for k, t, _, _ in md], tk, tt, True):
        bid, ask = (0.0, 0.0) if blk < 0 else (md[blk][2][r], md[blk][3][r])
        rows.append([str(int(tk[i])), _utc(tt[i]), _num(bid), _num(tv[i]), _num(ask)])
    return rows


def asof_join_long(join_pairs, keys=1000):
    """00927_asof_join_long: tvs(k, t = 3 * number, tv = t) for 1000 keys x 10000 times; trades(k, t = 10 * number, price = t) for 1000 keys x
    3000 times; SELECT SUM(trades.price - tvs.tv) FROM trades ASOF LEFT JOIN tvs USING(k, t)"""
    bk = np.repeat(np.arange(keys, dtype=np.uint32), 10000)
    bt = np.tile((np.arange(10000, dtype=np.uint32) * 3), keys)
    lk = np.repeat(np.arange(keys, dtype=np.uint32), 3000)
    lt = np.tile((np.arange(3000, dtype=np.uint32) * 10), keys)
    total = 0
    pairs = join_pairs([(bk, bt, None, None)], lk, lt, True)
    idx = np.array([(i, r) for i, blk, r in pairs], dtype=np.int64) if not isinstance(pairs, tuple) else None
    if idx is None:                                                  # (left rows, right rows) arrays straight from the device wrapper
        li, ri = pairs
    else:
        li, ri = idx[:, 0], idx[:, 1]
    tvv = np.where(ri >= 0, bt[np.maximum(ri, 0)].astype(np.int64), 0)
    total = int((lt[li].astype(np.int64) - tvv).sum())
    return [[str(total)]]
