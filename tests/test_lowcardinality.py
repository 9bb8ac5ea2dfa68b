"""GROUP BY over LowCardinality(String) keys — per-Block dictionaries resolved against a query-wide
dictionary on the host, rows translated on the device (k_lc_remap), then the ordinary UInt32 GROUP BY."""
import numpy as np
import pytest

from oracle import lowcardinality as OL


def _blocks(rng, n_blocks, rows, n_values, index_dtype):
    """Blocks whose dictionaries are different subsets / orders of one value universe"""
    universe = [f"value-{i:06d}" for i in range(n_values)]
    out = []
    for b in range(n_blocks):
        size = int(rng.integers(max(1, n_values // 2), n_values + 1))
        d = [universe[i] for i in rng.permutation(n_values)[:size]]
        idx = rng.integers(0, size, size=rows).astype(index_dtype)
        vals = rng.integers(-2**40, 2**40, size=rows, dtype=np.int64)
        out.append((d, idx, vals))
    return out


def test_oracle_groups_by_value_across_block_dictionaries():
    b1 = (["ASIA", "EUROPE", "AFRICA"], np.array([0, 1, 1, 2], dtype=np.uint8), np.array([1, 10, 100, 1000], dtype=np.int64))
    b2 = (["EUROPE", "AMERICA", "ASIA"], np.array([0, 2, 1, 2], dtype=np.uint8), np.array([5, 7, -3, 2**63 - 1], dtype=np.int64))
    got = OL.group_by_sum_count([b1, b2])
    assert got == {"ASIA": (1 + 7 + 2**63 - 1 - 2**64, 3), "EUROPE": (115, 3), "AFRICA": (1000, 1), "AMERICA": (-3, 1)}
    assert OL.convert_to_full(b1[0], b1[1]) == ["ASIA", "EUROPE", "EUROPE", "AFRICA"]


@pytest.mark.gpu
@pytest.mark.parametrize("index_dtype,n_values,rows", [(np.uint8, 200, 100_003), (np.uint16, 5000, 300_001), (np.uint32, 40_000, 250_000),
                                                         (np.uint64, 300, 70_001), (np.uint16, 25, 1_000_003)])
def test_gpu_lowcardinality_group_by_matches_oracle(index_dtype, n_values, rows):
    import clickhouse_amd as ch
    from clickhouse_amd.lowcardinality import ColumnLowCardinality, LowCardinalityAggregator
    rng = np.random.Generator(np.random.PCG64(rows))
    ctx = ch.Context()
    blocks = _blocks(rng, 4, rows, n_values, index_dtype)
    agg = LowCardinalityAggregator([(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], ctx=ctx)
    for d, idx, vals in blocks:
        agg.execute_on_block(ColumnLowCardinality(d, ctx.upload(idx)), [ctx.upload(vals), None])
    keys, (sums, cnts) = agg.convert_to_block()
    want = OL.group_by_sum_count(blocks)
    assert len(keys) == len(set(keys)) == len(want) == len(agg)
    assert {k: (int(s), int(c)) for k, s, c in zip(keys, sums, cnts)} == want


@pytest.mark.gpu
def test_gpu_lc_remap_views_tails_and_big_dictionaries():
    import clickhouse_amd as ch
    from clickhouse_amd.lowcardinality import ColumnLowCardinality, LowCardinalityDictionary
    rng = np.random.Generator(np.random.PCG64(99))
    ctx = ch.Context()
    for index_dtype, dict_size in [(np.uint8, 256), (np.uint16, 65_536), (np.uint32, 100_000), (np.uint32, 7), (np.uint64, 33_000)]:
        d = [int(x) for x in rng.permutation(dict_size)]  # numeric dictionary values: the remap is a permutation
        gd = LowCardinalityDictionary(ctx)
        for n in (0, 1, 15, 16, 17, 4097, 200_003):
            idx = rng.integers(0, dict_size, size=n + 3).astype(index_dtype)
            col = ctx.upload(idx)
            for start in (0, 1, 3):  # unaligned views
                got = gd.map_block(ColumnLowCardinality(d, col.cut(start, n))).numpy()
                ids = np.array([gd._ids[v] for v in d], dtype=np.uint32)
                assert got.dtype == np.uint32 and np.array_equal(got, ids[idx[start:start + n].astype(np.int64)])
        assert len(gd) == dict_size


@pytest.mark.gpu
def test_gpu_lowcardinality_with_where_mask_and_filtered_column():
    import clickhouse_amd as ch
    from clickhouse_amd.lowcardinality import ColumnLowCardinality, LowCardinalityAggregator
    rng = np.random.Generator(np.random.PCG64(5))
    ctx = ch.Context()
    n = 400_001
    nations = ["ALGERIA", "ARGENTINA", "BRAZIL", "CANADA", "EGYPT", "ETHIOPIA", "FRANCE", "GERMANY", "INDIA", "INDONESIA"]
    lc = ColumnLowCardinality.from_values(ctx, np.array(nations)[rng.integers(0, 10, size=n)])
    full = np.array(lc.convert_to_full_column())
    v = rng.integers(0, 1000, size=n, dtype=np.int64)
    mask = (v % 3 == 0).astype(np.uint8)
    # (a) WHERE fused into the aggregation  (b) ColumnLowCardinality::filter first
    a = LowCardinalityAggregator([(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], ctx=ctx)
    a.execute_on_block(lc, [ctx.upload(v), None], filter=ctx.upload(mask))
    b = LowCardinalityAggregator([(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], ctx=ctx)
    m = ctx.upload(mask)
    b.execute_on_block(lc.filter(m), [ctx.upload(v).filter(m), None])
    want = {k: (int(v[(full == k) & (mask != 0)].sum()), int(((full == k) & (mask != 0)).sum())) for k in nations}
    for agg in (a, b):
        keys, (sums, cnts) = agg.convert_to_block()
        assert {k: (int(s), int(c)) for k, s, c in zip(keys, sums, cnts)} == want


def _strings(rng, n, n_distinct, max_len):
    pool = []
    for k in range(n_distinct):
        ln = int(rng.integers(0, max_len + 1))
        b = rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes() if k % 3 else bytes(rng.integers(97, 100, size=ln, dtype=np.uint8))  # zero bytes inside; tiny alphabets
        pool.append(b)
    pool = list(dict.fromkeys(pool + [b"", b"a", b"ab", b"ab\0", b"abc", b"x" * 64, b"x" * 65, b"x" * 63 + b"y"]))
    return [pool[int(i)] for i in rng.integers(0, len(pool), size=n)]


def test_oracle_dictionary_encode_first_appearance_order():
    ids, d, first = OL.dictionary_encode([b"b", b"a", b"b", b"", b"a", b"c"])
    assert ids.tolist() == [0, 1, 0, 2, 1, 3] and d == [b"b", b"a", b"", b"c"] and first.tolist() == [0, 1, 3, 5]


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_distinct,max_len", [(0, 1, 4), (1, 1, 8), (1000, 3, 5), (100_003, 500, 40), (200_001, 150_000, 12), (50_000, 50, 300)])
def test_gpu_string_dictionary_encode_matches_oracle(n, n_distinct, max_len):
    import clickhouse_amd as ch
    rng = np.random.Generator(np.random.PCG64(n + 17))
    ctx = ch.Context()
    vals = _strings(rng, n, n_distinct, max_len)
    lc = ch.ColumnString.from_values(ctx, vals).dictionary_encode()
    ids, d, _ = OL.dictionary_encode(vals)
    assert lc.dictionary == d and np.array_equal(lc.indexes.numpy(), ids)
    # without the host copy of the Block: dictionary strings read back from the device column
    cs = ch.ColumnString.from_values(ctx, vals)
    cs._host_values = None
    assert cs.dictionary_encode().dictionary == d


@pytest.mark.gpu
def test_gpu_group_by_string_key_over_stripes():
    """GROUP BY a String key: each stripe dictionary-encoded on the device, ids unified by LowCardinalityDictionary"""
    import clickhouse_amd as ch
    rng = np.random.Generator(np.random.PCG64(23))
    ctx = ch.Context()
    cities = [f"CITY-{i:04d}".encode() for i in range(2500)]
    agg = ch.LowCardinalityAggregator([(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], ctx=ctx)
    blocks = []
    for stripe in range(3):
        n = 150_000 + stripe
        vals = [cities[int(i)] for i in rng.integers(stripe * 500, 1500 + stripe * 500, size=n)]
        v = rng.integers(-10**9, 10**9, size=n, dtype=np.int64)
        lc = ch.ColumnString.from_values(ctx, vals).dictionary_encode()
        agg.execute_on_block(lc, [ctx.upload(v), None])
        ids, d, _ = OL.dictionary_encode(vals)
        blocks.append((d, ids, v))
    keys, (sums, cnts) = agg.convert_to_block()
    assert {k: (int(s), int(c)) for k, s, c in zip(keys, sums, cnts)} == OL.group_by_sum_count(blocks)


@pytest.mark.gpu
def test_gpu_ssb_q31_shape_group_by_two_string_keys_and_year():
    """SELECT c_nation, s_nation, d_year, sum(lo_revenue) ... GROUP BY c_nation, s_nation, d_year with real String keys:
    each stripe's String columns dictionary-encoded on the device, ids unified per key, 2 + 2 + 2 key bytes packed"""
    import clickhouse_amd as ch
    rng = np.random.Generator(np.random.PCG64(31))
    ctx = ch.Context()
    nations = [n.encode() for n in ("CHINA", "INDIA", "INDONESIA", "JAPAN", "VIETNAM", "FRANCE", "GERMANY", "PERU", "BRAZIL", "UNITED KINGDOM")]
    agg = ch.PackedKeysAggregator(["lc", "lc", np.uint16], [(ch.AGG_SUM, np.uint32), (ch.AGG_COUNT, None)], ctx=ctx)
    want = {}
    for stripe in range(3):
        n = 120_000 + 7 * stripe
        c = rng.integers(0, 10, size=n)
        s = rng.integers(0, 10, size=n)
        y = rng.integers(1992, 1999, size=n).astype(np.uint16)
        rev = rng.integers(1, 10_000_000, size=n).astype(np.uint32)
        order = rng.permutation(10)  # every stripe meets the values in another order: different local dictionaries
        cs = ch.ColumnString.from_values(ctx, [nations[order[i]] for i in c]).dictionary_encode()
        ss = ch.ColumnString.from_values(ctx, [nations[order[i]] for i in s]).dictionary_encode()
        agg.execute_on_block([cs, ss, ctx.upload(y)], [ctx.upload(rev), None])
        for cn, sn, yy, r in zip(order[c], order[s], y, rev):
            k = (nations[cn], nations[sn], int(yy))
            a = want.get(k, (0, 0))
            want[k] = (a[0] + int(r), a[1] + 1)
    (kc, ks, ky), (sums, cnts) = agg.convert_to_block()
    got = {(a, b, int(yv)): (int(sv), int(cv)) for a, b, yv, sv, cv in zip(kc, ks, ky, sums, cnts)}
    assert len(got) == len(kc) == len(agg) and got == want
    with pytest.raises(ch.ChgpuError) as ei:
        ch.PackedKeysAggregator(["lc", np.uint64], [(ch.AGG_COUNT, None)], ctx=ctx)  # 10 key bytes
    assert ei.value.code == ch._capi.ERR_NOT_IMPLEMENTED


def test_oracle_string_filter():
    vals = [b"ab", b"", b"xyz", b"q\0r"]
    chars = np.frombuffer(b"".join(v + b"\0" for v in vals), dtype=np.uint8)
    offs = np.cumsum([len(v) + 1 for v in vals]).astype(np.uint64)
    o, c = OL.string_filter(offs, chars, np.array([0, 1, 0, 7], dtype=np.uint8))
    assert o.tolist() == [1, 5] and c.tobytes() == b"\0q\0r\0"
    with pytest.raises(ValueError):
        OL.string_filter(offs, chars, np.zeros(3, dtype=np.uint8))


@pytest.mark.gpu
def test_gpu_string_filter_matches_oracle():
    import clickhouse_amd as ch
    rng = np.random.Generator(np.random.PCG64(41))
    ctx = ch.Context()
    for n, max_len, keep in [(1, 5, 1.0), (1000, 0, 0.5), (50_003, 40, 0.1), (70_001, 300, 0.9), (20_000, 12, 0.0), (20_000, 12, 1.0)]:
        vals = _strings(rng, n, max(1, n // 3), max_len)
        cs = ch.ColumnString.from_values(ctx, vals)
        mask = (rng.random(n) < keep).astype(np.uint8) * rng.integers(1, 256, size=n).astype(np.uint8)
        got = cs.filter(ctx.upload(mask))
        eo, ec = OL.string_filter(cs.offsets.numpy(), cs.chars.numpy(), mask)
        assert np.array_equal(got.offsets.numpy(), eo) and np.array_equal(got.chars.numpy(), ec), (n, max_len, keep)
        assert got.to_list() == [v for v, m in zip(vals, mask) if m]
    with pytest.raises(ch.ChgpuError) as ei:
        cs.filter(ctx.upload(np.ones(3, dtype=np.uint8)))
    assert ei.value.code == ch._capi.ERR_SIZES_MISMATCH
    # WHERE on a numeric column, String column filtered by the same mask, then GROUP BY the String key
    n = 100_000
    names = [f"BRAND#{i % 40}".encode() for i in rng.integers(0, 1000, size=n)]
    qty = rng.integers(0, 50, size=n).astype(np.int64)
    m = ch.cmp_const(ctx.upload(qty), ch.LT, 25)
    fs = ch.ColumnString.from_values(ctx, names).filter(m)
    fq = ctx.upload(qty).filter(m)
    agg = ch.LowCardinalityAggregator([(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], ctx=ctx)
    agg.execute_on_block(fs.dictionary_encode(), [fq, None])
    keys, (sums, cnts) = agg.convert_to_block()
    want = {}
    for nm, q in zip(names, qty):
        if q < 25:
            a = want.get(nm, (0, 0))
            want[nm] = (a[0] + int(q), a[1] + 1)
    assert {k: (int(s), int(c)) for k, s, c in zip(keys, sums, cnts)} == want


@pytest.mark.gpu
def test_gpu_group_by_nullable_key():
    """GROUP BY a Nullable(UInt32) key: the NULL group is kept out of the table (AggregationDataWithNullKey); nested values under a
    set null-map byte are ignored"""
    import clickhouse_amd as ch
    rng = np.random.Generator(np.random.PCG64(12))
    ctx = ch.Context()
    agg = ch.NullableKeyAggregator(np.uint32, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None), (ch.AGG_AVG, np.float64)], ctx=ctx)
    want, null_state = {}, [0, 0, 0.0]
    for block in range(3):
        n = 100_000 + block
        k = rng.integers(0, 500, size=n).astype(np.uint32)
        nm = (rng.random(n) < (0.0 if block == 0 else 0.2)).astype(np.uint8)
        k[nm != 0] = rng.integers(0, 2**32, size=int(nm.sum()), dtype=np.uint32)  # garbage under the null map
        v = rng.integers(-1000, 1000, size=n, dtype=np.int64)
        f = rng.random(n)
        agg.execute_on_block(k, nm, [v, None, f])
        for kk in np.unique(k[nm == 0]):
            m = (k == kk) & (nm == 0)
            a = want.get(int(kk), [0, 0, 0.0])
            want[int(kk)] = [a[0] + int(v[m].sum()), a[1] + int(m.sum()), a[2] + float(f[m].sum())]
        null_state = [null_state[0] + int(v[nm != 0].sum()), null_state[1] + int((nm != 0).sum()), null_state[2] + float(f[nm != 0].sum())]
    keys, nulls, (s, c, a) = agg.convert_to_block()
    assert len(agg) == len(want) + 1 == keys.shape[0] and nulls[-1] == 1 and int(nulls[:-1].sum()) == 0
    for kk, ss, cc, aa in zip(keys[:-1].tolist(), s[:-1].tolist(), c[:-1].tolist(), a[:-1].tolist()):
        w = want[kk]
        assert (ss, cc) == (w[0], w[1]) and abs(aa - w[2] / w[1]) <= 1e-6 * abs(w[2] / w[1])
    assert (int(s[-1]), int(c[-1])) == (null_state[0], null_state[1]) and abs(a[-1] - null_state[2] / null_state[1]) <= 1e-9


@pytest.mark.gpu
def test_gpu_join_on_string_keys_through_a_shared_dictionary():
    """INNER ALL JOIN ON l.name = r.name with String keys: both sides dictionary-encoded on the device and mapped through ONE
    query-wide dictionary, then the ordinary UInt32 hash join"""
    import clickhouse_amd as ch
    rng = np.random.Generator(np.random.PCG64(4))
    ctx = ch.Context()
    names = [f"SUPPLIER#{i:05d}".encode() for i in range(3000)]
    right = [names[int(i)] for i in rng.integers(0, 2000, size=5000)]     # duplicates on the build side
    left = [names[int(i)] for i in rng.integers(1000, 3000, size=20_000)]  # half of them have no partner
    d = ch.LowCardinalityDictionary(ctx)
    rk = d.map_block(ch.ColumnString.from_values(ctx, right).dictionary_encode())
    lk = d.map_block(ch.ColumnString.from_values(ctx, left).dictionary_encode())
    j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, key_dtype=np.uint32, ctx=ctx)
    j.add_block(rk)
    l, b, r, c = j.joined_pairs(lk)
    got = sorted(zip(l.tolist(), r.tolist()))
    pos = {}
    for i, v in enumerate(right):
        pos.setdefault(v, []).append(i)
    want = sorted((i, rr) for i, v in enumerate(left) for rr in pos.get(v, []))
    assert c == len(left) and got == want


def _string_rows():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "string_key_rows.json")) as f:
        return json.load(f)


def test_oracle_string_key_reference_rows():
    """00127_group_by_concat restated over the oracle: GROUP BY ('' as a String key, number % 123) over numbers(1000)"""
    want = [(r[0].encode(), int(r[1]), int(r[2])) for r in _string_rows()["00127_group_by_concat"]["rows"]]
    number = np.arange(1000, dtype=np.uint64)
    k2 = (number % 123).astype(np.uint8)
    got = OL.group_by_sum_count([([b""], np.zeros(1000, dtype=np.uint8) + 0 * k2, None)])  # one String group ...
    assert got == {b"": (0, 1000)}
    cnt = np.bincount(k2, minlength=123)                                                   # ... times the numeric key
    assert [(b"", int(k), int(c)) for k, c in enumerate(cnt)] == want


@pytest.mark.gpu
def test_gpu_string_key_reference_rows():
    """00054_join_string (ALL LEFT JOIN USING a String key, defaults for the misses), 00056_join_number_string (USING a number and a
    String key, packed) and 00127_group_by_concat (GROUP BY a String and number % 123, the modulo computed on the device) against the
    reference's expected rows"""
    import clickhouse_amd as ch
    ctx = ch.Context()
    rows = _string_rows()
    # ---- 00054: left k = 'A'..'J'; right k = char('A' + number div 2), joined = number; ALL LEFT JOIN USING k ORDER BY k, joined ----
    left = [bytes([65 + i]) for i in range(10)]
    right = [bytes([65 + i // 2]) for i in range(10)]
    joined = np.arange(10, dtype=np.uint64)
    d = ch.LowCardinalityDictionary(ctx)
    rk = d.map_block(ch.ColumnString.from_values(ctx, right).dictionary_encode())
    lk = d.map_block(ch.ColumnString.from_values(ctx, left).dictionary_encode())
    j = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_ALL, key_dtype=np.uint32, ctx=ctx)
    j.add_block(rk)
    r = j.probe_columns(lk)
    k_out = lk.replicate(r["offsets"])
    joined_out = ctx.upload(joined).index(r["right_rowid"], default_for_missing=True)
    got = sorted(zip(d.decode(k_out.numpy()), joined_out.numpy().tolist()))
    assert [(k.decode(), str(v)) for k, v in got] == [tuple(x) for x in rows["00054_join_string"]["rows"]]
    # ---- 00056: ALL LEFT JOIN USING (k1 = number % 4 | % 2, k2 = toString(number % 3 | % 6)): a number and a String key packed ----
    n10 = np.arange(10, dtype=np.uint64)
    d2 = ch.LowCardinalityDictionary(ctx)
    narrow = ch.ActionsDAG()
    narrow.add_function("toUInt16", narrow.add_input(0, np.uint32))
    narrow = narrow.compile()

    def packed(k1, strings):
        ids = d2.map_block(ch.ColumnString.from_values(ctx, strings).dictionary_encode())
        return ch.pack_fixed_keys([ctx.upload(k1.astype(np.uint8)), narrow.execute(ctx, [ids], [1])[0]])
    rkey = packed(n10 % 2, [str(int(x)).encode() for x in n10 % 6])
    lkey = packed(n10 % 4, [str(int(x)).encode() for x in n10 % 3])
    j2 = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_ALL, key_dtype=np.uint64, ctx=ctx)
    j2.add_block(rkey)
    r2 = j2.probe_columns(lkey)
    left_out = ctx.upload(n10).replicate(r2["offsets"])
    right_out = ctx.upload(n10).index(r2["right_rowid"], default_for_missing=True)
    got = sorted(zip(left_out.numpy().tolist(), right_out.numpy().tolist()))
    assert [[str(a), str(b)] for a, b in got] == rows["00056_join_number_string"]["rows"]
    # ---- 00127: GROUP BY materialize('') AS k1, number % 123 AS k2 over numbers(1000), count() ----
    dag = ch.ActionsDAG()
    k2n = dag.add_function("modulo", dag.add_input(0, np.uint64), dag.add_column(123, np.uint8))
    k2 = dag.compile().execute(ctx, [ctx.upload(np.arange(1000, dtype=np.uint64))], [k2n])[0]
    k1 = ch.ColumnString.from_values(ctx, [b""] * 1000).dictionary_encode()
    agg = ch.PackedKeysAggregator(["lc", np.uint8], [(ch.AGG_COUNT, None)], ctx=ctx)
    agg.execute_on_block([k1, k2], [None])
    (g1, g2), (cnt,) = agg.convert_to_block()
    got = sorted(zip(g1, g2.tolist(), cnt.tolist()))
    assert [[a.decode(), str(b), str(c)] for a, b, c in got] == rows["00127_group_by_concat"]["rows"]
