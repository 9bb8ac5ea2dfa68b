"""ordered output (IColumn::getPermutation, stable; sortBlock over several columns; LIMIT) and the
two-level (bucket_num) form of the partial GROUP BY states."""
import numpy as np
import pytest

from oracle import sorting as OS

TYPES = [np.int64, np.uint64, np.int32, np.uint32, np.int16, np.uint16, np.int8, np.uint8, np.float64, np.float32]


def _column(rng, dtype, n, few_values):
    dt = np.dtype(dtype)
    if dt.kind == "f":
        x = (rng.standard_normal(n) * 100).astype(dt)
        if few_values:
            x = np.round(x / 50).astype(dt)
        sp = np.array([np.nan, -np.nan, np.inf, -np.inf, 0.0, -0.0, 1.5, -1.5], dtype=dt)
        idx = rng.integers(0, n, size=max(1, n // 5))
        x[idx] = rng.choice(sp, size=idx.shape[0])
        return x
    info = np.iinfo(dt)
    if few_values:
        return rng.integers(max(info.min, -3), min(info.max, 4), size=n, dtype=dt)
    x = rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True)
    x[rng.integers(0, n, size=max(1, n // 10))] = info.min
    x[rng.integers(0, n, size=max(1, n // 10))] = info.max
    return x


def test_oracle_stable_permutation_semantics():
    x = np.array([2.0, np.nan, -0.0, 1.0, 0.0, np.nan, 2.0, -np.inf], dtype=np.float64)
    assert OS.get_permutation(x, False, 1).tolist() == [7, 2, 4, 3, 0, 6, 1, 5]    # NaN last, -0.0 == 0.0 keeps row order
    assert OS.get_permutation(x, False, -1).tolist() == [1, 5, 7, 2, 4, 3, 0, 6]   # NaN first
    assert OS.get_permutation(x, True, -1).tolist() == [0, 6, 3, 2, 4, 7, 1, 5]    # DESC, NaN last; ties still by row number
    assert OS.get_permutation(x, True, 1).tolist() == [1, 5, 0, 6, 3, 2, 4, 7]
    a = np.array([1, 0, 1, 0, 1], dtype=np.uint8)
    b = np.array([5, 7, 5, 9, 3], dtype=np.int32)
    assert OS.sort_block([(a, False, 1), (b, True, 1)]).tolist() == [3, 1, 0, 2, 4]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", TYPES)
def test_gpu_sort_permutation_equals_stable_oracle(dtype):
    import clickhouse_amd as ch
    ctx = ch.Context()
    rng = np.random.Generator(np.random.PCG64(np.dtype(dtype).itemsize * 131 + (np.dtype(dtype).kind == "f") * 7 + (np.dtype(dtype).kind == "i")))
    for n, few in [(0, False), (1, False), (2, True), (255, False), (256, True), (20_001, False), (50_003, True)]:
        x = _column(rng, dtype, max(n, 1), few)[:n]
        col = ctx.upload(x)
        for desc in (False, True):
            for hint in (1, -1):
                got = ch.sort_permutation(col, None, desc, hint).numpy()
                assert got.dtype == np.uint64 and np.array_equal(got, OS.get_permutation(x, desc, hint)), (dtype, n, desc, hint)


@pytest.mark.gpu
def test_gpu_sort_block_multi_column_and_limit():
    import clickhouse_amd as ch
    ctx = ch.Context()
    rng = np.random.Generator(np.random.PCG64(77))
    n = 60_007
    year = rng.integers(1992, 1999, size=n).astype(np.uint16)
    nation = rng.integers(0, 25, size=n).astype(np.uint8)
    profit = (rng.standard_normal(n) * 1e6).astype(np.float64)
    profit[rng.integers(0, n, size=50)] = np.nan
    cols = [ctx.upload(year), ctx.upload(nation), ctx.upload(profit)]
    desc = [(0, False, 1), (1, True, 1), (2, True, -1)]  # ORDER BY year ASC, nation DESC, profit DESC (NaN last)
    want = OS.sort_block([(year, False, 1), (nation, True, 1), (profit, True, -1)])
    out, perm = ch.sort_block(cols, desc)
    assert np.array_equal(perm.numpy(), want)
    w = want.astype(np.int64)
    assert np.array_equal(out[0].numpy(), year[w]) and np.array_equal(out[1].numpy(), nation[w])
    assert np.array_equal(out[2].numpy().view(np.uint64), profit[w].view(np.uint64))
    out10, perm10 = ch.sort_block(cols, desc, limit=10)  # ORDER BY ... LIMIT 10
    assert np.array_equal(perm10.numpy(), want[:10]) and out10[0].size() == 10


@pytest.mark.gpu
def test_gpu_sort_properties_at_2_pow_26_rows():
    """size-independent: the result is a permutation, the keys are ordered, equal keys keep their row order"""
    import clickhouse_amd as ch
    import torch
    ctx = ch.Context()
    n = 1 << 26
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randint(-2**20, 2**20, (n,), dtype=torch.int64, device="cuda", generator=g)
    col = ctx.wrap(x.data_ptr(), np.int64, n, x)
    for desc in (False, True):
        perm = ch.sort_permutation(col, None, desc, 1)
        p = torch.from_numpy(perm.numpy().astype(np.int64)).cuda()
        assert int(torch.bincount(p, minlength=n).max().item()) == 1 and int(p.min()) == 0 and int(p.max()) == n - 1
        s = x[p]
        d = s[1:] - s[:-1]
        assert bool(((d <= 0) if desc else (d >= 0)).all())
        ties = d == 0
        assert bool((p[1:][ties] > p[:-1][ties]).all())
        del p, s, d, ties, perm


@pytest.mark.gpu
def test_gpu_two_level_export_buckets_match_reference_hash(golden, oracle_mod):
    import clickhouse_amd as ch
    ctx = ch.Context()
    rng = np.random.Generator(np.random.PCG64(3))
    kat = {int(e["key"]): e["two_level_bucket"] for e in golden["kat"]}
    base = np.array(list(kat.keys()), dtype=np.uint64)
    keys = np.concatenate([base, base, rng.integers(0, 2**64, size=200_000, dtype=np.uint64)])
    vals = rng.integers(-1000, 1000, size=keys.shape[0], dtype=np.int64)
    agg = ch.Aggregator(np.uint64, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None), (ch.AGG_AVG, np.int64)], ctx=ctx)
    agg.execute_on_block(ctx.upload(keys), [ctx.upload(vals), None, ctx.upload(vals)])
    k, states, groups, counts = agg.export_state_columns_two_level()
    k = k.numpy()
    assert groups == len(np.unique(keys)) == k.shape[0] == sum(counts) and len(set(k.tolist())) == groups
    bucket_of_row = np.repeat(np.arange(256), counts)
    want_bucket = (oracle_mod.hash_crc32(k) >> 24) & 0xFF          # getBucketFromHash over the restated HashCRC32
    assert np.array_equal(bucket_of_row, want_bucket.astype(np.int64))
    for key, b in kat.items():                                      # the reference's own Hash.h answers
        assert bucket_of_row[np.nonzero(k == np.uint64(key))[0][0]] == b
    # the states travelled with their keys
    s, c, num, den = [x.numpy() for x in states]
    order = np.argsort(k, kind="stable")
    uk, inv = np.unique(keys, return_inverse=True)
    ws = np.zeros(uk.shape[0], dtype=np.int64)
    np.add.at(ws, inv, vals)
    wc = np.bincount(inv, minlength=uk.shape[0])
    assert np.array_equal(k[order], uk) and np.array_equal(s[order].view(np.int64), ws) and np.array_equal(c[order], wc.astype(np.uint64))
    assert np.array_equal(num[order].view(np.int64), ws) and np.array_equal(den[order], wc.astype(np.uint64))


def _stable_order(x, descending):
    """numpy restatement of the stable order for large inputs (NaN last in both directions): order-preserving unsigned keys"""
    dt = x.dtype
    if dt.kind == "f":
        key = np.where(np.isnan(x), 0.0, x).astype(np.float64)
        # primary: NaN last; then the value (-0.0 == 0.0); ties by row number
        order = np.lexsort((np.arange(x.shape[0]), -key if descending else key, np.isnan(x)))
        return order.astype(np.uint64)
    u = x.astype(np.int64).view(np.uint64) ^ np.uint64(1 << 63) if dt.kind == "i" else x.astype(np.uint64)
    return np.argsort(~u if descending else u, kind="stable").astype(np.uint64)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.int64, np.uint32, np.float64, np.float32, np.int16, np.uint64])
def test_gpu_sort_permutation_limit_is_the_prefix_of_the_full_stable_sort(dtype):
    """ORDER BY x LIMIT n: sampled threshold + candidates, or one of its fallbacks -- always the exact prefix"""
    import clickhouse_amd as ch
    ctx = ch.Context()
    rng = np.random.Generator(np.random.PCG64(np.dtype(dtype).itemsize + 100))
    n = (1 << 21) + 12345
    for flavour in ("wide", "few values", "one value repeated", "nans"):
        if flavour == "nans" and np.dtype(dtype).kind != "f":
            continue
        x = _column(rng, dtype, n, few_values=(flavour == "few values"))
        if flavour == "one value repeated":
            x[rng.random(n) < 0.6] = x[0]
        if flavour == "nans":
            x[rng.random(n) < 0.3] = np.nan
        col = ctx.upload(x)
        for desc in (False, True):
            want = _stable_order(x, desc)
            hint = -1 if desc else 1  # NaN last
            for limit in (1, 10, 1000, 30_000, n // 2, n + 5):
                got = ch.sort_permutation_limit(col, limit, desc, hint).numpy()
                assert np.array_equal(got, want[:limit]), (dtype, flavour, desc, limit)
        if np.dtype(dtype).kind == "f":  # NaN first: no threshold describes it -> the full sort, still exact
            got = ch.sort_permutation_limit(col, 100, False, -1).numpy()
            full = ch.sort_permutation(col, None, False, -1).numpy()
            assert np.array_equal(got, full[:100])
    # sortBlock with one sort column and LIMIT takes this path; filterToIndices
    (o,), perm = ch.sort_block([col], [(0, False, 1)], limit=7)
    assert o.size() == 7 and np.array_equal(perm.numpy(), _stable_order(x, False)[:7])
    m = (rng.random(n) < 0.01).astype(np.uint8)
    assert np.array_equal(ch.filter_to_indices(ctx.upload(m)).numpy(), np.nonzero(m)[0].astype(np.uint64))


def _nan_order_cases():
    """03447_float_nan_order: a = number + number / number (NaN for number = 0) over numbers(3) and numbers(256), ORDER BY a ASC | DESC
    NULLS FIRST | LAST.  nulls_direction: NaN counts as greater than every number when it is +1 -- ASC NULLS LAST and DESC NULLS FIRST."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sort_nan_order.json")) as f:
        blocks = json.load(f)["blocks"]
    for size, n in (("short", 3), ("long", 256)):
        number = np.arange(n, dtype=np.uint64)
        with np.errstate(invalid="ignore", divide="ignore"):
            a = number.astype(np.float64) + number.astype(np.float64) / number.astype(np.float64)
        for direction, desc in (("ASC", False), ("DESC", True)):
            for nulls, hint in (("FIRST", 1 if desc else -1), ("LAST", -1 if desc else 1)):
                for partial in ("", "partial "):
                    want = [float(x) for x in blocks[f"{size} array {partial}{direction} NULLS {nulls}"]]
                    yield a, desc, hint, want


def _same_floats(got, want):
    return len(got) == len(want) and all((g == w) or (g != g and w != w) for g, w in zip(got, want))


def test_oracle_sort_nan_order_reference_blocks():
    n = 0
    for a, desc, hint, want in _nan_order_cases():
        assert _same_floats(a[OS.get_permutation(a, desc, hint).astype(np.int64)].tolist(), want), (a.shape, desc, hint)
        n += 1
    assert n == 16


@pytest.mark.gpu
def test_gpu_sort_nan_order_reference_blocks():
    import clickhouse_amd as ch
    ctx = ch.Context()
    for a, desc, hint, want in _nan_order_cases():
        col = ctx.upload(a)
        perm = ch.sort_permutation(col, None, desc, hint)
        assert _same_floats(col.index(perm).numpy().tolist(), want), (a.shape, desc, hint)
        f32 = ctx.upload(a.astype(np.float32))
        assert _same_floats(f32.index(ch.sort_permutation(f32, None, desc, hint)).numpy().astype(np.float64).tolist(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("key_dtype,val_dtype,groups", [(np.uint32, np.int64, 1000), (np.int64, np.int32, 50_000), (np.uint16, np.uint8, 7), (np.uint64, np.uint64, 1)])
def test_gpu_group_by_min_max_through_ordered_output(key_dtype, val_dtype, groups):
    """GROUP BY key -> min(value), max(value) ORDER BY key: min / max states of the hash aggregator, the groups sorted afterwards; numpy
    minimum.at / maximum.at beside it (tests/test_gpu_agg_join.py holds the oracle and reference-row comparisons)"""
    import clickhouse_amd as ch
    ctx = ch.Context()
    rng = np.random.Generator(np.random.PCG64(groups))
    n = 300_007
    ki = np.iinfo(key_dtype)
    k = (rng.integers(0, groups, size=n).astype(np.int64) + (ki.min if ki.min < 0 else 0) // 2).astype(key_dtype)
    v = _column(rng, val_dtype, n, few_values=False)
    gk, gmin, gmax = ch.group_by_min_max(ctx, ctx.upload(k), ctx.upload(v))
    uk, inv = np.unique(k, return_inverse=True)
    wmin = np.full(uk.shape[0], np.iinfo(val_dtype).max, dtype=val_dtype)
    wmax = np.full(uk.shape[0], np.iinfo(val_dtype).min, dtype=val_dtype)
    np.minimum.at(wmin, inv, v)
    np.maximum.at(wmax, inv, v)
    assert np.array_equal(gk.numpy(), uk) and np.array_equal(gmin.numpy(), wmin) and np.array_equal(gmax.numpy(), wmax)
    vf = v.astype(np.float64) * 0.5
    gk, gmin, gmax = ch.group_by_min_max(ctx, ctx.upload(k), ctx.upload(vf))
    fmin, fmax = np.full(uk.shape[0], np.inf), np.full(uk.shape[0], -np.inf)
    np.minimum.at(fmin, inv, vf)
    np.maximum.at(fmax, inv, vf)
    assert np.array_equal(gk.numpy(), uk) and np.array_equal(gmin.numpy(), fmin) and np.array_equal(gmax.numpy(), fmax)
