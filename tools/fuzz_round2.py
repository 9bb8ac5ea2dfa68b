#!/usr/bin/env python3
"""Randomised soak of the round-2 device paths against numpy (run on the GPU box): the tile-sorted GROUP BY (row counts around tile and
workgroup boundaries, key / argument types, skew, hints far off), the LDS-staged filter-only join probe (key domains around the slice
boundaries) and the multi-column filter.  usage: fuzz_round2.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import clickhouse_amd as ch

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.Generator(np.random.PCG64(seed))
ctx = ch.Context(0)
t0 = time.time()
n_cases = {"groupby": 0, "semi": 0, "filter": 0, "join_agg": 0}
print("seed", seed, flush=True)


def case_groupby():
    tile = int(rng.choice([12288, 8192]))
    cus = 256
    base = int(rng.integers(3_200_000, 9_000_000))
    # land on / next to multiples of the tile and of tile * CUs now and then
    n = int(rng.choice([base, base // tile * tile, base // tile * tile + 1, base // tile * tile - 1, (base // (tile * cus) + 1) * tile * cus + int(rng.integers(0, 3))]))
    key_dtype = rng.choice([np.uint32, np.int32, np.uint64])
    arg_dtype = rng.choice([np.int64, np.uint64, np.float64, np.uint32, np.int32, np.float32])
    groups = int(rng.choice([20_000, 100_000, 400_000, 1_500_000]))
    hint = int(groups * rng.choice([1.0, 1.0, 0.05, 3.0]))
    pattern = rng.choice(["uniform", "hot", "zipf"])
    if pattern == "uniform":
        k = rng.integers(0, groups, size=n)
    elif pattern == "hot":
        k = rng.integers(0, groups, size=n)
        k[rng.random(n) < rng.choice([0.3, 0.9])] = int(rng.integers(0, groups))
    else:
        k = np.minimum(rng.zipf(1.3, size=n), groups - 1)
    k = k.astype(key_dtype)
    if np.dtype(arg_dtype).kind == "f":
        v = (rng.random(n) * 2000 - 1000).astype(arg_dtype)
    else:
        info = np.iinfo(arg_dtype)
        v = rng.integers(max(info.min, -2**40), min(info.max, 2**40), size=n).astype(arg_dtype)   # (no wrap-around of a group's 64-bit sum)
    aggs = [[(ch.AGG_SUM, arg_dtype), (ch.AGG_COUNT, None)], [(ch.AGG_COUNT, None), (ch.AGG_SUM, arg_dtype)], [(ch.AGG_SUM, arg_dtype)], [(ch.AGG_AVG, arg_dtype)]][int(rng.integers(0, 4))]
    A = ch.Aggregator(key_dtype, aggs, size_hint=hint, ctx=ctx)
    A.execute_on_block(ctx.upload(k), [ctx.upload(v) if kind != ch.AGG_COUNT else None for kind, _ in aggs])
    gk, res = A.convert_to_block()
    uk, inv = np.unique(k, return_inverse=True)
    order = np.argsort(gk)
    assert np.array_equal(gk[order], uk), ("keys", n, key_dtype, arg_dtype, groups, hint, pattern)
    cnt = np.bincount(inv, minlength=uk.shape[0])
    for (kind, _), r in zip(aggs, res):
        if kind == ch.AGG_COUNT:
            assert np.array_equal(r[order], cnt.astype(np.uint64)), ("count", n, key_dtype, arg_dtype, groups, hint, pattern)
        elif np.dtype(arg_dtype).kind == "f":
            s = np.bincount(inv, weights=v.astype(np.float64), minlength=uk.shape[0])
            want = s / cnt if kind == ch.AGG_AVG else s
            assert np.allclose(r[order], want, rtol=1e-9, atol=1e-6), ("fsum", n, key_dtype, arg_dtype, groups, hint, pattern)
        elif kind == ch.AGG_AVG:
            isum = np.zeros(uk.shape[0], dtype=np.int64)
            np.add.at(isum, inv, v.astype(np.int64))
            want = (isum.astype(np.uint64) if np.dtype(arg_dtype).kind == "u" else isum).astype(np.float64) / cnt
            assert np.allclose(r[order], want, rtol=1e-12, atol=0), ("iavg", n, key_dtype, arg_dtype, groups, hint, pattern)
        else:
            want = np.zeros(uk.shape[0], dtype=np.uint64)
            np.add.at(want, inv, v.astype(np.int64).astype(np.uint64))
            assert np.array_equal(r[order].astype(np.uint64), want), ("isum", n, key_dtype, arg_dtype, groups, hint, pattern)
    n_cases["groupby"] += 1


def case_semi():
    slice_bits = 150 * 1024 * 8
    domain = int(rng.choice([5000, slice_bits - 1, slice_bits, slice_bits + 1, 2 * slice_bits, 2 * slice_bits + 33, 4 * slice_bits - 31, 4 * slice_bits, int(rng.integers(10_000, 5_000_000))]))
    n = int(rng.integers(1 << 20, 6_000_000))
    build = rng.integers(0, domain, size=max(domain // 16 + 1, 1000)).astype(np.uint32)
    build[0] = domain - 1
    if rng.random() < 0.5:
        build[1] = 0
    left = rng.integers(0, domain + domain // 2 + 2, size=n).astype(np.uint32)
    nulls = (rng.random(n) < 0.05).astype(np.uint8) if rng.random() < 0.5 else None
    strict = ch.STRICT_SEMI if rng.random() < 0.5 else ch.STRICT_ANTI
    j = ch.HashJoin(ch.JOIN_LEFT, strict, key_dtype=np.uint32, ctx=ctx)
    j.add_block(build)
    r = j.probe_columns(left, null_map=nulls, need_right_rows=False)
    found = np.isin(left, build)
    if nulls is not None:
        found &= nulls == 0
    want = found if strict == ch.STRICT_SEMI else ~found
    assert np.array_equal(r["filter"].numpy().astype(bool), want) and r["n_out"] == int(want.sum()), ("semi", domain, n, strict, nulls is not None)
    n_cases["semi"] += 1


def case_filter():
    n = int(rng.integers(1, 3_000_000))
    cols = []
    for _ in range(int(rng.integers(1, 9))):
        dt = rng.choice([np.uint32, np.int64, np.float64, np.uint16, np.uint8, np.float32])
        cols.append(rng.integers(0, 200, size=n).astype(dt))
    f = (rng.random(n) < rng.choice([0.0, 0.01, 0.3, 1.0])).astype(np.uint8) * int(rng.integers(1, 255))
    start = int(rng.integers(0, min(n, 5)))
    dev = [ctx.upload(c).cut(start, n - start) for c in cols]
    outs = ch.filter_columns(dev, ctx.upload(f).cut(start, n - start))
    for o, c in zip(outs, cols):
        assert np.array_equal(o.numpy(), c[start:][f[start:] != 0]), ("filter", n, start, [str(c.dtype) for c in cols])
    n_cases["filter"] += 1


def case_join_agg():
    """the fused probe over one- and two-block build sides: the radix join (no table), the table's LDS-staged slices after the key count
    has built it, duplicate build keys (both fall back), every join variant; against a sorted-search join in numpy"""
    nb = int(rng.choice([1_100_000, 2_097_152, int(rng.integers(1_000_000, 4_500_000))]))
    npb = int(rng.integers(8_400_000, 10_500_000))
    space = int(rng.choice([2**40, 2**63, nb * 3]))
    if space == nb * 3:
        bk = rng.permutation(nb * 3)[:nb].astype(np.uint64)                      # dense small keys (incl. 0)
    else:
        bk = (rng.permutation(nb).astype(np.uint64) * np.uint64(2654435761) + np.uint64(int(rng.integers(0, 1000)))) % np.uint64(space)
        bk = np.unique(bk)
        nb = bk.shape[0]
        rng.shuffle(bk)
    dups = rng.random() < 0.25
    if dups:
        bk[-5:] = bk[:5]
    bv = rng.integers(-2**45, 2**45, size=nb, dtype=np.int64)
    pk = np.where(rng.random(npb) < rng.choice([0.1, 0.5, 0.95]), bk[rng.integers(0, nb, size=npb)], rng.integers(0, min(space * 2, 2**63), size=npb, dtype=np.uint64))
    kind, strict = [(ch.JOIN_INNER, ch.STRICT_ALL), (ch.JOIN_LEFT, ch.STRICT_ALL), (ch.JOIN_LEFT, ch.STRICT_SEMI), (ch.JOIN_LEFT, ch.STRICT_ANTI)][int(rng.integers(0, 4))]
    j = ch.HashJoin(kind, strict, ctx=ctx)
    blocks = 1 if rng.random() < 0.7 else 2
    cut = nb // blocks
    for b in range(blocks):
        j.add_block(bk[b * cut:(b + 1) * cut if b + 1 < blocks else nb])
    pkc, bvc = ctx.upload(pk), ctx.upload(bv)
    c, s = j.probe_count_sum(pkc, bvc)
    mult_keys, mult = np.unique(bk, return_counts=True)
    sums = np.zeros(mult_keys.shape[0], dtype=np.uint64)
    np.add.at(sums, np.searchsorted(mult_keys, bk), bv.astype(np.uint64))
    pos = np.searchsorted(mult_keys, pk)
    pos[pos == mult_keys.shape[0]] = 0
    hit = mult_keys[pos] == pk
    if strict == ch.STRICT_ANTI:
        want_c, want_s = int((~hit).sum()), 0
    elif strict == ch.STRICT_SEMI:
        want_c, want_s = int(hit.sum()), None if dups else int(sums[pos[hit]].sum(dtype=np.uint64))   # with duplicates SEMI takes ONE of the rows
    else:
        want_c = int(mult[pos[hit]].sum()) + (int((~hit).sum()) if kind == ch.JOIN_LEFT else 0)
        want_s = int(sums[pos[hit]].sum(dtype=np.uint64))
    assert c == want_c and (want_s is None or s % 2**64 == want_s), ("join_agg", nb, npb, space, dups, blocks, kind, strict, c, want_c)
    if rng.random() < 0.5:
        assert j.n_keys == mult_keys.shape[0]                                    # builds the table
        c2, s2 = j.probe_count_sum(pkc, bvc)
        assert c2 == c and (want_s is None or s2 == s), ("join_agg after the table", nb, npb, dups, blocks, kind, strict)
    n_cases["join_agg"] += 1


while time.time() - t0 < budget:
    x = rng.random()
    (case_groupby if x < 0.45 else case_semi if x < 0.65 else case_filter if x < 0.75 else case_join_agg)()
    if sum(n_cases.values()) % 5 == 0:
        print(f"{time.time() - t0:6.1f}s {n_cases}", flush=True)
print("fuzz_round2 OK", n_cases, flush=True)
