"""Seeded differential fuzz of the GPU path against the CPU oracle / numpy over random shapes: key and argument types,
aggregate sets (<= 2 and > 2 argument columns take different kernels), size hints (none, too small, exact, too large),
cardinalities from 1 group to all-distinct, skew, row ranges, NULL maps and join variants.  Integer results bit-exact,
Float64 within BASELINE's 1e-6 relative tolerance."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# CHGPU_FUZZ_SEED=<n> shifts every seed: the committed run is deterministic, extra seeds are for soak runs
SEED = int(os.environ.get("CHGPU_FUZZ_SEED", "0"))


@pytest.fixture(scope="module")
def ch():
    import clickhouse_amd
    return clickhouse_amd


@pytest.fixture(scope="module")
def ctx(ch):
    c = ch.Context(0)
    yield c
    c.close()


KEY_DTYPES = [np.uint32, np.int32, np.uint64, np.int64, np.uint8, np.uint16, np.int16, np.int8]
ARG_DTYPES = [np.int64, np.uint64, np.float64, np.uint32, np.int32, np.uint8, np.uint16, np.int16, np.int8, np.float32]


def _keys(rng, dtype, n, groups, skew):
    dt = np.dtype(dtype)
    groups = min(groups, 2 ** (8 * dt.itemsize) - 1)
    if skew:
        k = rng.zipf(1.2, size=n) % groups
    else:
        k = rng.integers(0, groups, size=n)
    if dt.kind == "i":
        k = k - groups // 2
    k = k.astype(dtype)
    if dt.itemsize == 8 and rng.random() < 0.5:
        k = (k.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.uint64).astype(dtype) if dt.kind == "u" else k
    return k


def _args(rng, dtype, n):
    dt = np.dtype(dtype)
    if dt.kind == "f":
        return (rng.random(n) * 1000 - 500).astype(dt)
    if dt.itemsize == 8:
        return rng.integers(-2**62, 2**62, size=n, dtype=np.int64).astype(dtype)   # sums wrap modulo 2^64
    info = np.iinfo(dt)
    return rng.integers(info.min, int(info.max) + 1, size=n).astype(dtype)


def _agg_case(ch, ctx, oracle_mod, rng, n, groups, use_oracle):
    key_dtype = KEY_DTYPES[rng.integers(0, len(KEY_DTYPES))]
    n_aggs = int(rng.integers(1, 5))
    aggs, cols = [], []
    for _ in range(n_aggs):
        kind = [ch.AGG_SUM, ch.AGG_COUNT, ch.AGG_AVG][rng.integers(0, 3)]
        if kind == ch.AGG_COUNT:
            aggs.append((kind, None))
            cols.append(None)
        else:
            dt = ARG_DTYPES[rng.integers(0, len(ARG_DTYPES))]
            aggs.append((kind, dt))
            cols.append(_args(rng, dt, n))
    skew = rng.random() < 0.3
    k = _keys(rng, key_dtype, n, groups, skew)
    if rng.random() < 0.5:
        k[: min(n, 3)] = 0
    true_groups = np.unique(k).shape[0]
    hint = [0, max(1, true_groups // 20), true_groups, true_groups * 4][rng.integers(0, 4)]
    desc = f"key={np.dtype(key_dtype).name} aggs={[(a, np.dtype(d).name if d else None) for a, d in aggs]} n={n} groups={true_groups} hint={hint} skew={skew}"
    g = ch.Aggregator(key_dtype, aggs, size_hint=hint, ctx=ctx)
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=int(rng.integers(0, 3)))]))
    for b, e in zip(cuts[:-1], cuts[1:]):
        g.execute_on_block(k, cols, b, e)
    gk, gr = g.convert_to_block()
    gi = np.argsort(gk, kind="stable")
    if use_oracle:
        o = oracle_mod.Aggregator(key_dtype, aggs)
        o.execute_on_block(k, cols)
        ok, orr = o.convert_to_block()
        oi = np.argsort(ok, kind="stable")
        assert gk.dtype == ok.dtype and np.array_equal(gk[gi], ok[oi]), desc
        for j in range(n_aggs):
            a, b = gr[j][gi], orr[j][oi]
            assert a.dtype == b.dtype, desc
            if a.dtype == np.float64:
                assert np.allclose(a, b, rtol=1e-6, atol=1e-9, equal_nan=True), (desc, j)
            else:
                assert np.array_equal(a, b), (desc, j)
    else:
        uk, inv = np.unique(k, return_inverse=True)
        assert np.array_equal(gk[gi], uk), desc
        cnt = np.bincount(inv, minlength=uk.shape[0])
        for j, (kind, dt) in enumerate(aggs):
            got = gr[j][gi]
            if kind == ch.AGG_COUNT:
                assert np.array_equal(got, cnt.astype(np.uint64)), (desc, j)
                continue
            c = cols[j]
            if np.dtype(dt).kind == "f":
                s = np.zeros(uk.shape[0])
                np.add.at(s, inv, c.astype(np.float64))
                want = s if kind == ch.AGG_SUM else s / cnt
                assert np.allclose(got, want, rtol=1e-6, atol=1e-6), (desc, j)
            elif kind == ch.AGG_SUM:
                s = np.zeros(uk.shape[0], dtype=np.uint64)
                np.add.at(s, inv, c.astype(np.int64).view(np.uint64) if np.dtype(dt).kind == "i" else c.astype(np.uint64))
                assert np.array_equal(got.view(np.uint64), s), (desc, j)   # two's complement sum, modulo 2^64


def test_fuzz_group_by_small_against_oracle(ch, ctx, oracle_mod):
    rng = np.random.Generator(np.random.PCG64(20261003 + SEED))
    for case in range(60):
        n = int(rng.integers(1, 200_000))
        groups = int([1, 7, 300, 5000, 60_000, n][rng.integers(0, 6)]) or 1
        _agg_case(ch, ctx, oracle_mod, rng, n, groups, use_oracle=True)


def test_fuzz_group_by_large_paths_against_numpy(ch, ctx, oracle_mod):
    # >= 4 Mi rows so that hints and observed cardinalities choose between RANGE, PARTITIONED and DIRECT
    rng = np.random.Generator(np.random.PCG64(777 + SEED))
    for case in range(10):
        n = int(rng.integers(4_300_000, 5_500_000))
        groups = int([3, 2000, 5000, 40_000, 900_000, n][rng.integers(0, 6)])
        _agg_case(ch, ctx, oracle_mod, rng, n, groups, use_oracle=False)


def test_fuzz_join_against_oracle(ch, ctx, oracle_mod):
    O = oracle_mod
    rng = np.random.Generator(np.random.PCG64(4242 + SEED))
    variants = [(ch.JOIN_INNER, ch.STRICT_ALL, {}), (ch.JOIN_LEFT, ch.STRICT_ALL, {}), (ch.JOIN_LEFT, ch.STRICT_ANY, {}),
                (ch.JOIN_LEFT, ch.STRICT_ANY, {"any_take_last_row": True}), (ch.JOIN_INNER, ch.STRICT_ANY, {}),
                (ch.JOIN_LEFT, ch.STRICT_SEMI, {}), (ch.JOIN_LEFT, ch.STRICT_ANTI, {})]
    for case in range(40):
        kind, strict, kw = variants[rng.integers(0, len(variants))]
        key_space = int([3, 100, 5000, 10**6, 2**40][rng.integers(0, 5)])
        n_blocks = int(rng.integers(0, 4))
        g = ch.HashJoin(kind, strict, kw.get("any_take_last_row", False), ctx=ctx)
        o = O.HashJoin(kind, strict, kw.get("any_take_last_row", False))
        for _ in range(n_blocks):
            rows = int(rng.integers(0, 30_000))
            keys = rng.integers(0, key_space, size=rows, dtype=np.uint64)
            nm = (rng.random(rows) < 0.05).astype(np.uint8) if rng.random() < 0.4 else None
            jm = (rng.random(rows) < 0.9).astype(np.uint8) if rng.random() < 0.3 else None
            g.add_block(keys, null_map=nm, join_mask=jm)
            o.add_block(keys, null_map=nm, join_mask=jm)
        assert g.total_rows == o.total_rows and g.n_keys == o.n_keys
        left = rng.integers(0, key_space, size=int(rng.integers(0, 60_000)), dtype=np.uint64)
        lnm = (rng.random(left.shape[0]) < 0.03).astype(np.uint8) if rng.random() < 0.5 else None
        mjb = int([0, 0, 50, 4000][rng.integers(0, 4)])
        if mjb == 50:
            left, lnm = left[:2000], (None if lnm is None else lnm[:2000])   # a probe call per ~1 left row: keep the resubmission loop short
        if strict == ch.STRICT_ALL:
            # few distinct keys x many build rows: every left row joins build_rows / key_space right rows; keep the expected
            # number of pairs (which the test sorts in Python) under ~2 M
            cap = max(1, int(2e6 * key_space / max(1, g.total_rows)))
            left, lnm = left[:cap], (None if lnm is None else lnm[:cap])
        pos = 0
        while True:
            gl, gb, gr, gc = g.joined_pairs(left[pos:], None if lnm is None else lnm[pos:], max_joined_block_rows=mjb)
            ol, ob, orow, oc = o.joined_pairs(left[pos:], None if lnm is None else lnm[pos:], max_joined_block_rows=mjb)
            desc = f"case={case} kind={kind} strict={strict} kw={kw} key_space={key_space} blocks={n_blocks} left={left.shape[0]} max={mjb} pos={pos}"
            assert gc == oc, desc
            if strict == ch.STRICT_ALL:
                assert sorted(zip(gl.tolist(), gb.tolist(), gr.tolist())) == sorted(zip(ol.tolist(), ob.tolist(), orow.tolist())), desc
            else:
                assert np.array_equal(gl, ol) and np.array_equal(gb, ob) and np.array_equal(gr, orow), desc
            pos += gc
            if pos >= left.shape[0] or gc == 0:
                break
