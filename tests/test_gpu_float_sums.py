"""Float64 GROUP BY sums are bit-reproducible: the state of sum / avg over a float argument is a 128-bit fixed-point integer
(agg_kernels.hip, Fx128), so the order in which the hardware serves the rows cannot show.  The reference adds in row order and is
deterministic for a fixed block split (AggregateFunctionSum.h:72-101); here every plan gives the SAME bits -- and, while the magnitudes of a
column stay within 2^44 of each other, those bits are the correctly rounded exact sum (math.fsum), which the oracle's row-order double sum
matches to 1e-6 relative and better."""
import math

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


def _exact(k, v):
    uk, inv = np.unique(k, return_inverse=True)
    order = np.argsort(inv, kind="stable")
    cuts = np.searchsorted(inv[order], np.arange(uk.shape[0] + 1))
    vs = v[order]
    return uk, np.array([math.fsum(vs[cuts[g]:cuts[g + 1]]) for g in range(uk.shape[0])])


def _run(ch, ctx, k, v, hint, aggs=None, blocks=1):
    aggs = aggs or [(ch.AGG_SUM, v.dtype.type), (ch.AGG_COUNT, None)]
    A = ch.Aggregator(k.dtype.type, aggs, size_hint=hint, ctx=ctx)
    for part in range(blocks):
        lo, hi = part * k.shape[0] // blocks, (part + 1) * k.shape[0] // blocks
        A.execute_on_block(ctx.upload(k[lo:hi]), [ctx.upload(v[lo:hi]) if a[0] != ch.AGG_COUNT else None for a in aggs])
    gk, res = A.convert_to_block()
    order = np.argsort(gk)
    return gk[order], [r[order] for r in res]


@pytest.mark.parametrize("shape", ["lds_range", "tile_sorted", "scatter_u64", "direct_small", "rows_lds"])
def test_float_group_sums_are_exact_and_identical_in_every_plan(ch, ctx, shape):
    rng = np.random.Generator(np.random.PCG64(5))
    if shape == "lds_range":
        n, groups, hint, kt = 3_000_000, 900, 1000, np.uint32
    elif shape == "tile_sorted":
        n, groups, hint, kt = 7_340_033, 250_000, 250_000, np.uint32
    elif shape == "scatter_u64":
        n, groups, hint, kt = 5_000_000, 300_000, 300_000, np.uint64
    elif shape == "rows_lds":
        n, groups, hint, kt = 200_000, 30_000, 30_000, np.uint16
    else:
        n, groups, hint, kt = 100_000, 70_000, 1_000_000, np.uint64
    k = rng.integers(0, groups, size=n).astype(kt)
    v = (rng.random(n) * 2e6 - 1e6) * np.exp2(rng.integers(-20, 20, size=n))       # magnitudes over 2^40: inside the exact window
    v[::1013] = 0.0
    v[5::4001] = -0.0
    uk, want = _exact(k, v)
    runs = [_run(ch, ctx, k, v, hint) for _ in range(2)]
    for gk, (gs, gc) in runs:
        assert np.array_equal(gk, uk)
        assert np.array_equal(gs.view(np.uint64), want.view(np.uint64)), np.abs(gs - want).max()      # correctly rounded exact sums, bit for bit
        assert np.array_equal(gc, np.bincount(np.unique(k, return_inverse=True)[1]).astype(np.uint64))
    # other plans over the same rows: no hint (cardinality probe + whatever it picks) and three blocks
    for hint2, blocks in ((0, 1), (hint, 3)):
        gk, (gs, gc) = _run(ch, ctx, k, v, hint2, blocks=blocks)
        assert np.array_equal(gs.view(np.uint64), want.view(np.uint64))


def test_float32_and_avg_are_exact_too(ch, ctx, oracle_mod):
    rng = np.random.Generator(np.random.PCG64(6))
    n = 1_500_000
    k = rng.integers(0, 5000, size=n).astype(np.uint32)
    v = (rng.random(n) * 200 - 100).astype(np.float32)
    w = rng.random(n) * 1e-3
    aggs = [(ch.AGG_AVG, np.float32), (ch.AGG_SUM, np.float64), (ch.AGG_AVG, np.float64)]
    A = ch.Aggregator(np.uint32, aggs, size_hint=5000, ctx=ctx)
    A.execute_on_block(ctx.upload(k), [ctx.upload(v), ctx.upload(w), ctx.upload(w)])
    gk, (a32, s64, a64) = A.convert_to_block()
    order = np.argsort(gk)
    uk, e32 = _exact(k, v.astype(np.float64))
    _, e64 = _exact(k, w)
    cnt = np.bincount(np.unique(k, return_inverse=True)[1]).astype(np.float64)
    assert np.array_equal(a32[order].view(np.uint64), (e32 / cnt).view(np.uint64))
    assert np.array_equal(s64[order].view(np.uint64), e64.view(np.uint64))
    assert np.array_equal(a64[order].view(np.uint64), (e64 / cnt).view(np.uint64))
    R = oracle_mod.Aggregator(np.uint32, aggs)
    R.execute_on_block(k, [v, w, w])
    ok, (o32, os64, o64) = R.convert_to_block()
    oo = np.argsort(ok)
    assert np.allclose(a32[order], o32[oo], rtol=1e-6) and np.allclose(s64[order], os64[oo], rtol=1e-6) and np.allclose(a64[order], o64[oo], rtol=1e-6)


def test_window_moves_with_the_blocks_and_the_row_count(ch, ctx):
    """Later blocks bring larger magnitudes (the states are shifted to a coarser unit) and smaller ones; the result stays within one unit of
    the window per shift, far inside 1e-12 relative, and identical from run to run."""
    rng = np.random.Generator(np.random.PCG64(7))
    n = 400_000
    k = rng.integers(0, 3000, size=n).astype(np.uint64)
    scales = [1.0, 1e9, 1e-3, 1e12, 1.0]
    blocks = [(rng.random(n) + 0.5) * s for s in scales]                            # 2^52 between the smallest and the largest value: one window holds them
    outs = []
    for _ in range(2):
        A = ch.Aggregator(np.uint64, [(ch.AGG_SUM, np.float64)], size_hint=3000, ctx=ctx)
        for b in blocks:
            A.execute_on_block(ctx.upload(k), [ctx.upload(b)])
        gk, (gs,) = A.convert_to_block()
        outs.append(gs[np.argsort(gk)])
    assert np.array_equal(outs[0].view(np.uint64), outs[1].view(np.uint64))
    uk, want = _exact(np.tile(k, len(scales)), np.concatenate(blocks))
    assert np.allclose(outs[0], want, rtol=1e-13, atol=0)


def test_nan_inf_and_huge_ranges_fall_back_to_double_states(ch, ctx, oracle_mod):
    rng = np.random.Generator(np.random.PCG64(8))
    n = 300_000
    k = rng.integers(0, 1000, size=n).astype(np.uint32)
    v = rng.random(n) * 10
    first = ctx.upload(v)
    poisoned = v.copy()
    poisoned[k == 7] = np.nan
    poisoned[np.flatnonzero(k == 8)[:1]] = np.inf
    poisoned[np.flatnonzero(k == 9)[:1]] = np.inf
    poisoned[np.flatnonzero(k == 9)[1:2]] = -np.inf
    A = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.float64), (ch.AGG_AVG, np.float64)], size_hint=1000, ctx=ctx)
    A.execute_on_block(ctx.upload(k), [first, first])                               # fixed-point states ...
    A.execute_on_block(ctx.upload(k), [ctx.upload(poisoned)] * 2)                   # ... turned into doubles by the block that carries a NaN
    gk, (gs, ga) = A.convert_to_block()
    order = np.argsort(gk)
    R = oracle_mod.Aggregator(np.uint32, [(ch.AGG_SUM, np.float64), (ch.AGG_AVG, np.float64)])
    R.execute_on_block(k, [v, v])
    R.execute_on_block(k, [poisoned, poisoned])
    ok, (os_, oa) = R.convert_to_block()
    oo = np.argsort(ok)
    assert np.array_equal(np.isnan(gs[order]), np.isnan(os_[oo])) and np.isnan(gs[order][7]) and np.isnan(gs[order][9]) and gs[order][8] == np.inf
    fin = np.isfinite(os_[oo])
    assert np.allclose(gs[order][fin], os_[oo][fin], rtol=1e-9) and np.allclose(ga[order][fin], oa[oo][fin], rtol=1e-9)
    # magnitudes 1e-30 and 1e30 in one column: the small group's values would vanish in any one window
    k2 = np.repeat(np.array([1, 2], dtype=np.uint32), 50_000)
    v2 = np.concatenate([rng.random(50_000) * 1e-30, rng.random(50_000) * 1e30])
    for blocks in (1, 2):                                                           # in one block, and the small values first
        gk2, (gs2, _) = _run(ch, ctx, k2, v2, 10, blocks=blocks)
        assert np.allclose(gs2, [v2[:50_000].sum(), v2[50_000:].sum()], rtol=1e-9)


def test_merge_and_state_round_trip_keep_the_sums_exact(ch, ctx):
    rng = np.random.Generator(np.random.PCG64(9))
    n = 500_000
    k = rng.integers(0, 20_000, size=n).astype(np.uint64)
    v = rng.random(n) * 1e3
    v2 = rng.random(n) * 1e7
    aggs = [(ch.AGG_SUM, np.float64), (ch.AGG_COUNT, None)]
    A = ch.Aggregator(np.uint64, aggs, size_hint=20_000, ctx=ctx)
    B = ch.Aggregator(np.uint64, aggs, size_hint=20_000, ctx=ctx)
    A.execute_on_block(ctx.upload(k), [ctx.upload(v), None])
    B.execute_on_block(ctx.upload(k[::-1].copy()), [ctx.upload(v2), None])
    A.merge(B)                                                                      # two windows meet in one
    uk, want = _exact(np.concatenate([k, k[::-1]]), np.concatenate([v, v2]))
    gk, (gs, gc) = A.convert_to_block()
    order = np.argsort(gk)
    assert np.array_equal(gk[order], uk) and np.array_equal(gs[order].view(np.uint64), want.view(np.uint64))
    # partial states leave as Float64 columns (the reference's wire format) and merge back into a fresh aggregation
    keys, states, rows = A.export_state_columns()
    assert len(states) == 2 and states[0].numpy().dtype == np.float64
    C = ch.Aggregator(np.uint64, aggs, size_hint=20_000, ctx=ctx)
    C.merge_states(keys, states, rows)
    C.execute_on_block(ctx.upload(k), [ctx.upload(v), None])
    gk3, (gs3, gc3) = C.convert_to_block()
    o3 = np.argsort(gk3)
    assert np.array_equal(np.unique(k), uk)
    per_key_rows = np.bincount(np.unique(k, return_inverse=True)[1]).astype(np.uint64)
    assert np.allclose(gs3[o3], want + _exact(k, v)[1], rtol=1e-15)                 # (one rounding at the export, one at the end)
    assert np.array_equal(gc3[o3], gc[order] + per_key_rows)


def test_option_off_keeps_double_states(ch):
    c2 = ch.Context(0)
    c2.set_option("deterministic_float_sums", 0)
    rng = np.random.Generator(np.random.PCG64(10))
    k = rng.integers(0, 100, size=100_000).astype(np.uint32)
    v = rng.random(100_000)
    A = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.float64)], size_hint=100, ctx=c2)
    A.execute_on_block(c2.upload(k), [c2.upload(v)])
    gk, (gs,) = A.convert_to_block()
    assert np.allclose(gs[np.argsort(gk)], np.bincount(k, weights=v), rtol=1e-12)
