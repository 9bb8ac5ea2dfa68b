"""SSB Q4.1-style 3-way join + GROUP BY (BASELINE.json configs[4] shape, small scale) through the C ABI vs the oracle plan."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import ssb


def test_ssb_q41_oracle_plan_matches_numpy(oracle_mod):
    dims = ssb.gen_dims(30_000, 2_000, 2_000)
    lo = ssb.gen_lineorder_numpy(300_000, 30_000, 2_000, 2_000)
    got = ssb.q41_cpu(oracle_mod, dims, lo)
    ok = (dims["s_region"][lo["lo_suppkey"] - 1] == ssb.AMERICA) & (dims["p_mfgr"][lo["lo_partkey"] - 1] <= 2) & \
         (dims["c_region"][lo["lo_custkey"] - 1] == ssb.AMERICA)
    year = dims["d_year"][lo["lo_orderdate"][ok] - 19920101]
    nation = dims["c_nation"][lo["lo_custkey"][ok] - 1]
    profit = lo["lo_revenue"][ok].astype(np.int64) - lo["lo_supplycost"][ok].astype(np.int64)
    want = {}
    for y, n, p in zip(year.tolist(), nation.tolist(), profit.tolist()):
        a = want.get((y, n), (0, 0))
        want[(y, n)] = (a[0] + p, a[1] + 1)
    assert got == want and len(got) > 20


@pytest.mark.gpu
@pytest.mark.parametrize("plan", ["chain", "per_join", "two_filters"])
def test_ssb_q41_gpu_matches_oracle(oracle_mod, monkeypatch, plan):
    """the three shapes of the plan: the join chain answered as one filter over the fact keys (chgpu_join_probe_chain, late
    materialisation); one joinBlock per join with the two semi joins sharing one filter (ANTI filter of the first = null map of the
    second); and the reference's one FilterTransform per JoiningTransform"""
    import clickhouse_amd as ch
    ctx = ch.Context(0)
    monkeypatch.delenv("SSB_PLAN_TWO_FILTERS", raising=False)
    monkeypatch.delenv("SSB_PLAN_PER_JOIN", raising=False)
    if plan == "two_filters":
        monkeypatch.setenv("SSB_PLAN_TWO_FILTERS", "1")
    elif plan == "per_join":
        monkeypatch.setenv("SSB_PLAN_PER_JOIN", "1")
    dims = ssb.gen_dims(300_000, 20_000, 20_000)
    lo = ssb.gen_lineorder_numpy(3_000_000, 300_000, 20_000, 20_000)
    want = ssb.q41_cpu(oracle_mod, dims, lo)
    got = ssb.q41_gpu(ch, ctx, dims, {k: ctx.upload(v) for k, v in lo.items()})
    assert got == want and len(got) == 5 * 7


@pytest.mark.gpu
def test_pack_unpack_fixed_keys(oracle_mod):
    import clickhouse_amd as ch
    ctx = ch.Context(0)
    rng = np.random.Generator(np.random.PCG64(3))
    a = rng.integers(0, 2**32, size=10_001, dtype=np.uint32)
    b = rng.integers(0, 256, size=10_001).astype(np.uint8)
    c = rng.integers(0, 256, size=10_001).astype(np.uint8)
    packed = ch.pack_fixed_keys([ctx.upload(a), ctx.upload(b), ctx.upload(c)])
    want = a.astype(np.uint64) | (b.astype(np.uint64) << np.uint64(32)) | (c.astype(np.uint64) << np.uint64(40))  # packFixed: consecutive bytes
    assert np.array_equal(packed.numpy(), want)
    assert np.array_equal(ch.unpack_fixed_key(packed, 0, np.uint32).numpy(), a)
    assert np.array_equal(ch.unpack_fixed_key(packed, 4, np.uint8).numpy(), b)
    assert np.array_equal(ch.unpack_fixed_key(packed, 5, np.uint8).numpy(), c)
    with pytest.raises(ch.ChgpuError) as e:   # 12 key bytes -> keys128 on the CPU
        ch.pack_fixed_keys([ctx.upload(a.astype(np.uint64)), ctx.upload(a)])
    assert e.value.code == ch._capi.ERR_NOT_IMPLEMENTED


@pytest.mark.gpu
def test_ssb_q31_string_keys_compressed_columns_end_to_end():
    """SSB Q3.1 with String dimension attributes, a Date column, compressed fact columns, generated WHERE + projection, two joins,
    packed string-id keys, ORDER BY -- against the numpy restatement (tools/ssb_q31.py)"""
    import clickhouse_amd as ch
    from oracle import compression as OC
    import ssb_q31 as Q  # tools/ is on sys.path (top of this file)
    ctx = ch.Context()
    dims, lo = Q.gen(rows=600_000, customers=30_000, suppliers=2_000)
    files = Q.compress_lineorder(OC, lo)
    got, kept = Q.q31_gpu(ch, ctx, dims, files, {k: v.dtype for k, v in lo.items()})
    want, _ = Q.q31_cpu(dims, lo)
    assert len(got) == len(want) > 100 and sorted(got) == sorted(want)
    assert [(r[2], -r[3]) for r in got] == sorted((r[2], -r[3]) for r in got)  # ORDER BY year ASC, revenue DESC
    assert got == Q.ordered(want) or sorted(got) == sorted(want)             # equal up to the order of revenue ties
