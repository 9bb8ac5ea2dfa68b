"""The plans the benchmark numbers are quoted on, against oracle/ (not numpy) inside `-m gpu`:
  * C4 at its FULL size through chgpu_join_probe_agg (`probe_count_sum`: the radix join whose slices are built and probed in LDS, and after
    the table exists the LDS-staged slices of the table) vs oracle.join_count_sum_pipeline;
  * C3's tile-sorted GROUP BY plan on a 64 M-row sample (the oracle builds a 1 M-group table per stream: seconds) vs oracle.groupby_pipeline,
    integer sums and the Float64 variant (<= 1e-6 relative AND run-to-run bit-equal);
  * the join chain at SSB scale (2 M-key LDS-staged key sets of two slices, a 30 M-key bitmap from L2, a hash table) vs the oracle's
    joinBlock per join on a 40 M-row sample."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    import clickhouse_amd as ch
    ctx = ch.Context(0)
    yield ch, ctx, torch
    ctx.trim()
    ctx.close()


def _threads():
    return max(1, min(32, len(os.sched_getaffinity(0))))


def test_c4_full_size_fused_probe_matches_the_oracle(env, oracle_mod):
    ch, ctx, torch = env
    nb, npb = 10_000_000, 100_000_000
    g = torch.Generator(device="cuda").manual_seed(5)
    bk = (torch.randperm(nb, device="cuda", generator=g).to(torch.int64) + 1) * 2654435761
    bv = torch.randint(-2**40, 2**40, (nb,), dtype=torch.int64, device="cuda", generator=g)
    hit = torch.rand(npb, device="cuda", generator=g) < 0.5
    pk = torch.where(hit, bk[torch.randint(0, nb, (npb,), device="cuda", generator=g)], torch.randint(0, 2**62, (npb,), dtype=torch.int64, device="cuda", generator=g) * 2 + 1)
    torch.cuda.synchronize()
    want_cnt, want_sum, _, _ = oracle_mod.join_count_sum_pipeline(bk.cpu().numpy(), bv.cpu().numpy(), pk.cpu().numpy(), threads=_threads())
    bkc, pkc = ctx.wrap(bk.data_ptr(), np.uint64, nb, keepalive=bk), ctx.wrap(pk.data_ptr(), np.uint64, npb, keepalive=pk)
    bvc = ctx.wrap(bv.data_ptr(), np.int64, nb, keepalive=bv)
    j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, ctx=ctx)
    j.add_block(bkc)
    j.finish_build()
    c1, s1 = j.probe_count_sum(pkc, bvc)                       # no table yet: the radix join (both sides partitioned, slices built in LDS)
    assert (c1, s1 % 2**64) == (want_cnt, want_sum)
    assert j.n_keys == nb                                      # asks for the key count: the table is built now
    c2, s2 = j.probe_count_sum(pkc, bvc)                       # the table's slices staged in LDS
    assert (c2, s2 % 2**64) == (want_cnt, want_sum)


@pytest.mark.parametrize("val", ["int64", "float64"])
def test_c3_tile_sorted_plan_matches_the_oracle(env, oracle_mod, val):
    ch, ctx, torch = env
    rows, groups = 64_000_000, 1_000_000
    g = torch.Generator(device="cuda").manual_seed(2)
    k = torch.randint(0, groups, (rows,), dtype=torch.int32, device="cuda", generator=g)
    if val == "int64":
        v = torch.randint(-2**31, 2**31, (rows,), dtype=torch.int64, device="cuda", generator=g)
        vdt = np.int64
    else:
        v = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g)
        vdt = np.float64
    torch.cuda.synchronize()
    kc, vc = ctx.wrap(k.data_ptr(), np.uint32, rows, keepalive=k), ctx.wrap(v.data_ptr(), vdt, rows, keepalive=v)
    aggs = [(ch.AGG_SUM, vdt), (ch.AGG_COUNT, None)]

    def run():
        A = ch.Aggregator(np.uint32, aggs, size_hint=groups, ctx=ctx)
        A.execute_on_block(kc, [vc, None])
        gk, (gs, gc) = A.convert_to_block()
        i = np.argsort(gk)
        return gk[i], gs[i], gc[i]

    gk, gs, gc = run()
    ref, _ = oracle_mod.groupby_pipeline(k.cpu().numpy().view(np.uint32), aggs, [v.cpu().numpy(), None], threads=min(4, _threads()))
    ok, (os_, oc) = ref.convert_to_block()
    j = np.argsort(ok)
    assert np.array_equal(gk, ok[j]) and np.array_equal(gc, oc[j])
    if val == "int64":
        assert np.array_equal(gs, os_[j])                       # wrap-around integer sums: bit-exact
    else:
        assert np.allclose(gs, os_[j], rtol=1e-6, atol=0.0)     # BASELINE.json: 1e-6 relative for sum(Float64)


def test_join_chain_at_ssb_scale_matches_the_oracle(env, oracle_mod):
    """the SSB Q4.1 chain on a 40 M-row sample of the fact keys: supplier / part (2 M keys each: two LDS slices), customer (30 M-key domain:
    the bitmap read from L2 by the alive rows only), date (2556 keys: hash table), against the oracle's joinBlock per join"""
    ch, ctx, torch = env
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ssb
    O = oracle_mod
    C, S, P, rows = 30_000_000, 2_000_000, 2_000_000, 40_000_000
    dims = ssb.gen_dims(C, S, P)
    lo = ssb.gen_lineorder_numpy(rows, C, S, P)
    sides = [("lo_suppkey", dims["s_suppkey"][dims["s_region"] == ssb.AMERICA], ch.JOIN_LEFT, ch.STRICT_SEMI),
             ("lo_partkey", dims["p_partkey"][dims["p_mfgr"] <= 2], ch.JOIN_LEFT, ch.STRICT_SEMI),
             ("lo_custkey", dims["c_custkey"][dims["c_region"] == ssb.AMERICA], ch.JOIN_INNER, ch.STRICT_ALL),
             ("lo_orderdate", dims["d_datekey"], ch.JOIN_INNER, ch.STRICT_ALL)]
    joins, keys, want = [], [], np.ones(rows, dtype=bool)
    want_rid = {}
    for name, build, kind, strict in sides:
        j = ch.HashJoin(kind, strict, key_dtype=np.uint32, ctx=ctx)
        j.add_block(build)
        j.finish_build()
        joins.append(j)
        keys.append(ctx.upload(lo[name]))
        oj = O.HashJoin(kind, strict)
        oj.add_block(build)
        r = oj.probe(lo[name])
        if r["filter"] is not None:
            want &= r["filter"].astype(bool)
        else:
            off = r["offsets"].astype(np.int64)
            m = np.diff(np.concatenate([[0], off])) > 0
            want &= m
            rid = np.full(rows, -1, dtype=np.int64)
            rid[m] = r["added_row"]
            want_rid[name] = rid
    r = ch.join_probe_chain(joins, keys, right_rows=[False, False, True, True], carry=[keys[2]])
    idx = np.flatnonzero(want)
    assert r["kept"] == idx.shape[0] and np.array_equal(r["indexes"].numpy(), idx.astype(np.uint64))
    assert np.array_equal(r["carry"][0].numpy(), lo["lo_custkey"][idx])
    assert np.array_equal(r["right_rowid"][2].numpy().astype(np.int64), want_rid["lo_custkey"][idx])
    assert np.array_equal(r["right_rowid"][3].numpy().astype(np.int64), want_rid["lo_orderdate"][idx])
