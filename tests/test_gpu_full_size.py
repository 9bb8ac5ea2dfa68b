"""BASELINE.json's FULL sizes on one MI355X, checked through size-independent properties (the oracle would take minutes here):
C2 1e9-row filter+sum, C3 1e9-row GROUP BY with 1e6 groups, C4 1e8-row probe against a 1e7-row build."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THR = 214748365


@pytest.fixture(scope="module")
def env():
    import torch

    import clickhouse_amd as ch
    ctx = ch.Context(0)
    yield ch, ctx, torch
    ctx.trim()
    ctx.close()


def test_c2_filter_sum_one_billion_rows(env):
    ch, ctx, torch = env
    n = 1_000_000_000
    g = torch.Generator(device="cuda").manual_seed(1)
    t = torch.randint(0, 2**31, (n,), dtype=torch.int64, device="cuda", generator=g)
    torch.cuda.synchronize()
    col = ctx.wrap(t.data_ptr(), np.int64, n, keepalive=t)
    s_lt, c_lt = ch.filter_sum(col, ch.LT, THR)
    s_ge, c_ge = ch.filter_sum(col, ch.GE, THR)
    total = int(ch.sum_add_many(col)[0])
    assert c_lt + c_ge == n and abs(c_lt / n - 0.1) < 1e-3                 # count(p) + count(not p) = n ; ~10 % pass
    assert int(s_lt) + int(s_ge) == total == int(t.sum().item())           # linearity, independent checksum
    third = n // 3 + 7
    parts = [int(ch.sum_add_many(col, lo, hi)[0]) for lo, hi in ((0, third), (third, 2 * third), (2 * third, n))]
    assert sum(parts) == total                                              # checksum of checksums over a ragged split
    mask = ch.cmp_const(col, ch.LT, THR)
    assert ch.count_bytes_in_filter(mask) == c_lt
    kept = col.filter(mask)
    assert kept.size() == c_lt and int(ch.sum_add_many(kept)[0]) == int(s_lt)
    again = kept.filter(ch.cmp_const(kept, ch.LT, THR))                      # idempotence: filtering the filtered column keeps all
    assert again.size() == c_lt
    assert np.array_equal(kept.numpy(10_000_000), t[t < THR][:10_000_000].cpu().numpy())   # order preserved (prefix compared)
    del t


def test_c3_group_by_one_billion_rows_one_million_groups(env):
    ch, ctx, torch = env
    n, groups = 1_000_000_000, 1_000_000
    g = torch.Generator(device="cuda").manual_seed(2)
    k = torch.randint(0, groups, (n,), dtype=torch.int32, device="cuda", generator=g)
    v = torch.randint(-2**31, 2**31, (n,), dtype=torch.int64, device="cuda", generator=g)
    torch.cuda.synchronize()
    A = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], size_hint=groups, ctx=ctx)
    A.execute_on_block(ctx.wrap(k.data_ptr(), np.uint32, n, keepalive=k), [ctx.wrap(v.data_ptr(), np.int64, n, keepalive=v), None])
    keys, (sums, counts) = A.convert_to_block()
    assert keys.shape[0] == groups and np.array_equal(np.sort(keys), np.arange(groups, dtype=np.uint32))   # every key exactly once
    assert int(counts.sum()) == n                                                                           # counts partition the rows
    assert int(sums.sum()) == int(v.sum().item())                                                           # checksum of checksums
    # spot-check 5 groups against an independent selection
    for key in (0, 1, 499_999, 777_777, groups - 1):
        sel = k == key
        i = int(np.nonzero(keys == key)[0][0])
        assert int(counts[i]) == int(sel.sum().item()) and int(sums[i]) == int(v[sel].sum().item())
    del k, v


def test_c4_join_hundred_million_probe_ten_million_build(env):
    ch, ctx, torch = env
    nb, npb = 10_000_000, 100_000_000
    g = torch.Generator(device="cuda").manual_seed(5)
    bk = (torch.randperm(nb, device="cuda", generator=g).to(torch.int64) + 1) * 2654435761      # unique build keys
    bv = torch.randint(-2**40, 2**40, (nb,), dtype=torch.int64, device="cuda", generator=g)
    hit = torch.rand(npb, device="cuda", generator=g) < 0.5
    pk = torch.where(hit, bk[torch.randint(0, nb, (npb,), device="cuda", generator=g)], torch.randint(0, 2**62, (npb,), dtype=torch.int64, device="cuda", generator=g) * 2 + 1)
    torch.cuda.synchronize()
    bkc, pkc = ctx.wrap(bk.data_ptr(), np.uint64, nb, keepalive=bk), ctx.wrap(pk.data_ptr(), np.uint64, npb, keepalive=pk)
    bvc = ctx.wrap(bv.data_ptr(), np.int64, nb, keepalive=bv)
    j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, ctx=ctx)
    j.add_block(bkc)
    assert j.n_keys == nb
    r = j.probe_columns(pkc)
    assert r["consumed"] == npb
    offs = r["offsets"]
    # unique build keys: INNER ALL == SEMI; matches == probe keys present in the build side
    semi = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_SEMI, ctx=ctx)
    semi.add_block(bkc)
    rs = semi.probe_columns(pkc)
    assert ch.count_bytes_in_filter(rs["filter"]) == r["n_out"]
    anti = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_ANTI, ctx=ctx)
    anti.add_block(bkc)
    assert ch.count_bytes_in_filter(anti.probe_columns(pkc)["filter"]) == npb - r["n_out"]     # semi + anti partition the probe side
    left = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_ALL, ctx=ctx)
    left.add_block(bkc)
    assert left.probe_columns(pkc)["n_out"] == npb                                              # LEFT keeps every probe row once
    # round trip: the joined build keys equal the replicated probe keys, row for row
    joined_bk = bkc.index(r["right_rowid"])
    joined_pk = pkc.replicate(offs)
    eq = ch.filter_sum(joined_bk, ch.GE, 0, scalar_tag=ch.U64)
    s_bk, s_pk = ch.sum_add_many(joined_bk)[0], ch.sum_add_many(joined_pk)[0]
    assert int(s_bk) == int(s_pk) and eq[1] == r["n_out"]
    a, b = joined_bk.numpy(5_000_000), joined_pk.numpy(5_000_000)
    assert np.array_equal(a, b)
    payload = bvc.index(r["right_rowid"])
    sorted_bk, perm = torch.sort(bk)
    pos = torch.searchsorted(sorted_bk, pk).clamp(max=nb - 1)
    found = sorted_bk[pos] == pk
    assert int(found.sum().item()) == r["n_out"]
    assert int(ch.sum_add_many(payload)[0]) == int(bv[perm[pos[found]]].sum().item())          # independent payload checksum
    del bk, bv, pk


def test_block_of_2_pow_32_rows_is_rejected_like_the_reference(env):
    ch, ctx, torch = env
    big = ctx.alloc(np.uint8, 2**32)                 # HashJoin.cpp:563-564: "Too many rows in right table block"
    j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, key_dtype=np.uint8, ctx=ctx)
    with pytest.raises(ch.ChgpuError) as e:
        j.add_block(big)
    assert e.value.code == ch._capi.ERR_TOO_MANY_ROWS and "Too many rows" in str(e.value)
    big.free()
