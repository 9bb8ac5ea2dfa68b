"""GPU parity of the round-2 additions: the fused join -> aggregate entry (chgpu_join_probe_agg), checked against the CPU oracle's
joinBlock + payload gather + sum over the same inputs."""
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


def _oracle_count_sum(O, kind, strict, right_blocks, payload_blocks, left, lnull, dtype):
    """count() and sum(payload) of the rows joinBlock emits, from the oracle's row pairs (default rows add the type default 0)"""
    o = O.HashJoin(kind, strict)
    for keys, nm, jm in right_blocks:
        o.add_block(keys, null_map=nm, join_mask=jm)
    _, blk, row, consumed = o.joined_pairs(left, lnull)
    assert consumed == left.shape[0]
    hit = blk >= 0
    vals = np.zeros(blk.shape[0], dtype=dtype)
    for b, p in enumerate(payload_blocks):
        m = hit & (blk == b)
        vals[m] = p[row[m]]
    if np.dtype(dtype).kind == "f":
        return blk.shape[0], float(vals.astype(np.float64).sum())
    acc = vals.astype(np.int64 if np.dtype(dtype).kind == "i" else np.uint64)
    return blk.shape[0], int(acc.sum(dtype=acc.dtype))  # wraps like AggregateFunctionSum (Sum.h:36-39)


@pytest.mark.parametrize("kind,strict", [("INNER", "ALL"), ("LEFT", "ALL"), ("LEFT", "ANY"), ("LEFT", "SEMI"), ("LEFT", "ANTI")])
@pytest.mark.parametrize("pdtype", [np.int64, np.uint32, np.int16, np.float64])
def test_join_probe_agg_matches_oracle(ch, ctx, oracle_mod, kind, strict, pdtype):
    O = oracle_mod
    K_ = {"INNER": ch.JOIN_INNER, "LEFT": ch.JOIN_LEFT}[kind]
    S_ = {"ALL": ch.STRICT_ALL, "ANY": ch.STRICT_ANY, "SEMI": ch.STRICT_SEMI, "ANTI": ch.STRICT_ANTI}[strict]
    rng = np.random.Generator(np.random.PCG64(77))
    rb = [rng.integers(0, 5000, size=n, dtype=np.uint64) for n in (20_000, 1, 30_001)]  # duplicates, key 0, three right blocks
    rnull = (rng.integers(0, 50, size=20_000) == 0).astype(np.uint8)
    rmask = (rng.integers(0, 20, size=30_001) != 0).astype(np.uint8)
    left = rng.integers(0, 10_000, size=70_003, dtype=np.uint64)
    lnull = (rng.integers(0, 40, size=left.shape[0]) == 0).astype(np.uint8)
    if np.dtype(pdtype).kind == "f":
        pay = [rng.random(k.shape[0]) for k in rb]
    else:
        info = np.iinfo(pdtype)
        pay = [rng.integers(info.min, info.max, size=k.shape[0], dtype=pdtype, endpoint=True) for k in rb]
    j = ch.HashJoin(K_, S_, ctx=ctx)
    j.add_block(rb[0], null_map=rnull)
    j.add_block(rb[1])
    j.add_block(rb[2], join_mask=rmask)
    payload = ctx.upload(np.concatenate(pay))
    want_c, want_s = _oracle_count_sum(O, K_, S_, [(rb[0], rnull, None), (rb[1], None, None), (rb[2], None, rmask)], pay, left, lnull, pdtype)
    for nm in (lnull, None):
        if nm is None:
            want_c, want_s = _oracle_count_sum(O, K_, S_, [(rb[0], rnull, None), (rb[1], None, None), (rb[2], None, rmask)], pay, left, None, pdtype)
        c, s = j.probe_count_sum(left, payload, null_map=nm)
        assert c == want_c
        if np.dtype(pdtype).kind == "f":
            assert abs(s - want_s) <= 1e-6 * max(1.0, abs(want_s))  # BASELINE: 1e-6 relative for sum(Float64)
            c2, s2 = j.probe_count_sum(left, payload, null_map=nm)
            assert (c2, s2) == (c, s)                                 # fixed reduction order: reproducible bit for bit
        else:
            assert s == want_s
        c0, s0 = j.probe_count_sum(left, None, null_map=nm)           # count only
        assert c0 == want_c and s0 is None


def test_join_probe_agg_refuses_stateful_variants(ch, ctx):
    j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ANY, ctx=ctx)
    j.add_block(np.arange(10, dtype=np.uint64))
    with pytest.raises(ch.ChgpuError) as e:
        j.probe_count_sum(np.arange(5, dtype=np.uint64), np.arange(10, dtype=np.int64))
    assert e.value.code == ch._capi.ERR_NOT_IMPLEMENTED
    j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, ctx=ctx)
    j.add_block(np.arange(10, dtype=np.uint64))
    with pytest.raises(ch.ChgpuError) as e:
        j.probe_count_sum(np.arange(5, dtype=np.uint64), np.arange(9, dtype=np.int64))  # payload shorter than the right side
    assert e.value.code == ch._capi.ERR_SIZES_MISMATCH
    assert j.probe_count_sum(np.zeros(0, dtype=np.uint64), np.arange(10, dtype=np.int64)) == (0, 0)
