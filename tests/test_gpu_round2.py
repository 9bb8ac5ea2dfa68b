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


# ---- the RCCL exchange through the C ABI at world size 1 (one GPU per rank is all a test box has; world 2 runs the same orchestration over
#      gloo in tests/test_gpu_distributed.py and tests/test_distributed_gloo.py) -----------------------------------------------------------
def test_rccl_comm_world1_and_sharded_operators_match_single_gpu(ch, ctx):
    from clickhouse_amd import distributed as D
    comm = D.Comm(ctx, 0, 1, D.Comm.unique_id())
    try:
        assert comm.all_to_all_counts([12345]) == [12345]
        src = ctx.upload(np.arange(1000, dtype=np.int64) * 3)
        got = comm.all_to_all(src, [1000], [1000])
        assert np.array_equal(got.numpy(), np.arange(1000, dtype=np.int64) * 3)
        assert comm.all_reduce_u64([5, 2**64 - 1]) == [5, 2**64 - 1]
        col = ctx.upload(np.array([1, 2, 3], dtype=np.uint64))
        comm.all_reduce_column(col)
        assert col.numpy().tolist() == [1, 2, 3]
        comm.barrier()
        with pytest.raises(ch.ChgpuError) as e:
            comm.all_to_all(src, [999], [999])                      # counts must add up to the column
        assert e.value.code == ch._capi.ERR_SIZES_MISMATCH
        eng = D.LocalEngine(ctx, comm)
        rng = np.random.Generator(np.random.PCG64(9))
        k = rng.integers(0, 70_000, size=500_000, dtype=np.uint32)
        v = rng.integers(-2**62, 2**62, size=500_000, dtype=np.int64)
        aggs = [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)]
        sg = D.ShardedGroupBy(eng, np.uint32, aggs)
        sg.add_block(ctx.upload(k), [ctx.upload(v), None])
        gk, (gs, gc) = sg.finish()
        one = ch.Aggregator(np.uint32, aggs, ctx=ctx)
        one.execute_on_block(k, [v, None])
        ok, (os_, oc) = one.convert_to_block()
        i, j = np.argsort(gk), np.argsort(ok)
        assert np.array_equal(gk[i], ok[j]) and np.array_equal(gs[i], os_[j]) and np.array_equal(gc[i], oc[j])
        bk = rng.permutation(100_000).astype(np.uint64) * np.uint64(7) + np.uint64(1)
        bv = rng.integers(-2**40, 2**40, size=100_000, dtype=np.int64)
        pk = rng.integers(0, 800_000, size=300_000, dtype=np.uint64)
        sj = D.ShardedHashJoin(eng, ch.JOIN_INNER, ch.STRICT_ALL)
        sj.add_build_rows(ctx.upload(bk), [ctx.upload(bv)])
        cnt, sm = sj.probe_count_sum(ctx.upload(pk), 0)
        hit = np.isin(pk, bk)
        lut = dict(zip(bk.tolist(), bv.tolist()))
        assert cnt == int(hit.sum()) and sm == sum(lut[x] for x in pk[hit].tolist()) % 2**64
        n_out, left, right = sj.probe(ctx.upload(pk), [])
        assert n_out == cnt and np.array_equal(right[0].numpy(), np.array([lut[x] for x in left[0].numpy().tolist()], dtype=np.int64))
    finally:
        comm.close()


def test_context_outlives_nothing_children_keep_it_alive(ch):
    """chgpu_ctx_destroy with live children only marks the context; the last child tears it down (no use-after-free in any order)"""
    c = ch.Context(0)
    col = c.upload(np.arange(10, dtype=np.int64))
    agg = ch.Aggregator(np.uint32, [(ch.AGG_COUNT, None)], ctx=c)
    agg.execute_on_block(np.arange(5, dtype=np.uint32), [None])
    c.close()                                   # children still alive
    assert col.numpy().tolist() == list(range(10))
    assert len(agg) == 5
    agg.close()
    col.free()                                  # last reference: the context goes now


def test_pinned_async_upload_overlaps_and_orders(ch, ctx):
    import ctypes as C
    K = ch._capi
    n = 1 << 20
    host = C.c_void_p()
    K.check(K.lib().chgpu_host_alloc(n * 8, C.byref(host)))
    try:
        arr = np.ctypeslib.as_array(C.cast(host, C.POINTER(C.c_int64)), shape=(n,))
        total = 0
        tickets = []
        for rep in range(4):
            if tickets:
                K.check(K.lib().chgpu_upload_wait(ctx._h, tickets[-1]))   # the buffer is refilled only after its upload finished
            arr[:] = np.arange(n, dtype=np.int64) + rep
            h, t = C.c_void_p(), C.c_uint64(0)
            K.check(K.lib().chgpu_col_upload_async(ctx._h, K.I64, host, n, C.byref(h), C.byref(t)))
            col = ch.Column(ctx, h)
            s, c = ch.filter_sum(col, ch.GE, 0)                            # launched right behind the copy: must see the uploaded rows
            assert (int(s), c) == (int(arr.sum()), n)
            tickets.append(t.value)
            total += 1
        assert tickets == sorted(tickets) and len(set(tickets)) == 4
        assert K.lib().chgpu_upload_wait(ctx._h, 10**9) == K.ERR_BAD_ARGUMENTS
    finally:
        K.check(K.lib().chgpu_host_free(host))


# ---- GROUP BY partitioned path: carried-tail scatter, 32-bit partition hash, compile-time state updates -----------------------------
@pytest.mark.parametrize("key_dtype,pattern", [(np.uint32, "uniform"), (np.uint32, "stride"), (np.int32, "zipf"), (np.uint64, "uniform"), (np.uint16, "uniform")])
@pytest.mark.parametrize("aggs_name", ["sum_count", "count_sum", "sum", "avg_f64", "sum_sum_count"])
def test_groupby_partitioned_shapes_match_numpy(ch, ctx, key_dtype, pattern, aggs_name):
    rng = np.random.Generator(np.random.PCG64(2024))
    n = 6_000_011   # ragged: not a multiple of any tile; >= 4 Mi rows so the partitioned strategy is taken
    groups = 300_000
    if pattern == "uniform":
        k = rng.integers(0, min(groups, np.iinfo(key_dtype).max), size=n).astype(key_dtype)
    elif pattern == "stride":
        k = (rng.integers(0, groups, size=n).astype(np.uint64) * np.uint64(4096)).astype(key_dtype)  # keys differ in high bits only (+ key 0)
    else:
        k = np.minimum(rng.zipf(1.2, size=n), groups).astype(key_dtype)                                 # a few very hot keys
    v = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)
    w = rng.integers(-2**31, 2**31, size=n, dtype=np.int64)
    f = rng.random(n)
    spec = {"sum_count": [(ch.AGG_SUM, np.int64, v), (ch.AGG_COUNT, None, None)],
            "count_sum": [(ch.AGG_COUNT, None, None), (ch.AGG_SUM, np.int64, v)],
            "sum": [(ch.AGG_SUM, np.int64, v)],
            "avg_f64": [(ch.AGG_AVG, np.float64, f)],
            "sum_sum_count": [(ch.AGG_SUM, np.int64, v), (ch.AGG_SUM, np.int64, w), (ch.AGG_COUNT, None, None)]}[aggs_name]
    A = ch.Aggregator(key_dtype, [(kind, dt) for kind, dt, _ in spec], size_hint=groups, ctx=ctx)
    cols = {id(a): ctx.upload(a) for _, _, a in spec if a is not None}
    A.execute_on_block(ctx.upload(k), [cols[id(a)] if a is not None else None for _, _, a in spec])
    gk, res = A.convert_to_block()
    uk, inv = np.unique(k, return_inverse=True)
    order = np.argsort(gk)
    assert np.array_equal(gk[order], uk)
    cnt = np.bincount(inv, minlength=uk.shape[0])
    for (kind, dt, a), r in zip(spec, res):
        if kind == ch.AGG_COUNT:
            assert np.array_equal(r[order], cnt.astype(np.uint64))
        elif kind == ch.AGG_SUM:
            want = np.zeros(uk.shape[0], dtype=np.uint64)
            np.add.at(want, inv, a.astype(np.uint64))               # wraps modulo 2^64 like AggregateFunctionSum
            assert np.array_equal(r[order].view(np.uint64), want)
        else:
            want = np.bincount(inv, weights=a, minlength=uk.shape[0]) / cnt
            assert np.allclose(r[order], want, rtol=1e-6, atol=0)    # BASELINE: 1e-6 relative for avg(Float64)


# ---- fused join -> aggregate over a build side far beyond L2: probe keys partitioned by table region --------------------------------
@pytest.mark.parametrize("kind", ["INNER", "LEFT"])
def test_join_probe_agg_region_partitioned_matches_numpy(ch, ctx, kind):
    rng = np.random.Generator(np.random.PCG64(99))
    nb, npb, space = 3_200_000, 6_000_003, 6_000_000      # table 2^23 cells x 16 B = 128 MB: the region plan applies
    bk = rng.integers(0, space // 2, size=nb, dtype=np.uint64)          # duplicates (MULTI cells) and the zero key
    bv = rng.integers(-2**50, 2**50, size=nb, dtype=np.int64)
    pk = rng.integers(0, space, size=npb, dtype=np.uint64)
    j = ch.HashJoin(ch.JOIN_INNER if kind == "INNER" else ch.JOIN_LEFT, ch.STRICT_ALL, ctx=ctx)
    j.add_block(bk[:1_000_000])
    j.add_block(bk[1_000_000:])                                         # two right blocks: flat row ordinals
    c, s = j.probe_count_sum(ctx.upload(pk), ctx.upload(bv))
    mult = np.bincount(bk.astype(np.int64), minlength=space)
    sums = np.zeros(space, dtype=np.uint64)
    np.add.at(sums, bk.astype(np.int64), bv.astype(np.uint64))
    m = mult[pk.astype(np.int64)]
    want_c = int(m.sum()) + (int((m == 0).sum()) if kind == "LEFT" else 0)
    want_s = int(sums[pk.astype(np.int64)].sum(dtype=np.uint64))
    assert c == want_c and s % 2**64 == want_s
    c2, s2 = j.probe_count_sum(ctx.upload(pk[:1000]), ctx.upload(bv))   # a small probe takes the one-pass kernel: same answers
    m2 = mult[pk[:1000].astype(np.int64)]
    assert c2 == int(m2.sum()) + (int((m2 == 0).sum()) if kind == "LEFT" else 0)
