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


# ---- fused join -> aggregate, unique build keys: probe keys partitioned twice, table slices staged in LDS ----------------------------
@pytest.mark.parametrize("kind,strict", [("INNER", "ALL"), ("LEFT", "ALL"), ("LEFT", "SEMI"), ("LEFT", "ANTI")])
@pytest.mark.parametrize("nb,blocks", [(400_000, 1), (3_000_000, 2), (3_000_000, 1)])
def test_join_probe_agg_lds_staged_slices_match_numpy(ch, ctx, kind, strict, nb, blocks):
    """unique build keys + integer payload + >= 8 Mi probe rows: k_rp_hist_wide / k_rp_scatter (64 partitions) -> k_rp_tilesort_keys ->
    k_join_probe_lds (8192-cell table slices in LDS).  Tables of 2^21 and 2^24 cells (2 and 16 slices per first-level partition), one and
    two right blocks, the zero key on both sides, probe keys beyond the build key range, every join variant chgpu_join_probe_agg takes."""
    K_ = {"INNER": ch.JOIN_INNER, "LEFT": ch.JOIN_LEFT}[kind]
    S_ = {"ALL": ch.STRICT_ALL, "SEMI": ch.STRICT_SEMI, "ANTI": ch.STRICT_ANTI}[strict]
    rng = np.random.Generator(np.random.PCG64(nb % 97))
    bk = (rng.permutation(nb).astype(np.uint64) * np.uint64(2654435761)) % np.uint64(2**40)   # unique (odd multiplier mod 2^40), contains 0
    assert np.unique(bk).shape[0] == nb
    bv = rng.integers(-2**50, 2**50, size=nb, dtype=np.int64)
    npb = 9_000_017
    pk = np.where(rng.random(npb) < 0.5, bk[rng.integers(0, nb, size=npb)], rng.integers(0, 2**41, size=npb, dtype=np.uint64))
    pk[::100_003] = 0
    j = ch.HashJoin(K_, S_, ctx=ctx)
    cut = nb // blocks
    for b in range(blocks):
        j.add_block(bk[b * cut:(b + 1) * cut if b + 1 < blocks else nb])
    c, s = j.probe_count_sum(ctx.upload(pk), ctx.upload(bv))
    order = np.argsort(bk)
    pos = np.searchsorted(bk[order], pk)
    pos[pos == nb] = 0
    hit = bk[order][pos] == pk
    if strict == "ANTI":
        want_c, want_s = int((~hit).sum()), 0
    else:
        want_c = int(hit.sum()) + (int((~hit).sum()) if (kind == "LEFT" and strict == "ALL") else 0)
        want_s = int(bv[order][pos[hit]].astype(np.uint64).sum(dtype=np.uint64))
    assert c == want_c and s % 2**64 == want_s
    # the first probe of a one-block build side ran without a hash table (join_probe_agg_radix); asking for the key count builds the
    # table, and the same probe then goes through the table's slices (join_probe_agg_lds): same answer
    assert j.n_keys == nb
    c2, s2 = j.probe_count_sum(ctx.upload(pk), ctx.upload(bv))
    assert (c2, s2) == (c, s)


# ---- the unique-key build through LDS-built table slices (join_build_slices) --------------------------------------------------------
@pytest.mark.parametrize("dups", [False, True])
def test_join_slice_build_gives_the_same_table_as_the_generic_build(ch, ctx, dups):
    """one right block of 2.6 M rows without a prefilter: unique keys are partitioned down to 4096-cell slices, built in LDS and written out
    whole (rows whose chain leaves a slice go through an overflow list); a duplicate key anywhere makes the build fall back to the
    generic path.  Either way the ordered joinBlock over the table -- offsets and right row ids -- equals numpy's, the zero key included."""
    rng = np.random.Generator(np.random.PCG64(5 + dups))
    nb = 2_600_000
    bk = (rng.permutation(nb).astype(np.uint64) * np.uint64(2654435761)) % np.uint64(2**40)
    if dups:
        bk[-3:] = bk[:3]
    left = np.concatenate([bk[rng.integers(0, nb, size=300_000)], rng.integers(0, 2**41, size=300_000, dtype=np.uint64), np.zeros(3, dtype=np.uint64)])
    j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, ctx=ctx)
    j.add_block(bk)
    r = j.probe_columns(left)
    assert r["consumed"] == left.shape[0]
    off = r["offsets"].numpy().astype(np.int64)
    rid = r["right_rowid"].numpy()
    counts = np.diff(np.concatenate([[0], off]))
    order = np.argsort(bk, kind="stable")
    lo = np.searchsorted(bk[order], left, side="left")
    hi = np.searchsorted(bk[order], left, side="right")
    assert np.array_equal(counts, hi - lo)
    rows = (rid & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.all((rid >> np.uint64(32)) == 0) and np.array_equal(bk[rows], np.repeat(left, counts))      # every emitted right row carries the left row's key
    if not dups:
        assert r["n_out"] == int(np.isin(left, bk).sum())
    assert j.n_keys == np.unique(bk).shape[0]


@pytest.mark.parametrize("shape", ["dense", "dense_dups", "dense_two_blocks", "beyond_the_bitmap", "signed"])
def test_join_dense_prefilter_for_large_narrow_key_build_sides(ch, ctx, shape):
    """a filtered dimension table of more than 2 Mi rows joined on its 4-byte surrogate key (SSB's customer): the exact bitmap over
    [0, max key] is kept although the build side is beyond the usual prefilter limit -- after a slice build it is filled from the build
    keys, after the generic build (duplicates, two right blocks) by the finalise pass; keys beyond 32 Mi or negative ones get none.  The
    ordered joinBlock and the fused count / sum equal numpy's in every shape, zero key and out-of-range probes included."""
    rng = np.random.Generator(np.random.PCG64(31))
    nb = 2_300_000
    top = {"beyond_the_bitmap": 2**31 - 1, "signed": 2**31 - 1}.get(shape, 20_000_000)
    bk = rng.choice(top, size=nb, replace=False).astype(np.int64)
    dt = np.int32 if shape == "signed" else np.uint32
    if shape == "signed":
        bk[:1000] = -bk[:1000] - 1
    if shape == "dense_dups":
        bk[-4:] = bk[:4]
    bk = bk.astype(dt)
    left = np.concatenate([bk[rng.integers(0, nb, size=400_000)], rng.integers(0, top, size=1_200_000).astype(dt), np.array([0, top, 2**31 - 1], dtype=dt)])
    rng.shuffle(left)
    pay = rng.integers(-2**40, 2**40, size=nb, dtype=np.int64)
    j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, key_dtype=dt, ctx=ctx)
    if shape == "dense_two_blocks":
        j.add_block(bk[:nb // 2])
        j.add_block(bk[nb // 2:])
    else:
        j.add_block(bk)
    r = j.probe_columns(left)
    off = r["offsets"].numpy().astype(np.int64)
    counts = np.diff(np.concatenate([[0], off]))
    order = np.argsort(bk, kind="stable")
    sk = bk[order]
    lo, hi = np.searchsorted(sk, left, side="left"), np.searchsorted(sk, left, side="right")
    assert np.array_equal(counts, hi - lo)
    rid = r["right_rowid"].numpy()
    base = np.array([0, nb // 2], dtype=np.int64) if shape == "dense_two_blocks" else np.array([0], dtype=np.int64)
    rows = base[(rid >> np.uint64(32)).astype(np.int64)] + (rid & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.array_equal(bk[rows], np.repeat(left, counts))
    hit = counts > 0
    want_sum = sum(int(pay[order[a:b]].sum()) for a, b in zip(lo[hit][:2000], hi[hit][:2000]))   # the fused form over the first 2000 matching rows
    c, s_ = j.probe_count_sum(ctx.upload(left[hit][:2000]), ctx.upload(pay))
    assert c == int(counts[hit][:2000].sum()) and s_ % 2**64 == want_sum % 2**64
    assert j.n_keys == np.unique(bk).shape[0]


def test_join_build_phase_contract_with_the_lazy_table(ch, ctx):
    """onBuildPhaseFinish closes the build phase at once although the table is built later: a right block after it is a LOGICAL_ERROR,
    so is one after the first probe; total_rows needs no table, the key count builds it"""
    j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, ctx=ctx)
    j.add_block(np.arange(1, 1001, dtype=np.uint64))
    assert j.total_rows == 1000
    j.finish_build()
    with pytest.raises(ch.ChgpuError) as e:
        j.add_block(np.arange(5, dtype=np.uint64))
    assert e.value.code == ch._capi.ERR_LOGICAL
    assert j.n_keys == 1000
    assert j.probe_count_sum(np.array([1, 5, 5000], dtype=np.uint64), np.arange(1000, dtype=np.int64)) == (2, 0 + 4)
    j2 = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, ctx=ctx)
    j2.add_block(np.arange(1, 11, dtype=np.uint64))
    assert j2.probe_count_sum(np.array([3], dtype=np.uint64), np.arange(10, dtype=np.int64)) == (1, 2)   # no finish_build: the probe ends the phase
    with pytest.raises(ch.ChgpuError):
        j2.add_block(np.arange(5, dtype=np.uint64))


# ---- keys128 / keys256: the device dictionary (chgpu_keydict) under GROUP BY and joins ----------------------------------------------
@pytest.mark.parametrize("arg_dtype", [np.uint32, np.int32, np.float32])
def test_groupby_tile_sorted_plan_widens_narrow_arguments(ch, ctx, arg_dtype):
    """sum(UInt32 / Int32 / Float32 column): the partition pass widens the 4-byte values on the way into LDS (zero / sign extension,
    Float32 -> Float64), the aggregate pass sees 8-byte words; the sums are those of the widened values."""
    rng = np.random.Generator(np.random.PCG64(78))
    n, groups = 6_291_461, 150_000
    k = rng.integers(0, groups, size=n).astype(np.uint32)
    if arg_dtype == np.float32:
        v = (rng.random(n) * 2000 - 1000).astype(np.float32)
    else:
        info = np.iinfo(arg_dtype)
        v = rng.integers(info.min, info.max, size=n, endpoint=True).astype(arg_dtype)
    A = ch.Aggregator(np.uint32, [(ch.AGG_SUM, arg_dtype), (ch.AGG_COUNT, None)], size_hint=groups, ctx=ctx)
    A.execute_on_block(ctx.upload(k), [ctx.upload(v), None])
    gk, (gs, gc) = A.convert_to_block()
    uk, inv = np.unique(k, return_inverse=True)
    order = np.argsort(gk)
    assert np.array_equal(gk[order], uk)
    assert np.array_equal(gc[order], np.bincount(inv, minlength=uk.shape[0]).astype(np.uint64))
    if arg_dtype == np.float32:
        want = np.bincount(inv, weights=v.astype(np.float64), minlength=uk.shape[0])
        assert np.allclose(gs[order], want, rtol=1e-9, atol=1e-6)
    else:
        want = np.zeros(uk.shape[0], dtype=np.uint64)
        np.add.at(want, inv, v.astype(np.int64).astype(np.uint64))
        assert np.array_equal(gs[order].astype(np.uint64), want)


@pytest.mark.parametrize("key_dtype", [np.uint32, np.uint64])
@pytest.mark.parametrize("case", ["more_groups_than_promised", "one_key_half_the_rows", "zero_key_and_ragged_end", "f64_sum"])
def test_groupby_tile_sorted_plan_edge_cases(ch, ctx, key_dtype, case):
    """The two-pass plan (k_rp_tilesort + k_agg_tiles_lds) where its special paths run: LDS tables that overflow because the hint was
    far too small (rows left pending for the finish rounds), runs far longer than a wave (a key with half of all rows), the zero key,
    a row count that ends in the middle of a tile and of a 16-byte pair, Float64 sums (bit-equal from run to run)."""
    rng = np.random.Generator(np.random.PCG64(77))
    n = 7_340_033
    if case == "more_groups_than_promised":
        groups, hint = 3_000_000, 20_000
        k = rng.integers(1, groups, size=n).astype(key_dtype)
    elif case == "one_key_half_the_rows":
        groups, hint = 200_000, 200_000
        k = rng.integers(1, groups, size=n).astype(key_dtype)
        k[rng.random(n) < 0.5] = 123_457
    else:
        groups, hint = 250_000, 250_000
        k = rng.integers(0, groups, size=n).astype(key_dtype)
        k[::97] = 0
    if case == "f64_sum":
        v = rng.random(n) * 1e6 - 5e5
        aggs = [(ch.AGG_SUM, np.float64), (ch.AGG_COUNT, None)]
    else:
        v = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)
        aggs = [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)]
    A = ch.Aggregator(key_dtype, aggs, size_hint=hint, ctx=ctx)
    A.execute_on_block(ctx.upload(k), [ctx.upload(v), None])
    gk, (gs, gc) = A.convert_to_block()
    uk, inv = np.unique(k, return_inverse=True)
    order = np.argsort(gk)
    assert np.array_equal(gk[order], uk)
    assert np.array_equal(gc[order], np.bincount(inv, minlength=uk.shape[0]).astype(np.uint64))
    if case == "f64_sum":
        want = np.bincount(inv, weights=v, minlength=uk.shape[0])
        assert np.allclose(gs[order], want, rtol=1e-9, atol=1e-6)
        # a second run gives the same bits: the sums are kept in fixed point (tests/test_gpu_float_sums.py), no atomic adds doubles
        B = ch.Aggregator(key_dtype, aggs, size_hint=hint, ctx=ctx)
        B.execute_on_block(ctx.upload(k), [ctx.upload(v), None])
        gk2, (gs2, _) = B.convert_to_block()
        assert np.array_equal(gs[order].view(np.uint64), gs2[np.argsort(gk2)].view(np.uint64))
    else:
        want = np.zeros(uk.shape[0], dtype=np.uint64)
        np.add.at(want, inv, v.astype(np.uint64))
        assert np.array_equal(gs[order].astype(np.uint64), want)


@pytest.mark.parametrize("key_dtypes", [(np.uint64, np.uint64), (np.uint64, np.uint32, np.uint16), (np.uint64, np.uint64, np.uint64, np.uint32, np.uint8)])
def test_keys_fixed_group_by_matches_oracle(ch, ctx, oracle_mod, key_dtypes):
    O = oracle_mod
    rng = np.random.Generator(np.random.PCG64(11))
    aggs = [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)]
    G = ch.KeysFixedAggregator(key_dtypes, aggs, ctx=ctx)
    R = O.KeysFixedAggregator(key_dtypes, aggs)
    for n in (70_001, 1, 300_000):                                    # several blocks: ids persist, the table grows
        cols = [rng.integers(0, 40, size=n).astype(d) for d in key_dtypes]
        cols[0][: n // 50] = 0
        for c in cols[1:]:
            c[: n // 50] = 0                                          # the all-zero key
        v = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)
        G.execute_on_block(cols, [v, None])
        R.execute_on_block(cols, [v, None])
    gk, (gs, gc) = G.convert_to_block()
    rk, (rs, rc) = R.convert_to_block()
    assert len(G) == len(rk[0])
    go = np.lexsort([k.astype(np.uint64) for k in gk])
    ro = np.lexsort([k.astype(np.uint64) for k in rk])
    for a, b in zip(gk, rk):
        assert a.dtype == b.dtype and np.array_equal(a[go], b[ro])
    assert np.array_equal(gs[go], rs[ro]) and np.array_equal(gc[go], rc[ro])


def test_keys_fixed_00120_two_key_group_by_on_gpu(ch, ctx, oracle_mod, golden):
    """the reference's 00120_join_and_group_by as what it is: GROUP BY (UInt64, UInt32) -- 12 key bytes, the keys128 method"""
    L = oracle_mod.lib()
    n = np.arange(10, dtype=np.uint64)
    v1 = np.array([L.cho_sql_intHash64(int(x)) for x in n], dtype=np.uint64)
    v2 = np.array([L.cho_sql_intHash32(int(x)) for x in n], dtype=np.uint32)
    A = ch.KeysFixedAggregator([np.uint64, np.uint32], [(ch.AGG_SUM, np.uint64)], ctx=ctx)
    A.execute_on_block([v1, v2], [n])
    (k1, k2), (s,) = A.convert_to_block()
    rows = sorted([[str(int(x)), str(int(y)), str(int(z))] for x, y, z in zip(k1, k2, s)], key=lambda r: (int(r[0]), int(r[1])))
    assert rows == golden["rows"]["00120_join_and_group_by"]["rows"]


def test_keys_fixed_join_and_selector(ch, ctx, oracle_mod):
    O = oracle_mod
    rng = np.random.Generator(np.random.PCG64(12))
    bk = [rng.integers(0, 300, size=20_000, dtype=np.uint64), rng.integers(0, 5, size=20_000).astype(np.uint32), rng.integers(0, 3, size=20_000).astype(np.uint16)]
    pk = [rng.integers(0, 400, size=50_000, dtype=np.uint64), rng.integers(0, 6, size=50_000).astype(np.uint32), rng.integers(0, 3, size=50_000).astype(np.uint16)]
    bv = rng.integers(-2**40, 2**40, size=20_000, dtype=np.int64)
    j = ch.KeysFixedHashJoin([np.uint64, np.uint32, np.uint16], ch.JOIN_INNER, ch.STRICT_ALL, ctx=ctx)
    j.add_block(bk)
    c, s = j.probe_count_sum(pk, ctx.upload(bv))
    m = O.WideKeyMap(16)
    bid = m.batch(O.pack_fixed(bk, 16), True).astype(np.int64)
    pid = m.batch(O.pack_fixed(pk, 16), False)
    mult = np.bincount(bid, minlength=len(m))
    sums = np.zeros(len(m), dtype=np.uint64)
    np.add.at(sums, bid, bv.astype(np.uint64))
    hit = pid != np.uint64(2**64 - 1)
    assert c == int(mult[pid[hit].astype(np.int64)].sum()) and s % 2**64 == int(sums[pid[hit].astype(np.int64)].sum(dtype=np.uint64))
    # the shard of a wide key by the reference's own hash: UInt128HashCRC32 -> two-level bucket & (shards - 1)
    ids = j.dict.encode(bk, insert=False)
    sel = j.dict.selector(ids, 8).numpy()
    packed = O.pack_fixed(bk, 16)
    want = np.array([((O.hash_keys_fixed(r) >> 24) & 0xFF) & 7 for r in packed[:2000]], dtype=np.uint32)
    assert np.array_equal(sel[:2000], want)


def test_keys_fixed_tag_collisions_are_resolved_exactly(ch, oracle_mod):
    """with 20-bit tags (test hook) dozens of different keys share a tag: the verification rounds must still give exact ids"""
    import os
    import subprocess
    import sys
    code = '''
import numpy as np, clickhouse_amd as ch
ctx = ch.Context(0)
rng = np.random.Generator(np.random.PCG64(5))
a = rng.integers(0, 30000, size=400_000, dtype=np.uint64); b = rng.integers(0, 3, size=400_000, dtype=np.uint64)
d = ch.KeyDict([np.uint64, np.uint64], ctx)
ids = d.encode([a, b]).numpy()
pairs = np.stack([a, b], axis=1)
uniq = np.unique(pairs, axis=0)
assert len(d) == uniq.shape[0], (len(d), uniq.shape[0])
first = {}
for i, (x, y) in enumerate(pairs.tolist()):
    assert first.setdefault(int(ids[i]), (x, y)) == (x, y)
assert len(first) == uniq.shape[0]
k0, k1 = [c.numpy() for c in d.key_columns(ctx.upload(ids))]
assert np.array_equal(k0, a) and np.array_equal(k1, b)
print("ok")
'''
    env = dict(os.environ, CHGPU_TEST_KEYDICT_WEAK_TAGS="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_keys_fixed_dictionary_grows_when_rows_defer_at_its_limit(ch, ctx):
    """the table is sized for the keys it holds, not for the rows of a chunk: 3 M distinct keys into a dictionary made for 1024 run past
    limit = capacity / 2 twice (rows defer, the table grows fourfold, the deferred rows run again); ids stay dense, stable and exact"""
    rng = np.random.Generator(np.random.PCG64(21))
    n = 3_000_000
    a = rng.permutation(n).astype(np.uint64) * np.uint64(2654435761)
    b = (np.arange(n, dtype=np.uint64) * np.uint64(40503)) ^ np.uint64(0xDEADBEEF)
    d = ch.KeyDict([np.uint64, np.uint64], ctx)
    ids = d.encode([a, b]).numpy()
    assert len(d) == n and np.array_equal(np.sort(ids), np.arange(n, dtype=np.uint32))          # dense: every id exactly once
    k0, k1 = [c.numpy() for c in d.key_columns(ctx.upload(ids))]
    assert np.array_equal(k0, a) and np.array_equal(k1, b)
    # a second block: old keys (compared in place, no second kernel), new keys and repeats of the new keys inside the block
    a2 = np.concatenate([a[::7], a[:1000] + np.uint64(1), a[:1000] + np.uint64(1)])
    b2 = np.concatenate([b[::7], b[:1000], b[:1000]])
    ids2 = d.encode([a2, b2]).numpy()
    m = a[::7].shape[0]
    assert np.array_equal(ids2[:m], ids[::7]) and len(d) == n + 1000
    assert np.array_equal(ids2[m:m + 1000], ids2[m + 1000:]) and np.array_equal(np.sort(ids2[m:m + 1000]), np.arange(n, n + 1000, dtype=np.uint32))
    # findKey: present keys keep their ids, absent ones get NO_ID
    probe = d.encode([np.concatenate([a[:500], a[:500] + np.uint64(3)]), np.concatenate([b[:500], b[:500]])], insert=False).numpy()
    assert np.array_equal(probe[:500], ids[:500]) and (probe[500:] == 0xFFFFFFFF).all() and len(d) == n + 1000


# ---- partial states on the wire ------------------------------------------------------------------------------------------------------
def test_state_bytes_match_reference_vectors_and_round_trip(ch, ctx, golden):
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "round2_kat.json")))["agg_states"]
    # avgState(number) over numbers(10): the reference prints 2D000000000000000A (01926_bin_unbin)
    A = ch.Aggregator(None, [(ch.AGG_AVG, np.uint64), (ch.AGG_COUNT, None), (ch.AGG_SUM, np.uint64)], ctx=ctx)
    A.execute_on_block(None, [np.arange(10, dtype=np.uint64), None, np.arange(10, dtype=np.uint64)])
    _, words, rows = A.export_state_columns()
    assert rows == 1 and len(words) == 4                                   # avg numerator, avg denominator, count, sum
    b, off = ch.serialize_states(ctx, ch.AGG_AVG, words[0], words[1])
    assert b.numpy().tobytes().hex().upper() == kat["avgState_numbers10"]["hex"] and off.numpy().tolist() == [0, 9]
    b, _ = ch.serialize_states(ctx, ch.AGG_COUNT, words[2])
    assert b.numpy().tobytes().hex().upper() == kat["countState_10rows"]["hex"]     # countState over 10 rows: 0A (00357)
    b, _ = ch.serialize_states(ctx, ch.AGG_SUM, words[3])
    assert b.numpy().tobytes() == (45).to_bytes(8, "little")                        # the numerator bytes of the avg vector
    three = ctx.upload(np.array([3], dtype=np.uint64))
    assert ch.serialize_states(ctx, ch.AGG_COUNT, three)[0].numpy().tobytes().hex().upper() == kat["countState_3rows"]["hex"]
    # round trip over a two-level export: 256 bucket streams cut out of one buffer, counts across the VarUInt length boundaries
    rng = np.random.Generator(np.random.PCG64(21))
    keys = rng.integers(0, 50_000, size=400_000, dtype=np.uint64)
    vals = rng.integers(-2**62, 2**62, size=400_000, dtype=np.int64)
    G = ch.Aggregator(np.uint64, [(ch.AGG_AVG, np.int64), (ch.AGG_COUNT, None)], ctx=ctx)
    G.execute_on_block(keys, [vals, None])
    kcol, words, groups, bucket_counts = G.export_state_columns_two_level()
    big = ctx.upload(np.array([0, 127, 128, 16383, 16384, 2**32, 2**63, 2**64 - 1], dtype=np.uint64))
    bb, bo = ch.serialize_states(ctx, ch.AGG_COUNT, big)
    assert bo.numpy().tolist() == [0, 1, 2, 4, 6, 9, 14, 24, 34]     # 7 payload bits per byte: 2^63 and 2^64 - 1 take 10 bytes
    back, _ = ch.deserialize_states(ctx, ch.AGG_COUNT, bb, [8])
    assert np.array_equal(back.numpy(), big.numpy())
    ab, ao = ch.serialize_states(ctx, ch.AGG_AVG, words[0], words[1])
    ao_h = ao.numpy()
    starts = np.concatenate([[0], np.cumsum(bucket_counts)[:-1]]).astype(np.int64)
    num, den = ch.deserialize_states(ctx, ch.AGG_AVG, ab, bucket_counts, ao_h[starts].tolist())   # one stream per bucket block
    assert np.array_equal(num.numpy(), words[0].numpy()) and np.array_equal(den.numpy(), words[1].numpy())
    # a CPU-initiator style merge: the deserialised states folded into a fresh aggregator give the same final result
    M = ch.Aggregator(np.uint64, [(ch.AGG_AVG, np.int64), (ch.AGG_COUNT, None)], ctx=ctx)
    cb, _ = ch.serialize_states(ctx, ch.AGG_COUNT, words[2])
    cnt, _ = ch.deserialize_states(ctx, ch.AGG_COUNT, cb, [groups])
    M.merge_states(kcol, [num, den, cnt], groups)
    k1, r1 = G.convert_to_block()
    k2, r2 = M.convert_to_block()
    i, j = np.argsort(k1), np.argsort(k2)
    assert np.array_equal(k1[i], k2[j]) and np.array_equal(r1[0][i], r2[0][j]) and np.array_equal(r1[1][i], r2[1][j])
    trunc = ctx.upload(ab.numpy()[:-1])
    with pytest.raises(ch.ChgpuError):
        ch.deserialize_states(ctx, ch.AGG_AVG, trunc, [groups])


# ---- MergingAggregatedMemoryEfficientTransform on the device (clickhouse_amd/merging.py; block-order logic: tests/test_merging_transform.py) ----
@pytest.mark.parametrize("final", [True, False])
def test_merging_aggregated_transform_over_two_level_exports_and_wire_bytes(ch, ctx, final):
    """three sources: one hands its two-level export over as state word columns, one as serialized states (the wire form, one block per
    bucket), one as a single unsplit block; the merged blocks come out in bucket order, each holding exactly the keys of its bucket,
    and together they equal the aggregation of all rows in one aggregator"""
    rng = np.random.Generator(np.random.PCG64(31))
    aggs = [(ch.AGG_AVG, np.int64), (ch.AGG_COUNT, None), (ch.AGG_SUM, np.int64)]
    srcs = [(rng.integers(0, 60_000, size=300_000, dtype=np.uint64), rng.integers(-2**50, 2**50, size=300_000, dtype=np.int64)) for _ in range(3)]
    t = ch.MergingAggregatedMemoryEfficientTransform(np.uint64, aggs, num_inputs=3, final=final, ctx=ctx)
    exports = []
    for k, v in srcs:
        A = ch.Aggregator(np.uint64, aggs, ctx=ctx)
        A.execute_on_block(k, [v, None, v])
        exports.append(A)
    # source 2: unsplit
    k2, w2, n2 = exports[2].export_state_columns()
    t.add_chunk(2, k2, w2)
    t.finish_input(2)
    # sources 0 and 1: split, interleaved bucket by bucket
    k0, w0, _, c0 = exports[0].export_state_columns_two_level()
    k1, w1, _, c1 = exports[1].export_state_columns_two_level()
    avg_b, avg_o = ch.serialize_states(ctx, ch.AGG_AVG, w1[0], w1[1])
    cnt_b, cnt_o = ch.serialize_states(ctx, ch.AGG_COUNT, w1[2])
    sum_b, sum_o = ch.serialize_states(ctx, ch.AGG_SUM, w1[3])
    avg_o, cnt_o, sum_o = avg_o.numpy(), cnt_o.numpy(), sum_o.numpy()
    out, b0, b1 = [], 0, 0
    for b in range(256):
        if c0[b]:
            t.add_chunk(0, k0.cut(b0, c0[b]), [w.cut(b0, c0[b]) for w in w0], bucket_num=b)
        if c1[b]:
            cut = lambda data, off: data.cut(int(off[b1]), int(off[b1 + c1[b]] - off[b1]))
            t.add_serialized_chunk(1, k1.cut(b1, c1[b]), [cut(avg_b, avg_o), cut(cnt_b, cnt_o), cut(sum_b, sum_o)], bucket_num=b)
        b0 += c0[b]
        b1 += c1[b]
        if b % 64 == 63:
            got = t.pull()
            assert all(x.bucket_num < b for x in got)       # the bucket the sources are AT may still grow
            out += got
    t.finish_input(0)
    t.finish_input(1)
    out += t.pull()
    assert t.pull() == []
    nums = [x.bucket_num for x in out]
    assert nums == sorted(nums) and len(set(nums)) == len(nums) and min(nums) >= 0
    for x in out[::17]:
        assert np.all(ch.hash_to_selector(x.keys, 256).numpy() == x.bucket_num)
    if not final:
        # the not-final output is itself a two-level source: merge it once more into a final single block set
        t2 = ch.MergingAggregatedMemoryEfficientTransform(np.uint64, aggs, num_inputs=1, final=True, ctx=ctx)
        for x in out:
            t2.add_chunk(0, x.keys, x.columns, bucket_num=x.bucket_num)
        t2.finish_input(0)
        out = t2.pull()
    keys = np.concatenate([x.keys.numpy() for x in out])
    cols = [np.concatenate([x.columns[j].numpy() for x in out]) for j in range(3)]
    W = ch.Aggregator(np.uint64, aggs, ctx=ctx)
    W.execute_on_block(np.concatenate([k for k, _ in srcs]), [np.concatenate([v for _, v in srcs]), None, np.concatenate([v for _, v in srcs])])
    wk, wr = W.convert_to_block()
    o, wo = np.argsort(keys), np.argsort(wk)
    assert np.array_equal(keys[o], wk[wo])
    assert np.array_equal(cols[1][o], wr[1][wo]) and np.array_equal(cols[2][o], wr[2][wo])
    assert np.allclose(cols[0][o], wr[0][wo], rtol=1e-12, atol=0)


# ---- round 3: FixedString(N) keys (AggregatedDataVariants::key_fixed_string) ---------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 8, 9, 16, 17, 32])
def test_fixed_string_group_by_matches_oracle(ch, ctx, oracle_mod, n):
    """GROUP BY one FixedString(N): the value's words are the fixed keys (one UInt64 key for N <= 8, keys128 / keys256 beyond); the oracle groups
    the same words (Aggregator key64 / its keys128-256 restatement) and both must return every distinct byte string once"""
    O = oracle_mod
    rng = np.random.Generator(np.random.PCG64(n))
    rows, distinct = 120_000, 3_000
    pool = rng.integers(0, 256, size=(distinct, n), dtype=np.uint8)
    pool[0] = 0                                              # the all-zero value: the zero key of key64 (00134_aggregation_by_fixed_string_of_size_1_2_4_8)
    pool[1, n // 2:] = 0                                     # trailing zero padding belongs to the value
    vals = pool[rng.integers(0, distinct, size=rows)]
    arg = rng.integers(-1000, 1000, size=rows).astype(np.int64)
    aggs = [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)]
    G = ch.FixedStringAggregator(n, aggs, ctx=ctx)
    key = ch.ColumnFixedString.from_numpy(ctx, np.ascontiguousarray(vals).view(f"S{n}").reshape(-1))
    G.execute_on_block(key, [arg, None])
    gk, (gs, gc) = G.convert_to_block()
    nw = (n + 7) // 8
    padded = np.zeros((rows, nw * 8), dtype=np.uint8)
    padded[:, :n] = vals
    words = [np.ascontiguousarray(padded[:, 8 * w:8 * w + 8]).view(np.uint64).reshape(-1) for w in range(nw)]
    if nw == 1:
        R = O.Aggregator(np.uint64, aggs)
        R.execute_on_block(words[0], [arg, None])
        ok, (os_, oc) = R.convert_to_block()
        okeys = ok.view(np.uint8).reshape(-1, 8)[:, :n]
    else:
        R = O.KeysFixedAggregator([np.uint64] * nw, aggs)
        R.execute_on_block(words, [arg, None])
        oks, (os_, oc) = R.convert_to_block()
        okeys = np.concatenate([k.view(np.uint8).reshape(-1, 8) for k in oks], axis=1)[:, :n]
    got = {bytes(k.ljust(n, b"\0")): (int(s), int(c)) for k, s, c in zip(gk.tolist(), gs, gc)}
    want = {bytes(k.tobytes()): (int(s), int(c)) for k, s, c in zip(okeys, os_, oc)}
    assert got == want and len(got) == np.unique(vals, axis=0).shape[0]
    assert gk.dtype == np.dtype(f"S{n}")


def test_fixed_string_reference_rows_00134_and_00128(ch, ctx):
    """00134: GROUP BY materialize(toFixedString('', N)) over one row -> one group of N zero bytes, N = 1..9.
    00128: GROUP BY number, FixedString(3) '   ' over 100000 rows -> 100000 groups; ORDER BY n DESC LIMIT 10 starts at 99999."""
    for n in range(1, 10):
        G = ch.FixedStringAggregator(n, [(ch.AGG_COUNT, None)], ctx=ctx)
        G.execute_on_block(ch.ColumnFixedString.from_numpy(ctx, np.zeros(1, dtype=f"S{n}")), [None])
        gk, (gc,) = G.convert_to_block()
        assert gk.tolist() == [b""] and gk.view(np.uint8).tolist() == [0] * n and gc.tolist() == [1]     # numpy prints S-values without their padding
    numbers = np.arange(100_000, dtype=np.uint64)
    k = np.full(100_000, b"   ", dtype="S3")
    A = ch.KeysFixedAggregator([np.uint64, np.uint64], [(ch.AGG_COUNT, None)], ctx=ctx)          # keys128: number + the FixedString's word
    A.execute_on_block([numbers, ch.ColumnFixedString.from_numpy(ctx, k).words()[0]], [None])
    (kn, kw), (cnt,) = A.convert_to_block()
    assert kn.shape[0] == 100_000 and (cnt == 1).all() and (kw == int.from_bytes(b"   ", "little")).all()
    assert sorted(kn.tolist(), reverse=True)[:5] == [99999, 99998, 99997, 99996, 99995]
    with pytest.raises(ch.ChgpuError) as e:
        ch.FixedStringAggregator(33, [(ch.AGG_COUNT, None)], ctx=ctx)
    assert e.value.code == ch._capi.ERR_NOT_IMPLEMENTED


# ---- round 3: two and more argument words = one tile-sorted pass per word (agg_kernels.hip, agg_localise_desc) -------------------------------
@pytest.mark.parametrize("key_dtype", [np.uint32, np.uint64])
@pytest.mark.parametrize("shape", ["sum_sum_count", "avg_sum", "sum_avg", "avg_avg_count", "f64sum_i32sum_count", "three_sums", "count_sum_sum"])
def test_groupby_several_argument_words_pass_per_word(ch, ctx, oracle_mod, key_dtype, shape):
    """sum(a), sum(b), count() and friends over enough rows and groups for the partitioned strategy: every argument word takes its own
    tile-sorted pass into LDS cells that hold only its state words (numbered locally, mapped back to the table's words); counts ride in the
    first pass.  Against the oracle: integers bit-exact, Float64 sums exact (fixed-point states)."""
    O = oracle_mod
    rng = np.random.Generator(np.random.PCG64(len(shape) + np.dtype(key_dtype).itemsize))
    n, groups = 6_300_001, 400_000
    k = rng.integers(0, groups, size=n).astype(key_dtype)
    k[::101] = 0
    a64 = rng.integers(-2**40, 2**40, size=n, dtype=np.int64)
    b64 = rng.integers(-2**40, 2**40, size=n, dtype=np.int64)
    c64 = rng.integers(0, 2**30, size=n, dtype=np.int64).astype(np.uint64)
    f = (rng.random(n) * 1e4 - 5e3)
    i32 = rng.integers(-2**31, 2**31, size=n, dtype=np.int64).astype(np.int32)
    aggs, args = {
        "sum_sum_count": ([(ch.AGG_SUM, np.int64), (ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], [a64, b64, None]),
        "avg_sum": ([(ch.AGG_AVG, np.int64), (ch.AGG_SUM, np.int64)], [a64, b64]),
        "sum_avg": ([(ch.AGG_SUM, np.int64), (ch.AGG_AVG, np.int64)], [a64, b64]),
        "avg_avg_count": ([(ch.AGG_AVG, np.int64), (ch.AGG_AVG, np.uint64), (ch.AGG_COUNT, None)], [a64, c64, None]),
        "f64sum_i32sum_count": ([(ch.AGG_SUM, np.float64), (ch.AGG_SUM, np.int32), (ch.AGG_COUNT, None)], [f, i32, None]),
        "three_sums": ([(ch.AGG_SUM, np.int64), (ch.AGG_SUM, np.uint64), (ch.AGG_SUM, np.int64)], [a64, c64, b64]),
        "count_sum_sum": ([(ch.AGG_COUNT, None), (ch.AGG_SUM, np.int64), (ch.AGG_SUM, np.int64)], [None, a64, b64]),
    }[shape]
    G = ch.Aggregator(key_dtype, aggs, size_hint=groups, ctx=ctx)
    R = O.Aggregator(key_dtype, aggs)
    for lo, hi in ((0, n - 1_000_000), (n - 1_000_000, n)):           # a big block (the passes) and a small one (another strategy) into the same table
        G.execute_on_block(ctx.upload(k[lo:hi]), [ctx.upload(x[lo:hi]) if x is not None else None for x in args])
        R.execute_on_block(k[lo:hi], [x[lo:hi] if x is not None else None for x in args])
    gk, gres = G.convert_to_block()
    ok, ores = R.convert_to_block()
    i, j = np.argsort(gk), np.argsort(ok)
    assert np.array_equal(gk[i], ok[j])
    for (kind, t), g, o in zip(aggs, gres, ores):
        if g.dtype.kind == "f":
            assert np.allclose(g[i], o[j], rtol=1e-9, atol=0)
        else:
            assert np.array_equal(g[i], o[j])
