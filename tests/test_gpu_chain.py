"""chgpu_join_probe_chain: a chain of filter-form joins answered with one IColumn::Filter, against the oracle's joinBlock run join by
join (the AND of the joins' filters; HashJoinMethodsImpl.h:68-202).  Covers the LDS sweep (dense 4-byte key sets of 1..3 slices), the
tail steps (bitmaps beyond LDS, hash tables, 8-byte keys), null maps, the zero key, ANTI steps, ragged ends and small blocks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle_step_filter(O, kind, strictness, build_keys, probe_keys, null_map=None):
    j = O.HashJoin(kind, strictness)
    j.add_block(build_keys)
    r = j.probe(probe_keys, null_map)
    if r["filter"] is not None:
        return r["filter"].astype(bool)
    if r["offsets"] is None:   # LEFT ANY: every left row is kept
        return np.ones(len(probe_keys), dtype=bool)
    off = r["offsets"].astype(np.int64)
    return np.diff(np.concatenate([[0], off])) > 0


def run_chain(ch, O, ctx, rows, steps, seed):
    """steps: [(kind, strictness, key dtype, key domain, build fraction, with_null_map, with_zero_key)]"""
    rng = np.random.Generator(np.random.PCG64(seed))
    joins, keys, nms, want = [], [], [], np.ones(rows, dtype=bool)
    host = []
    for kind, strictness, dt, domain, frac, with_nm, with_zero in steps:
        lo = 0 if with_zero else 1
        build = np.flatnonzero(rng.random(domain + 1) < frac).astype(dt)
        build = build[build >= lo]
        rng.shuffle(build)
        probe = rng.integers(0, domain + domain // 8 + 2, size=rows).astype(dt)   # some keys beyond the domain, some zero keys
        nm = (rng.random(rows) < 0.05).astype(np.uint8) if with_nm else None
        j = ch.HashJoin(kind, strictness, key_dtype=dt, ctx=ctx)
        j.add_block(build)
        j.finish_build()
        joins.append(j)
        keys.append(ctx.upload(probe))
        nms.append(ctx.upload(nm) if nm is not None else None)
        host.append((kind, strictness, build, probe, nm))
        want &= oracle_step_filter(O, kind, strictness, build, probe, nm)
    carry_host = [rng.integers(0, 2**32, size=rows, dtype=np.uint32), rng.integers(0, 256, size=rows).astype(np.uint8),
                  rng.integers(0, 2**63, size=rows, dtype=np.uint64), rng.integers(0, 2**16, size=rows).astype(np.uint16)]
    r = ch.join_probe_chain(joins, keys, nms if any(m is not None for m in nms) else None, right_rows=[True] * len(joins),
                            carry=[ctx.upload(c) for c in carry_host], want_filter=True)
    got = r["filter"].numpy().astype(bool)
    assert got.shape[0] == rows
    assert r["kept"] == int(want.sum()), (r["kept"], int(want.sum()))
    assert np.array_equal(got, want)
    idx = np.flatnonzero(want)
    assert np.array_equal(r["indexes"].numpy(), idx.astype(np.uint64))            # filterToIndices: ascending row numbers
    for c, h in zip(r["carry"], carry_host):
        assert np.array_equal(c.numpy(), h[idx])                                  # IColumn::index of the left columns
    for s, (kind, strictness, build, probe, nm) in enumerate(host):
        # the matched right rows: the oracle's joinBlock over the surviving rows appends exactly one row per survivor
        j = O.HashJoin(kind, strictness)
        j.add_block(build)
        o = j.probe(probe[idx], nm[idx] if nm is not None else None)
        assert o["added_row"].shape[0] == idx.shape[0]
        want_rid = np.where(o["added_row"] < 0, np.uint64(0xFFFFFFFFFFFFFFFF),
                            (o["added_block"].astype(np.uint64) << np.uint64(32)) | o["added_row"].astype(np.uint64))
        assert np.array_equal(r["right_rowid"][s].numpy(), want_rid), s
    # the same chain with one right column per step gathered inside the call: the column's value at the matched row, the default at a miss
    pay_dt = [np.uint8, np.uint32, np.uint64, np.uint16]
    pays = [((np.arange(h[2].shape[0], dtype=np.uint64) * 2654435761 + 7) % 251).astype(pay_dt[s % 4]) for s, h in enumerate(host)]
    r3 = ch.join_probe_chain(joins, keys, nms if any(m is not None for m in nms) else None, right_rows=[True] * len(joins),
                             right_cols=[ctx.upload(p) for p in pays], want_indexes=False)
    assert r3["kept"] == r["kept"]
    for s in range(len(joins)):
        rid = r["right_rowid"][s].numpy()
        miss = rid == np.uint64(0xFFFFFFFFFFFFFFFF)
        want_pay = np.where(miss, 0, pays[s][np.where(miss, 0, rid & np.uint64(0xFFFFFFFF)).astype(np.int64)]).astype(pays[s].dtype) if rid.shape[0] else pays[s][:0]
        got_pay = r3["right_rowid"][s].numpy()
        assert got_pay.dtype == pays[s].dtype and np.array_equal(got_pay, want_pay), s
    # the same chain asked for nothing but the count
    r2 = ch.join_probe_chain(joins, keys, nms if any(m is not None for m in nms) else None, want_indexes=False)
    assert r2["kept"] == r["kept"] and r2["indexes"] is None and r2["filter"] is None
    return r["kept"]


@pytest.mark.parametrize("rows", [3 * 65536 + 12345, 2_000_003])
def test_chain_lds_steps_and_tail_steps_match_the_oracle(oracle_mod, rows):
    import clickhouse_amd as ch
    O = oracle_mod
    ctx = ch.Context(0)
    steps = [
        (ch.JOIN_LEFT, ch.STRICT_SEMI, np.uint32, 20_000, 0.3, False, False),      # one LDS slice
        (ch.JOIN_LEFT, ch.STRICT_SEMI, np.uint32, 2_400_000, 0.5, False, True),    # three LDS slices, zero key present
        (ch.JOIN_INNER, ch.STRICT_ALL, np.uint32, 6_000_000, 0.6, False, False),   # dense bitmap beyond LDS: tail, from L2
        (ch.JOIN_INNER, ch.STRICT_ALL, np.uint64, 5_000, 0.9, False, False),       # 8-byte keys: tail, hash table
    ]
    kept = run_chain(ch, O, ctx, rows, steps, seed=rows)
    assert kept > 0


def test_chain_anti_steps_null_maps_and_zero_keys(oracle_mod):
    import clickhouse_amd as ch
    O = oracle_mod
    ctx = ch.Context(0)
    rows = 1_500_001
    steps = [
        (ch.JOIN_LEFT, ch.STRICT_ANTI, np.uint32, 1_300_000, 0.2, True, True),     # LDS, two slices, null map, zero key, inverted
        (ch.JOIN_LEFT, ch.STRICT_SEMI, np.int32, 50_000, 0.7, True, False),        # LDS, signed keys
        (ch.JOIN_LEFT, ch.STRICT_ANTI, np.uint64, 100_000, 0.3, True, True),       # tail, hash table, inverted
        (ch.JOIN_LEFT, ch.STRICT_ANY, np.uint32, 1_000, 0.5, False, False),        # LEFT ANY keeps every row: skipped
    ]
    run_chain(ch, O, ctx, rows, steps, seed=7)


@pytest.mark.parametrize("rows", [0, 1, 3, 4, 1000, 65536 * 2 + 1])
def test_chain_small_blocks_take_the_generic_path(oracle_mod, rows):
    import clickhouse_amd as ch
    O = oracle_mod
    ctx = ch.Context(0)
    steps = [
        (ch.JOIN_LEFT, ch.STRICT_SEMI, np.uint32, 3_000, 0.5, False, True),
        (ch.JOIN_INNER, ch.STRICT_ALL, np.uint16, 500, 0.8, True, False),
        (ch.JOIN_LEFT, ch.STRICT_ANTI, np.uint8, 200, 0.1, False, False),
    ]
    run_chain(ch, O, ctx, rows, steps, seed=rows + 1)


def test_chain_rejects_what_is_not_a_filter(oracle_mod):
    import clickhouse_amd as ch
    ctx = ch.Context(0)
    k = ctx.upload(np.arange(10, dtype=np.uint32))
    dup = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, key_dtype=np.uint32, ctx=ctx)
    dup.add_block(np.array([1, 2, 2, 3], dtype=np.uint32))
    dup.finish_build()
    with pytest.raises(ch.ChgpuError) as e:   # ALL over duplicate keys replicates left rows
        ch.join_probe_chain([dup], [k])
    assert e.value.code == ch._capi.ERR_NOT_IMPLEMENTED
    anyj = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ANY, key_dtype=np.uint32, ctx=ctx)
    anyj.add_block(np.array([1, 2, 3], dtype=np.uint32))
    anyj.finish_build()
    with pytest.raises(ch.ChgpuError) as e:   # setUsedOnce: stateful
        ch.join_probe_chain([anyj], [k])
    assert e.value.code == ch._capi.ERR_NOT_IMPLEMENTED
    ok = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_SEMI, key_dtype=np.uint32, ctx=ctx)
    ok.add_block(np.array([1, 2, 3], dtype=np.uint32))
    ok.finish_build()
    with pytest.raises(ch.ChgpuError) as e:
        ch.join_probe_chain([ok, ok], [k, ctx.upload(np.arange(11, dtype=np.uint32))])
    assert e.value.code == ch._capi.ERR_SIZES_MISMATCH
    r = ch.join_probe_chain([ok], [k], want_filter=True)
    assert r["kept"] == 3 and r["filter"].numpy().tolist() == [0, 1, 1, 1, 0, 0, 0, 0, 0, 0] and r["indexes"].numpy().tolist() == [1, 2, 3]
