"""SSB Q4.1-style query (BASELINE.json configs[4]) composed from the hot-path operators, once over the C ABI (GPU) and once
over the oracle (CPU restatement, per Block) — used by tests/test_gpu_ssb.py and tools/bench_ssb.py.

    SELECT d_year, c_nation, sum(lo_revenue - lo_supplycost) AS profit
    FROM lineorder JOIN supplier ON lo_suppkey = s_suppkey JOIN part ON lo_partkey = p_partkey
                   JOIN customer ON lo_custkey = c_custkey JOIN date ON lo_orderdate = d_datekey
    WHERE c_region = AMERICA AND s_region = AMERICA AND p_mfgr <= 2          -- MFGR#1 or MFGR#2, dictionary-encoded
    GROUP BY d_year, c_nation

Plan (the reference's shape: filtered dimension tables become the right sides of hash joins, the fact table streams through
JoiningTransforms, then AggregatingTransform).  sum(a - b) is evaluated as sum(a) - sum(b) — identical for
wrap-around integer sums — so no arithmetic function is needed yet (expression fusion is SURVEY §8f rank 1).

Default GPU plan (round 3): the four joins are answered as ONE filter over the fact table's key columns (chgpu_join_probe_chain: every join of
Q4.1 is of the filter form -- two semi joins and two INNER ALL joins over unique dimension keys); for the ~1.6 % surviving rows the same
call returns the matched customer / date rows and the gathered revenue / supplycost values.  No fact column is ever copied whole.
SSB_PLAN_PER_JOIN=1 runs the per-operator plan below (one joinBlock + filter / replicate per join), which is what the C++ shim's
JoiningTransform chain does.

Per-operator plan: the two semi joins share ONE filter on the GPU: the supplier join is probed in its ANTI form and its filter becomes the null map of the
part join's key column (a NULL key matches nothing, HashJoinMethodsImpl.h:451-452), so the fact columns are compacted once at 8 % instead of
five columns at 20 % and four more at 40 % of that (C5 11.15 -> 10.2 ms on one box).  SSB_PLAN_TWO_FILTERS=1 runs the reference's shape,
one FilterTransform per JoiningTransform; q41_cpu always does.
"""
import os

import numpy as np

AMERICA = 1
N_DATES = 2556


def gen_dims(customers, suppliers, parts, seed=7):
    rng = np.random.Generator(np.random.PCG64(seed))
    c_nation = rng.integers(0, 25, size=customers).astype(np.uint8)
    dims = dict(
        c_custkey=np.arange(1, customers + 1, dtype=np.uint32), c_nation=c_nation, c_region=(c_nation // 5).astype(np.uint8),
        s_suppkey=np.arange(1, suppliers + 1, dtype=np.uint32), s_region=rng.integers(0, 5, size=suppliers).astype(np.uint8),
        p_partkey=np.arange(1, parts + 1, dtype=np.uint32), p_mfgr=rng.integers(1, 6, size=parts).astype(np.uint8),
        d_datekey=(np.arange(N_DATES, dtype=np.uint32) + 19920101), d_year=(1992 + np.arange(N_DATES) // 366).astype(np.uint32),
    )
    return dims


def gen_lineorder_numpy(rows, customers, suppliers, parts, seed=11):
    rng = np.random.Generator(np.random.PCG64(seed))
    return dict(
        lo_custkey=rng.integers(1, customers + 1, size=rows).astype(np.uint32),
        lo_suppkey=rng.integers(1, suppliers + 1, size=rows).astype(np.uint32),
        lo_partkey=rng.integers(1, parts + 1, size=rows).astype(np.uint32),
        lo_orderdate=(rng.integers(0, N_DATES, size=rows).astype(np.uint32) + 19920101),
        lo_revenue=rng.integers(0, 1_000_000, size=rows).astype(np.uint32),
        lo_supplycost=rng.integers(0, 100_000, size=rows).astype(np.uint32),
    )


def gen_lineorder_torch(rows, customers, suppliers, parts, device, seed=11):
    import torch
    g = torch.Generator(device=device).manual_seed(seed)

    def ri(lo, hi):
        return torch.randint(lo, hi, (rows,), dtype=torch.int32, device=device, generator=g)
    return dict(lo_custkey=ri(1, customers + 1), lo_suppkey=ri(1, suppliers + 1), lo_partkey=ri(1, parts + 1),
                lo_orderdate=ri(0, N_DATES) + 19920101, lo_revenue=ri(0, 1_000_000), lo_supplycost=ri(0, 100_000))


def upload_dims(ctx, dims):
    """Dimension columns made HBM-resident once (what a warm system holds); q41_gpu then skips the PCIe upload."""
    return {k: ctx.upload(v) for k, v in dims.items()}


def q41_gpu(ch, ctx, dims, lo, group_by=None):
    """dims: numpy arrays (uploaded inside the timed plan) or device Columns (see upload_dims); lo: dict of device Columns
    (UInt32).  Returns {(year, nation): (profit, count)}.  group_by: None = this GPU's own Aggregator; else a callable
    (packed key Column, [revenue, supplycost]) -> (keys ndarray, [sum(revenue), sum(supplycost), count] ndarrays) -- q41_sharded passes
    the sharded GROUP BY, whose result is the groups THIS rank owns."""
    up = lambda x: x if isinstance(x, ch.Column) else ctx.upload(x)
    # ---- right sides: filtered dimension tables -> hash tables (FillingRightJoinSideTransform) ----
    c_region, c_custkey, c_nation = up(dims["c_region"]), up(dims["c_custkey"]), up(dims["c_nation"])
    chain = not os.environ.get("SSB_PLAN_PER_JOIN") and not os.environ.get("SSB_PLAN_TWO_FILTERS")
    cm = ch.cmp_const(c_region, ch.EQ, AMERICA)
    sm = ch.cmp_const(up(dims["s_region"]), ch.EQ, AMERICA)
    pm = ch.cmp_const(up(dims["p_mfgr"]), ch.LE, 2)
    j_c = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, key_dtype=np.uint32, ctx=ctx)
    # the supplier join is probed in its ANTI form in the per-join plan: its filter (1 = no supplier of the region) is handed to the part join
    # as the null map of lo_partkey -- a row with a NULL key matches nothing -- so the two semi joins yield ONE filter and the fact columns
    # are compacted once, at 8 %, instead of five columns at 20 % and four more at 40 % of that
    j_s = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_SEMI if (chain or os.environ.get("SSB_PLAN_TWO_FILTERS")) else ch.STRICT_ANTI, key_dtype=np.uint32, ctx=ctx)
    j_p = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_SEMI, key_dtype=np.uint32, ctx=ctx)
    if chain:
        # the dimension predicates as the joins' ON masks (`... JOIN customer ON lo_custkey = c_custkey AND c_region = 'AMERICA'`: rows whose
        # mask is 0 are not inserted, HashJoinMethodsImpl.h:261-272): no filtered copy of a dimension, no row count to wait for, and the
        # right row ids index the dimension's own columns
        j_c.add_block(c_custkey, join_mask=cm)
        j_s.add_block(up(dims["s_suppkey"]), join_mask=sm)
        j_p.add_block(up(dims["p_partkey"]), join_mask=pm)
        cn = c_nation
    else:
        ck, cn = ch.filter_columns([c_custkey, c_nation], cm)
        j_c.add_block(ck)
        j_s.add_block(up(dims["s_suppkey"]).filter(sm))
        j_p.add_block(up(dims["p_partkey"]).filter(pm))
    d_year = up(dims["d_year"])
    j_d = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, key_dtype=np.uint32, ctx=ctx)
    j_d.add_block(up(dims["d_datekey"]))
    for j in (j_c, j_s, j_p, j_d):
        j.finish_build()
    # ---- the fact table through the joins (JoiningTransform x4), most selective first ----
    if chain:
        # (the two right columns the query reads -- c_nation, d_year -- are gathered inside the call: AddedColumns' lazy gather over the survivors)
        r = ch.join_probe_chain([j_s, j_p, j_c, j_d], [lo["lo_suppkey"], lo["lo_partkey"], lo["lo_custkey"], lo["lo_orderdate"]],
                                right_rows=[False, False, True, True], right_cols=[None, None, cn, d_year],
                                carry=[lo["lo_revenue"], lo["lo_supplycost"]], want_indexes=False)
        rev, cost = r["carry"]
        nation, year = r["right_rowid"][2], r["right_rowid"][3]
        return _q41_group_by(ch, ctx, year, nation, rev, cost, group_by)
    r = j_s.probe_columns(lo["lo_suppkey"], need_right_rows=False)   # semi joins: the dimension contributes no column
    f = r["filter"]
    if os.environ.get("SSB_PLAN_TWO_FILTERS"):                       # the reference's shape: one FilterTransform per join
        cust, part, date, rev, cost = ch.filter_columns([lo[k] for k in ("lo_custkey", "lo_partkey", "lo_orderdate", "lo_revenue", "lo_supplycost")], f)
        r = j_p.probe_columns(part, need_right_rows=False)
        f = r["filter"]
        cust, date, rev, cost = ch.filter_columns([cust, date, rev, cost], f)
    else:
        r = j_p.probe_columns(lo["lo_partkey"], null_map=f, need_right_rows=False)
        cust, date, rev, cost = ch.filter_columns([lo[k] for k in ("lo_custkey", "lo_orderdate", "lo_revenue", "lo_supplycost")], r["filter"])
    r = j_c.probe_columns(cust)
    off = r["offsets"]
    date, rev, cost = ch.replicate_columns([date, rev, cost], off)
    nation = cn.index(r["right_rowid"], default_for_missing=True)
    r = j_d.probe_columns(date)
    off = r["offsets"]
    rev, cost, nation = ch.replicate_columns([rev, cost, nation], off)
    year = d_year.index(r["right_rowid"], default_for_missing=True)
    return _q41_group_by(ch, ctx, year, nation, rev, cost, group_by)


Q41_AGGS = lambda M: [(M.AGG_SUM, np.uint32), (M.AGG_SUM, np.uint32), (M.AGG_COUNT, None)]


def _decode_groups(keys, s_rev, s_cost, cnt):
    """packFixed<UInt64>: 4 bytes year, 1 byte nation"""
    keys = np.asarray(keys, dtype=np.uint64)
    return {(int(k & np.uint64(0xFFFFFFFF)), int((k >> np.uint64(32)) & np.uint64(0xFF))): (int(a) - int(b), int(c))
            for k, a, b, c in zip(keys, s_rev, s_cost, cnt)}


def _q41_group_by(ch, ctx, year, nation, rev, cost, group_by=None):
    # ---- GROUP BY d_year, c_nation (keys64: packFixed) ----
    key = ch.pack_fixed_keys([year, nation])
    if group_by is not None:
        keys, (s_rev, s_cost, cnt) = group_by(key, [rev, cost])
        return _decode_groups(keys, s_rev, s_cost, cnt)
    agg = ch.Aggregator(np.uint64, Q41_AGGS(ch), ctx=ctx)
    agg.execute_on_block(key, [rev, cost, None])
    keys, (s_rev, s_cost, cnt) = agg.convert_to_block()   # 35 rows: the packed keys are taken apart on the host
    return _decode_groups(keys, s_rev, s_cost, cnt)


def q41_cpu(O, dims, lo, block_rows=65409, threads=1, make_agg=None):
    """Same plan over the oracle, fact table in Blocks of `block_rows` (lo: dict of numpy arrays).  threads > 1: the reference's pipeline
    shape -- the four right-side tables are built once and shared (joinBlock is concurrent on an immutable table, IJoin.h:92-93), every
    stream pushes its own Blocks through the joins into its own AggregatedDataVariants, the variants are merged at the end.
    make_agg (threads == 1): factory of the aggregation sink (execute_on_block / convert_to_block) -- the sharded plan passes a ShardedGroupBy."""
    import threading
    cm = O.cmp_const(dims["c_region"], O.EQ, AMERICA)
    ck, cn = O.filter_column(dims["c_custkey"], cm), O.filter_column(dims["c_nation"], cm)
    j_c = O.HashJoin(O.JOIN_INNER, O.STRICT_ALL)
    j_c.add_block(ck)
    j_s = O.HashJoin(O.JOIN_LEFT, O.STRICT_SEMI)
    j_s.add_block(O.filter_column(dims["s_suppkey"], O.cmp_const(dims["s_region"], O.EQ, AMERICA)))
    j_p = O.HashJoin(O.JOIN_LEFT, O.STRICT_SEMI)
    j_p.add_block(O.filter_column(dims["p_partkey"], O.cmp_const(dims["p_mfgr"], O.LE, 2)))
    j_d = O.HashJoin(O.JOIN_INNER, O.STRICT_ALL)
    j_d.add_block(dims["d_datekey"])
    aggs_spec = [(O.AGG_SUM, np.uint32), (O.AGG_SUM, np.uint32), (O.AGG_COUNT, None)]
    n = lo["lo_custkey"].shape[0]
    n_blocks = (n + block_rows - 1) // block_rows
    threads = max(1, min(threads, n_blocks))
    aggs = [make_agg() if make_agg is not None else O.Aggregator(np.uint64, aggs_spec) for _ in range(threads)]

    def stream(t):
        agg = aggs[t]
        for bi in range(n_blocks * t // threads, n_blocks * (t + 1) // threads):
            b, e = bi * block_rows, min(n, (bi + 1) * block_rows)
            blk = {k: v[b:e] for k, v in lo.items()}
            f = j_s.probe(blk["lo_suppkey"])["filter"]
            if not f.any():
                continue
            cust, part, date, rev, cost = (O.filter_column(blk[k], f) for k in ("lo_custkey", "lo_partkey", "lo_orderdate", "lo_revenue", "lo_supplycost"))
            f = j_p.probe(part)["filter"]
            if not f.any():
                continue
            cust, date, rev, cost = (O.filter_column(c, f) for c in (cust, date, rev, cost))
            r = j_c.probe(cust)
            off = r["offsets"]
            if off.shape[0] == 0 or off[-1] == 0:
                continue
            date, rev, cost = (O.replicate(c, off) for c in (date, rev, cost))
            nation = cn[r["added_row"]]
            r = j_d.probe(date)
            off = r["offsets"]
            rev, cost, nation = (O.replicate(c, off) for c in (rev, cost, nation))
            year = dims["d_year"][r["added_row"]]
            key = year.astype(np.uint64) | (nation.astype(np.uint64) << np.uint64(32))   # packFixed<UInt64>: 4 bytes year, 1 byte nation
            agg.execute_on_block(key, [rev, cost, None])

    if threads == 1:
        stream(0)
    else:
        th = [threading.Thread(target=stream, args=(t,)) for t in range(threads)]
        [x.start() for x in th]
        [x.join() for x in th]
        for a in aggs[1:]:
            aggs[0].merge(a)
    keys, (s_rev, s_cost, cnt) = aggs[0].convert_to_block()
    return _decode_groups(keys, s_rev, s_cost, cnt)


class ShardedSink:
    """A ShardedGroupBy (clickhouse_amd.distributed) behind the Aggregator calls the plans make: rows in on every rank, the groups this rank
    OWNS out (partial states routed by key hash in one exchange, owner-side merge)."""

    def __init__(self, sharded_group_by):
        self.sg = sharded_group_by

    def execute_on_block(self, key, args):
        self.sg.add_block(key, args)

    def convert_to_block(self):
        return self.sg.finish()


def q41_sharded_gpu(ch, D, engine, dims, lo_share):
    """BASELINE.json configs[4] across ranks: the lineorder rows are sharded by row range (lo_share = THIS rank's rows), the dimension tables
    are replicated -- the reference's shared probe map (src/Interpreters/ConcurrentHashJoin.h:25-39) -- every rank runs the Q4.1 plan over its
    rows, and the partial (year, nation) states meet at their owners through the sharded GROUP BY (one exchange;
    AggregatingTransform.cpp:120-136 / Aggregator::mergeBucketImpl).  -> {(year, nation): (profit, count)} of the groups this rank owns."""
    def group_by(key, args):
        sink = ShardedSink(D.ShardedGroupBy(engine, np.uint64, Q41_AGGS(ch)))
        sink.execute_on_block(key, args + [None])
        return sink.convert_to_block()
    return q41_gpu(ch, engine.ctx, dims, lo_share, group_by=group_by)


def q41_sharded_cpu(O, D, engine, dims, lo_share, block_rows=65409):
    """the same orchestration over the oracle (engine: tests/cpu_engine.CpuEngine over gloo)"""
    return q41_cpu(O, dims, lo_share, block_rows=block_rows, make_agg=lambda: ShardedSink(D.ShardedGroupBy(engine, np.uint64, Q41_AGGS(O))))
