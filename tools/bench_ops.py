#!/usr/bin/env python3
"""Ad-hoc timings of the non-headline operators through the C ABI on device-resident inputs (configs C3/C4 shapes).
usage: bench_ops.py [groupby|join|filter|all] [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import clickhouse_amd as ch

what = sys.argv[1] if len(sys.argv) > 1 else "all"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)


def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        ctx.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, r


if what in ("groupby", "all"):
    for groups, hint in ((1_000_000, 1_000_000), (1000, 0), (1_000_000, 0)):
        g = torch.Generator(device=dev).manual_seed(2)
        k = torch.randint(0, groups, (rows,), dtype=torch.int32, device=dev, generator=g)
        v = torch.randint(-2**31, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
        kc = ctx.wrap(k.data_ptr(), np.uint32, rows, keepalive=k)
        vc = ctx.wrap(v.data_ptr(), np.int64, rows, keepalive=v)

        def run():
            a = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], size_hint=hint, ctx=ctx)
            a.execute_on_block(kc, [vc, None])
            n = len(a)
            a.close()
            return n
        dt, n = timed(run)
        print(f"groupby rows={rows} groups={groups} hint={hint}: {dt*1e3:.2f} ms  {rows/dt:.3e} rows/s  {12*rows/dt/1e9:.1f} GB/s algorithmic (12 B/row)  groups_out={n}", flush=True)
        del k, v, kc, vc

if what == "groupby_sweep":
    # cardinality sweep (with and without a size hint) to place the strategy thresholds
    for groups in (16, 1000, 4096, 16384, 65536, 262144, 1_000_000, 4_000_000, 16_000_000):
        g = torch.Generator(device=dev).manual_seed(2)
        k = torch.randint(0, groups, (rows,), dtype=torch.int32, device=dev, generator=g)
        v = torch.randint(-2**31, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
        kc = ctx.wrap(k.data_ptr(), np.uint32, rows, keepalive=k)
        vc = ctx.wrap(v.data_ptr(), np.int64, rows, keepalive=v)
        res = []
        for hint in (groups, 0):
            def run():
                a = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], size_hint=hint, ctx=ctx)
                a.execute_on_block(kc, [vc, None])
                n = len(a)
                a.close()
                return n
            dt, n = timed(run)
            assert n == min(groups, n)
            res.append(dt * 1e3)
        print(f"groupby_sweep rows={rows} groups={groups}: hinted {res[0]:.2f} ms, no hint {res[1]:.2f} ms", flush=True)
        del k, v, kc, vc

if what == "groupby_finalize":
    # convertToBlockImplFinal on the device: table -> (keys, sum, count) columns, left in HBM
    for groups in (1000, 1_000_000, 4_000_000):
        g = torch.Generator(device=dev).manual_seed(2)
        k = torch.randint(0, groups, (rows,), dtype=torch.int32, device=dev, generator=g)
        v = torch.randint(-2**31, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
        kc = ctx.wrap(k.data_ptr(), np.uint32, rows, keepalive=k)
        vc = ctx.wrap(v.data_ptr(), np.int64, rows, keepalive=v)
        a = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], size_hint=groups, ctx=ctx)
        a.execute_on_block(kc, [vc, None])

        def fin():
            keys_c, cols = a.finalize_columns()
            return keys_c.size()
        dt, n = timed(fin)
        print(f"groupby_finalize groups={groups}: {dt*1e3:.3f} ms for {n} result rows", flush=True)
        a.close()
        del k, v, kc, vc

if what == "groupby_narrow_args":
    # low-cardinality GROUP BY over 4-byte argument columns (SSB: sum(lo_revenue), sum(lo_supplycost) GROUP BY year, nation)
    g = torch.Generator(device=dev).manual_seed(2)
    k = torch.randint(0, 1000, (rows,), dtype=torch.int32, device=dev, generator=g)
    r = torch.randint(0, 2**31 - 1, (rows,), dtype=torch.int32, device=dev, generator=g)
    c2 = torch.randint(-2**31, 2**31 - 1, (rows,), dtype=torch.int32, device=dev, generator=g)
    kc = ctx.wrap(k.data_ptr(), np.uint32, rows, keepalive=k)
    rc = ctx.wrap(r.data_ptr(), np.uint32, rows, keepalive=r)
    cc = ctx.wrap(c2.data_ptr(), np.int32, rows, keepalive=c2)

    def run():
        a = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.uint32), (ch.AGG_SUM, np.int32), (ch.AGG_COUNT, None)], ctx=ctx)
        a.execute_on_block(kc, [rc, cc, None])
        n = len(a)
        a.close()
        return n
    dt, n = timed(run)
    print(f"groupby_narrow_args rows={rows} groups={n}: {dt*1e3:.2f} ms  {rows/dt:.3e} rows/s  {12*rows/dt/1e9:.0f} GB/s algorithmic (4+4+4 B/row)", flush=True)

if what == "groupby_q1":
    # TPC-H Q1 shape: 4 groups, sum x4, avg x3, count over 1e9 rows (7 argument functions -> 4 RANGE passes)
    g = torch.Generator(device=dev).manual_seed(2)
    k = torch.randint(0, 4, (rows,), dtype=torch.int32, device=dev, generator=g)
    qty = torch.randint(1, 51, (rows,), dtype=torch.int64, device=dev, generator=g)
    price = torch.randint(90_000, 10_000_000, (rows,), dtype=torch.int64, device=dev, generator=g)
    disc = torch.rand(rows, dtype=torch.float64, device=dev, generator=g)
    kc = ctx.wrap(k.data_ptr(), np.uint32, rows, keepalive=k)
    qc = ctx.wrap(qty.data_ptr(), np.int64, rows, keepalive=qty)
    pc = ctx.wrap(price.data_ptr(), np.int64, rows, keepalive=price)
    dc = ctx.wrap(disc.data_ptr(), np.float64, rows, keepalive=disc)
    aggs = [(ch.AGG_SUM, np.int64), (ch.AGG_SUM, np.int64), (ch.AGG_SUM, np.float64), (ch.AGG_SUM, np.int64),
            (ch.AGG_AVG, np.int64), (ch.AGG_AVG, np.int64), (ch.AGG_AVG, np.float64), (ch.AGG_COUNT, None)]

    def run():
        a = ch.Aggregator(np.uint32, aggs, ctx=ctx)
        a.execute_on_block(kc, [qc, pc, dc, pc, qc, pc, dc, None])
        n = len(a)
        a.close()
        return n
    dt, n = timed(run)
    print(f"groupby_q1 rows={rows} groups={n}: {dt*1e3:.2f} ms  {rows/dt:.3e} rows/s  {60*rows/dt/1e9:.0f} GB/s algorithmic (4 + 7x8 B/row)", flush=True)

if what == "filter_groupby":
    # SELECT k, sum(v), count() FROM t WHERE a < C GROUP BY k -- BASELINE's "filter + GROUP BY" with a key: 10 % of 1e9 rows pass
    g = torch.Generator(device=dev).manual_seed(2)
    a = torch.randint(0, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
    v = torch.randint(-2**31, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
    ac = ctx.wrap(a.data_ptr(), np.int64, rows, keepalive=a)
    vc = ctx.wrap(v.data_ptr(), np.int64, rows, keepalive=v)
    for groups, thr in ((1000, 214748365), (1000, 1503238554), (1_000_000, 214748365)):   # 10 % and 70 % of the rows pass
        k = torch.randint(0, groups, (rows,), dtype=torch.int32, device=dev, generator=g)
        kc = ctx.wrap(k.data_ptr(), np.uint32, rows, keepalive=k)

        def fused():
            m = ch.cmp_const(ac, ch.LT, thr)
            agg = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], size_hint=groups, ctx=ctx)
            agg.execute_on_block(kc, [vc, None], filter=m)
            n = len(agg)
            agg.close()
            return n

        def unfused():
            m = ch.cmp_const(ac, ch.LT, thr)
            fk, fv = ch.filter_columns([kc, vc], m)
            agg = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], size_hint=groups, ctx=ctx)
            agg.execute_on_block(fk, [fv, None])
            n = len(agg)
            agg.close()
            return n
        t1, n1 = timed(fused)
        t2, n2 = timed(unfused)
        assert n1 == n2
        print(f"filter_groupby rows={rows} groups={n1} pass={thr/2**31:.0%}: add_block_filtered {t1*1e3:.2f} ms ({rows/t1:.3e} rows/s, {21*rows/t1/1e9:.0f} GB/s of 8+1+4+8 B/row)   "
              f"cmp + filter_columns + add_block {t2*1e3:.2f} ms", flush=True)
        del k, kc

if what == "groupby_zipf":
    # SURVEY C3's skew variant: keys ~ Zipf(1.1) folded into [0, 1e6) (continuous inverse-CDF approximation on device)
    groups, sz = 1_000_000, 1.1
    g = torch.Generator(device=dev).manual_seed(2)
    u = torch.rand(rows, device=dev, generator=g, dtype=torch.float64)
    z = torch.floor(torch.pow(u * (float(groups) ** (1 - sz) - 1) + 1, 1 / (1 - sz))).to(torch.int64)
    k = (z % groups).to(torch.int32)
    del u, z
    v = torch.randint(-2**31, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
    top = torch.bincount(k[: 10_000_000].to(torch.int64), minlength=groups).max().item() / 10_000_000
    kc = ctx.wrap(k.data_ptr(), np.uint32, rows, keepalive=k)
    vc = ctx.wrap(v.data_ptr(), np.int64, rows, keepalive=v)
    for hint in (groups, 0):
        def run():
            a = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], size_hint=hint, ctx=ctx)
            a.execute_on_block(kc, [vc, None])
            n = len(a)
            a.close()
            return n
        dt, n = timed(run)
        print(f"groupby_zipf rows={rows} hint={hint}: {dt*1e3:.2f} ms  {rows/dt:.3e} rows/s  groups_out={n}  hottest key share={top:.3f}", flush=True)

if what in ("join", "all"):
    nb, npb = 10_000_000, min(rows, 100_000_000)
    g = torch.Generator(device=dev).manual_seed(5)
    bk = (torch.randperm(nb, device=dev, generator=g).to(torch.int64) + 1) * 2654435761
    pk = torch.where(torch.rand(npb, device=dev, generator=g) < 0.5, bk[torch.randint(0, nb, (npb,), device=dev, generator=g)],
                     torch.randint(0, 2**62, (npb,), dtype=torch.int64, device=dev, generator=g))
    bv = torch.randint(-2**40, 2**40, (nb,), dtype=torch.int64, device=dev, generator=g)
    bkc = ctx.wrap(bk.data_ptr(), np.uint64, nb, keepalive=bk)
    pkc = ctx.wrap(pk.data_ptr(), np.uint64, npb, keepalive=pk)
    bvc = ctx.wrap(bv.data_ptr(), np.int64, nb, keepalive=bv)

    def build():
        j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, ctx=ctx)
        j.add_block(bkc)
        j.finish_build()
        return j
    dt, j = timed(build)
    print(f"join build rows={nb}: {dt*1e3:.2f} ms  {nb/dt:.3e} rows/s", flush=True)

    def probe():
        r = j.probe_columns(pkc)
        out = bvc.index(r["right_rowid"], default_for_missing=True)
        return r["n_out"], out.size()
    dt, (n_out, _) = timed(probe)
    print(f"join probe+gather rows={npb} matches={n_out}: {dt*1e3:.2f} ms  {npb/dt:.3e} probe rows/s  {(8*npb+16*n_out)/dt/1e9:.1f} GB/s algorithmic", flush=True)

if what in ("filter", "all"):
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randint(0, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
    col = ctx.wrap(a.data_ptr(), np.int64, rows, keepalive=a)
    dt, mask = timed(lambda: ch.cmp_const(col, ch.LT, 214748365))
    print(f"cmp_const rows={rows}: {dt*1e3:.2f} ms  {9*rows/dt/1e9:.1f} GB/s (9 B/row)", flush=True)
    dt, out = timed(lambda: col.filter(mask))
    sel = out.size() / rows
    print(f"filter rows={rows} sel={sel:.3f}: {dt*1e3:.2f} ms  {(10 + 8*sel)*rows/dt/1e9:.1f} GB/s (8+1+1+8s B/row)", flush=True)
