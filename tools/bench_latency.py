#!/usr/bin/env python3
"""Per-call latency of the main operators at ClickHouse-Block-sized to stripe-sized inputs (what stripe size amortises the launches
and the host synchronisation of a call).  Prints a table: rows | operator | us per call | rows/s.  usage: bench_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import clickhouse_amd as ch
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)
g = torch.Generator(device=dev).manual_seed(1)
NMAX = 64 << 20
a = torch.randint(0, 2**31, (NMAX,), dtype=torch.int64, device=dev, generator=g)
k = torch.randint(0, 1000, (NMAX,), dtype=torch.int32, device=dev, generator=g)
bk = (torch.randperm(1_000_000, device=dev, generator=g).to(torch.int64) + 1) * 2654435761
jn = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, ctx=ctx)
jn.add_block(ctx.wrap(bk.data_ptr(), np.uint64, bk.shape[0], keepalive=bk)); jn.finish_build()
pk = bk[torch.randint(0, bk.shape[0], (NMAX,), device=dev, generator=g)]

def timed(fn, reps):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps

print(f"{'rows':>10s}  {'operator':34s} {'us/call':>9s}  {'rows/s':>10s}")
for n in (65_409, 1 << 20, 16 << 20, 64 << 20):
    ac = ctx.wrap(a.data_ptr(), np.int64, n, keepalive=a)
    kc = ctx.wrap(k.data_ptr(), np.uint32, n, keepalive=k)
    pc = ctx.wrap(pk.data_ptr(), np.uint64, n, keepalive=pk)
    reps = max(3, min(200, (64 << 20) // n))
    def f_fs(): ch.filter_sum(ac, ch.LT, 214748365)
    def f_filter():
        m = ch.cmp_const(ac, ch.LT, 214748365)
        ch.filter_columns([ac, kc], m)
    def f_agg():
        ag = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], ctx=ctx)
        ag.execute_on_block(kc, [ac, None]); len(ag); ag.close()
    def f_probe(): jn.probe_columns(pc)
    for name, fn in (("filter_sum (fused, 1 sync)", f_fs), ("cmp_const + filter_columns(2 cols)", f_filter), ("GROUP BY 1000 groups (create..size)", f_agg), ("join probe (1e6-key table)", f_probe)):
        dt = timed(fn, reps)
        print(f"{n:10d}  {name:34s} {dt*1e6:9.1f}  {n/dt:10.3e}", flush=True)
