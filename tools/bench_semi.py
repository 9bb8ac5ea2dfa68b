#!/usr/bin/env python3
"""Filter-only LEFT SEMI probe of a large fact column against a small dense dimension (SSB supplier shape).
usage: bench_semi.py [rows] [dimension rows] [selectivity]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import clickhouse_amd as ch
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 750_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
sel = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)
g = torch.Generator(device=dev).manual_seed(7)
fact = torch.randint(1, dim + 1, (rows,), dtype=torch.int32, device=dev, generator=g)
keep = (torch.rand(dim, device=dev, generator=g) < sel).nonzero().flatten().to(torch.int32) + 1
fc = ctx.wrap(fact.data_ptr(), np.uint32, rows, keepalive=fact)
j = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_SEMI, key_dtype=np.uint32, ctx=ctx)
j.add_block(ctx.wrap(keep.data_ptr(), np.uint32, keep.shape[0], keepalive=keep)); j.finish_build()
best = None
for _ in range(4):
    ctx.synchronize(); t0 = time.perf_counter()
    r = j.probe_columns(fc, need_right_rows=False)
    ctx.synchronize(); dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
    n_out = r["n_out"]; del r
print(f"semi filter-only rows={rows} dim_keys={keep.shape[0]} kept={n_out}: {best*1e3:.2f} ms  {rows/best:.3e} rows/s  {5*rows/best/1e9:.0f} GB/s algorithmic (4+1 B/row)")
