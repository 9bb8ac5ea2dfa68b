#!/usr/bin/env python3
"""Fused filter+multiply+sum over 4 columns with different width patterns (which kernel variant runs, and how fast).
usage: bench_expr_widths.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import clickhouse_amd as ch
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)
g = torch.Generator(device=dev).manual_seed(3)
od = torch.randint(0, 70000, (rows,), dtype=torch.int32, device=dev, generator=g) + 19920101
disc = torch.randint(0, 11, (rows,), dtype=torch.int32, device=dev, generator=g)
qty = torch.randint(1, 51, (rows,), dtype=torch.int32, device=dev, generator=g)
price = torch.randint(90_000, 10_000_000, (rows,), dtype=torch.int32, device=dev, generator=g)
d8, q8 = disc.to(torch.uint8), qty.to(torch.uint8)
preds = [(0, ch.GE, 19930101), (0, ch.LE, 19931231), (1, ch.GE, 1), (1, ch.LE, 3), (2, ch.LT, 25)]
def W(t, dt): return ctx.wrap(t.data_ptr(), dt, rows, keepalive=t)
cases = {"u32 u32 u32 u32 (same-type kernel)": ([W(od, np.uint32), W(disc, np.uint32), W(qty, np.uint32), W(price, np.uint32)], 16),
         "u32 i32 i32 u32 (narrow kernel, all 4-byte)": ([W(od, np.uint32), W(disc, np.int32), W(qty, np.int32), W(price, np.uint32)], 16),
         "u32 u8 u8 u32 (narrow kernel)": ([W(od, np.uint32), W(d8, np.uint8), W(q8, np.uint8), W(price, np.uint32)], 10),
         "u32 u8 i32 u32 (narrow kernel)": ([W(od, np.uint32), W(d8, np.uint8), W(qty, np.int32), W(price, np.uint32)], 13)}
p64 = price.to(torch.int64)
cases["u32 u8 u8 i64 (generic mixed kernel)"] = ([W(od, np.uint32), W(d8, np.uint8), W(q8, np.uint8), W(p64, np.int64)], 14)
ref = None
for name, (cols, bpr) in cases.items():
    best = None
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s, c = ch.expr_filter_sum(cols, preds, ch.VAL_MUL, 3, 1)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    ref = ref or (int(s), c)
    assert (int(s), c) == ref
    print(f"{name:48s} {best*1e3:7.3f} ms  {bpr*rows/best/1e9:7.0f} GB/s algorithmic ({bpr} B/row)", flush=True)
