#!/usr/bin/env python3
"""SSB Q1.1-style fused filter + product + sum over HBM-resident UInt32 columns:
   SELECT sum(lo_extendedprice * lo_discount) WHERE lo_orderdate BETWEEN 19930101 AND 19931231
          AND lo_discount BETWEEN 1 AND 3 AND lo_quantity < 25
GPU: ONE fused kernel (16 B/row).  CPU: the oracle's per-Block pipeline on a sample.  usage: bench_q11.py [rows]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import clickhouse_amd as ch
import oracle as O

O.build()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)
g = torch.Generator(device=dev).manual_seed(3)
od = torch.randint(0, 70000, (rows,), dtype=torch.int32, device=dev, generator=g) + 19920101
disc = torch.randint(0, 11, (rows,), dtype=torch.int32, device=dev, generator=g)
qty = torch.randint(1, 51, (rows,), dtype=torch.int32, device=dev, generator=g)
price = torch.randint(90_000, 10_000_000, (rows,), dtype=torch.int32, device=dev, generator=g)
ts = [od, disc, qty, price]
cols = [ctx.wrap(t.data_ptr(), np.uint32, rows, keepalive=t) for t in ts]
preds = [(0, ch.GE, 19930101), (0, ch.LE, 19931231), (1, ch.GE, 1), (1, ch.LE, 3), (2, ch.LT, 25)]
best = None
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s, c = ch.expr_filter_sum(cols, preds, ch.VAL_MUL, 3, 1)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
m = min(rows, 200_000_000)
host = [t[:m].cpu().numpy().view(np.uint32) for t in ts]
cores = max(1, min(16, len(os.sched_getaffinity(0))))
t0 = time.perf_counter(); r1 = O.expr_filter_sum_pipeline(host, preds, O.VAL_MUL, 3, 1, threads=1); t1 = time.perf_counter() - t0
t0 = time.perf_counter(); rN = O.expr_filter_sum_pipeline(host, preds, O.VAL_MUL, 3, 1, threads=cores); tN = time.perf_counter() - t0
sg, cg = ch.expr_filter_sum([c_.cut(0, m) for c_ in cols], preds, ch.VAL_MUL, 3, 1)
assert (int(sg), cg) == (int(r1[0]), r1[1]) == (int(rN[0]), rN[1])
# the same query over the schema's real widths: UInt32 orderdate / extendedprice, UInt8 discount / quantity (10 B/row)
ts8 = [od, disc.to(torch.uint8), qty.to(torch.uint8), price]
cols8 = [ctx.wrap(t.data_ptr(), np.uint32 if t.dtype == torch.int32 else np.uint8, rows, keepalive=t) for t in ts8]
best8 = None
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s8, c8 = ch.expr_filter_sum(cols8, preds, ch.VAL_MUL, 3, 1)
    dt = time.perf_counter() - t0
    best8 = dt if best8 is None else min(best8, dt)
assert (int(s8), c8) == (int(s), c)
print(json.dumps({"query": "SSB Q1.1-style fused filter(5 predicates, 3 columns) + multiply + sum", "rows": rows,
                  "mixed_width_schema_u32_u8_u8_u32": {"gpu_ms_incl_readback": best8 * 1e3, "gpu_rows_per_s": rows / best8,
                                                       "algorithmic_GBps_10B_per_row": 10 * rows / best8 / 1e9, "roofline_frac": 10 * rows / best8 / 8e12,
                                                       "parity": "sum and count equal to the all-UInt32 run"}, "gpu_ms_incl_readback": best * 1e3,
                  "gpu_rows_per_s": rows / best, "algorithmic_GBps_16B_per_row": 16 * rows / best / 1e9, "roofline_frac": 16 * rows / best / 8e12,
                  "cpu_sample_rows": m, "cpu_1thread_rows_per_s": m / t1, f"cpu_{cores}threads_rows_per_s": m / tN, "selected_rows": c,
                  "parity": "sum and count bit-exact on the sample"}))
