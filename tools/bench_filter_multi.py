#!/usr/bin/env python3
"""filter_columns over four UInt32 columns + one more (the first FilterTransform of the SSB plan, C5): usage bench_filter_multi.py [rows] [selectivity]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import clickhouse_amd as ch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 750_000_000
sel = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
ctx = ch.Context(0)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
cols_t = [torch.randint(0, 2**31 - 1, (rows,), dtype=torch.int32, device=dev, generator=g) for _ in range(5)]
mask_t = (torch.rand(rows, device=dev, generator=g) < sel).to(torch.uint8)
cols = [ctx.wrap(t.data_ptr(), np.uint32, rows, keepalive=t) for t in cols_t]
mask = ctx.wrap(mask_t.data_ptr(), np.uint8, rows, keepalive=mask_t)
kept = int(mask_t.sum().item())
for nc in (4, 5, 1):
    best = None
    for _ in range(4):
        ctx.synchronize()
        t0 = time.perf_counter()
        out = ch.filter_columns(cols[:nc], mask)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        del out
    bytes_ = rows * (2 + 4 * nc) + kept * 4 * nc
    print(f"filter_columns x{nc} rows={rows} sel={kept / rows:.3f}: {best * 1e3:.2f} ms  {bytes_ / best / 1e9:.0f} GB/s", flush=True)
want = cols_t[0][mask_t.bool()]
out = ch.filter_columns(cols[:4], mask)
got = torch.from_numpy(out[0].numpy().view(np.int32))
assert torch.equal(got, want.cpu()), "filter result differs"
print("ok")
