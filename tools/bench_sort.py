#!/usr/bin/env python3
"""ORDER BY on the device: chgpu_sort_permutation (stable LSD radix sort, 8-bit digits over the stable partition kernels) and the
LowCardinality row translation, HBM-resident inputs.  usage: bench_sort.py [rows]  -> one JSON object"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import clickhouse_amd as ch
from clickhouse_amd.lowcardinality import ColumnLowCardinality, LowCardinalityDictionary

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)
g = torch.Generator(device=dev).manual_seed(5)
res = []


def best_of(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        del out
    return best


for name, t, npdt, width in [("Int64", torch.randint(-2**62, 2**62, (rows,), dtype=torch.int64, device=dev, generator=g), np.int64, 8),
                             ("UInt32", torch.randint(0, 2**31, (rows,), dtype=torch.int32, device=dev, generator=g), np.uint32, 4),
                             ("Float64", torch.randn((rows,), dtype=torch.float64, device=dev, generator=g), np.float64, 8)]:
    col = ctx.wrap(t.data_ptr(), npdt, rows, keepalive=t)
    dt = best_of(lambda: ch.sort_permutation(col, None, False, 1))
    m = min(rows, 20_000_000)
    host = t[:m].cpu().numpy().view(npdt)
    t0 = time.perf_counter(); np.argsort(host, kind="stable"); tc = time.perf_counter() - t0
    passes = width
    res.append({"case": f"sort_permutation {name}", "rows": rows, "ms": dt * 1e3, "rows_per_s": rows / dt,
                "algorithmic_B_per_row": passes * 2 * (width + 8), "GBps": passes * 2 * (width + 8) * rows / dt / 1e9,
                "roofline_frac": passes * 2 * (width + 8) * rows / dt / 8e12,
                "cpu_numpy_stable_argsort_rows_per_s_1thread": m / tc, "cpu_sample_rows": m})
    del col, t
    ctx.trim()
    torch.cuda.empty_cache()

n = rows * 10
idx = torch.randint(0, 25, (n,), dtype=torch.int32, device=dev, generator=g).to(torch.uint8)
lc = ColumnLowCardinality([f"NATION-{i}" for i in range(25)], ctx.wrap(idx.data_ptr(), np.uint8, n, keepalive=idx))
gd = LowCardinalityDictionary(ctx)
dt = best_of(lambda: gd.map_block(lc), reps=5)
res.append({"case": "k_lc_remap UInt8 -> UInt32 ids, 25-entry dictionary", "rows": n, "ms": dt * 1e3, "rows_per_s": n / dt,
            "algorithmic_B_per_row": 5, "GBps": 5 * n / dt / 1e9, "roofline_frac": 5 * n / dt / 8e12})
print(json.dumps({"results": res}))
