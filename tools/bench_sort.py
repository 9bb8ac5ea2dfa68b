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
    dl = best_of(lambda: ch.sort_permutation_limit(col, 10, False, 1))
    res.append({"case": f"ORDER BY {name} LIMIT 10 (sampled threshold + candidates)", "rows": rows, "ms": dl * 1e3, "rows_per_s": rows / dl,
                "algorithmic_B_per_row": width + 2, "GBps": (width + 2) * rows / dl / 1e9, "roofline_frac": (width + 2) * rows / dl / 8e12})
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
# String key dictionary encode: n values "CITY-dddd" (9 bytes + terminating zero), 2500 distinct
from clickhouse_amd.lowcardinality import ColumnString
m = rows
cid = torch.randint(0, 2500, (m,), dtype=torch.int64, device=dev, generator=g)
chars = torch.empty((m, 10), dtype=torch.uint8, device=dev)
for k, ch_ in enumerate(b"CITY-"):
    chars[:, k] = ch_
for k, div in enumerate((1000, 100, 10, 1)):
    chars[:, 5 + k] = ((cid // div) % 10 + 48).to(torch.uint8)
chars[:, 9] = 0
offs = (torch.arange(1, m + 1, dtype=torch.int64, device=dev) * 10)
cs = ColumnString(ctx.wrap(offs.data_ptr(), np.uint64, m, keepalive=offs), ctx.wrap(chars.data_ptr(), np.uint8, m * 10, keepalive=chars))
dt = best_of(lambda: cs.dictionary_encode(), reps=3)
lc2 = cs.dictionary_encode()
assert len(lc2.dictionary) == 2500 and lc2.dictionary[0] == bytes(chars[0, :9].cpu().numpy())
res.append({"case": "chgpu_string_dictionary_encode, 10-byte values, 2500 distinct", "rows": m, "ms": dt * 1e3, "rows_per_s": m / dt,
            "algorithmic_B_per_row": 22, "GBps": 22 * m / dt / 1e9, "roofline_frac": 22 * m / dt / 8e12})
print(json.dumps({"results": res}))
