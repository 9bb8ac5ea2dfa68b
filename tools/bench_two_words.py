#!/usr/bin/env python3
"""GROUP BY with two argument words -- sum(a), sum(b), count() -- at C3's shape: the scatter plan (the tile-sorted plan carries one word).
usage: python tools/bench_two_words.py [rows] [groups]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clickhouse_amd as ch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
k = torch.randint(0, groups, (rows,), dtype=torch.int32, device=dev, generator=g)
a = torch.randint(-2**40, 2**40, (rows,), dtype=torch.int64, device=dev, generator=g)
b = torch.randint(-2**40, 2**40, (rows,), dtype=torch.int64, device=dev, generator=g)
ctx = ch.Context(0)
kc = ctx.wrap(k.data_ptr(), np.uint32, rows, keepalive=k)
ac = ctx.wrap(a.data_ptr(), np.int64, rows, keepalive=a)
bc = ctx.wrap(b.data_ptr(), np.int64, rows, keepalive=b)
out = {"rows": rows, "groups": groups}
for name, aggs, args in (("sum_sum_count", [(ch.AGG_SUM, np.int64), (ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], [ac, bc, None]),
                         ("avg_sum", [(ch.AGG_AVG, np.int64), (ch.AGG_SUM, np.int64)], [ac, bc]),
                         ("sum_count (one word: the tile-sorted plan)", [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], [ac, None])):
    best = 1e9
    for it in range(4):
        A = ch.Aggregator(np.uint32, aggs, size_hint=groups, ctx=ctx)
        ctx.synchronize()
        t0 = time.perf_counter()
        A.execute_on_block(kc, args)
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
        n = len(A)
        A.close()
    bytes_per_row = 4 + 8 * sum(1 for x in args if x is not None)
    out[name] = {"ms": round(best, 3), "groups": n, "algorithmic_GBps": round(bytes_per_row * rows / best / 1e6, 1), "frac_of_8TBps": round(bytes_per_row * rows / best / 1e6 / 8000, 3)}
print(json.dumps(out))
