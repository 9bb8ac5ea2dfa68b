#!/usr/bin/env python3
"""GROUP BY with sum(Float64) at C3's shape, fixed-point states (deterministic_float_sums = 1, the default) against double states (= 0).
usage: python tools/bench_float_sums.py [rows] [groups]"""
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
k = torch.randint(0, groups, (rows,), dtype=torch.int64, device=dev, generator=g)
v = torch.rand(rows, dtype=torch.float64, device=dev, generator=g) * 1e6 - 5e5
out = {"rows": rows, "groups": groups}
for det in (1, 0):
    ctx = ch.Context(0)
    ctx.set_option("deterministic_float_sums", det)
    kc = ctx.wrap(k.data_ptr(), np.uint64, rows, keepalive=k)
    vc = ctx.wrap(v.data_ptr(), np.float64, rows, keepalive=v)
    for aggs, name in (([(ch.AGG_SUM, np.float64), (ch.AGG_COUNT, None)], "sum_count"), ([(ch.AGG_SUM, np.float64)], "sum"), ([(ch.AGG_AVG, np.float64)], "avg")):
        best, res = 1e9, None
        for it in range(4):
            A = ch.Aggregator(np.uint64, aggs, size_hint=groups, ctx=ctx)
            ctx.synchronize()
            t0 = time.perf_counter()
            A.execute_on_block(kc, [vc if a[0] != ch.AGG_COUNT else None for a in aggs])
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
            if it == 0:
                gk, r = A.convert_to_block()
                res = r[0][np.argsort(gk)]
            A.close()
        out[f"{name}_det{det}_ms"] = round(best, 3)
        out[f"{name}_det{det}_check"] = float(res[:1000].sum())
    ctx.close()
print(json.dumps(out))
