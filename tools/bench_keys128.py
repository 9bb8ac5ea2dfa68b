#!/usr/bin/env python3
"""GROUP BY over keys128 (two UInt64 key columns packed into 16 bytes: AggregatedDataVariants keys128), sum(Int64) + count():
packFixed -> the exact dictionary (chgpu_keydict_encode) -> the UInt32 GROUP BY on the ids.  Times the encode and the whole operator.
usage: python tools/bench_keys128.py [rows] [groups]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clickhouse_amd as ch
from clickhouse_amd.keysfixed import KeyDict, KeysFixedAggregator

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
gid = torch.randint(0, groups, (rows,), dtype=torch.int64, device=dev, generator=g)
k1 = gid * 2654435761 + 17            # two key columns that are both needed to tell the groups apart
k2 = (gid >> 3) * 40503 + (gid & 7)
v = torch.randint(-2**40, 2**40, (rows,), dtype=torch.int64, device=dev, generator=g)
ctx = ch.Context(0)
k1c = ctx.wrap(k1.data_ptr(), np.uint64, rows, keepalive=k1)
k2c = ctx.wrap(k2.data_ptr(), np.uint64, rows, keepalive=k2)
vc = ctx.wrap(v.data_ptr(), np.int64, rows, keepalive=v)
out = {"rows": rows, "groups": groups}

best = 1e9
for it in range(4):
    d = KeyDict([np.uint64, np.uint64], ctx, size_hint=groups)
    ctx.synchronize()
    t0 = time.perf_counter()
    ids = d.encode([k1c, k2c], True)
    ctx.synchronize()
    best = min(best, (time.perf_counter() - t0) * 1e3)
    n = len(d)
    del ids
    d.close()
out["encode"] = {"ms": round(best, 3), "keys": n, "rows_per_s": round(rows / best * 1e3)}

aggs = [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)]
best = 1e9
for it in range(4):
    A = KeysFixedAggregator([np.uint64, np.uint64], aggs, size_hint=groups, ctx=ctx)
    ctx.synchronize()
    t0 = time.perf_counter()
    A.execute_on_block([k1c, k2c], [vc, None])
    ctx.synchronize()
    best = min(best, (time.perf_counter() - t0) * 1e3)
    n = len(A)
    last = A
(kk1, kk2), (s, c) = last.convert_to_block()
assert n == groups or rows < 20 * groups, (n, groups)
assert int(c.sum()) == rows and int(s.astype(np.uint64).sum(dtype=np.uint64)) == int(v.sum().item()) % 2**64
out["group_by"] = {"ms": round(best, 3), "groups": n, "rows_per_s": round(rows / best * 1e3),
                   "algorithmic_GBps": round(24 * rows / best / 1e6, 1), "frac_of_8TBps": round(24 * rows / best / 1e6 / 8000, 3)}
print(json.dumps(out))
