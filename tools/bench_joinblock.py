#!/usr/bin/env python3
"""joinBlock's joinRightColumns through the ordered path (chgpu_join_probe: counts -> offsets -> emitted right row ids), the shape a
JoiningTransform meets with a dimension table on the right: probe rows x build rows of unique keys, half of the probe keys present.
usage: python tools/bench_joinblock.py [probe_rows] [build_rows] [kind: inner|left|semi] [key: u64|u32]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clickhouse_amd as ch

n_probe = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
n_build = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
kind = sys.argv[3] if len(sys.argv) > 3 else "inner"
key = sys.argv[4] if len(sys.argv) > 4 else "u64"
tdt, ndt = (torch.int64, np.uint64) if key == "u64" else (torch.int32, np.uint32)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
bk = (torch.randperm(n_build, device=dev, generator=g).to(tdt) + 1) * (2654435761 if key == "u64" else 3)
pk = torch.where(torch.rand(n_probe, device=dev, generator=g) < 0.5, bk[torch.randint(0, n_build, (n_probe,), device=dev, generator=g)],
                 torch.randint(0, 2**30, (n_probe,), device=dev, generator=g).to(tdt) * 2 + (7 if key == "u64" else 1))
ctx = ch.Context(0)
bkc = ctx.wrap(bk.data_ptr(), ndt, n_build, keepalive=bk)
pkc = ctx.wrap(pk.data_ptr(), ndt, n_probe, keepalive=pk)
j = ch.HashJoin(ch.JOIN_INNER if kind == "inner" else ch.JOIN_LEFT, ch.STRICT_SEMI if kind == "semi" else ch.STRICT_ALL, key_dtype=ndt, ctx=ctx)
j.add_block(bkc)
j.finish_build()
best = 1e9
for it in range(4):
    ctx.synchronize()
    t0 = time.perf_counter()
    r = j.probe_columns(pkc, need_right_rows=kind != "semi")  # semi: the filter-only probe (k_join_probe_filter)
    ctx.synchronize()
    best = min(best, (time.perf_counter() - t0) * 1e3)
    n_out = r["n_out"]
    del r
sbk, _ = torch.sort(bk)
pos = torch.searchsorted(sbk, pk).clamp_(max=n_build - 1)
hits = int((sbk[pos] == pk).sum().item())
assert n_out == (n_probe if kind == "left" else hits), (n_out, hits)
print(json.dumps({"probe_rows": n_probe, "build_rows": n_build, "kind": kind, "key": key, "n_out": n_out, "ms": round(best, 3), "probe_rows_per_s": round(n_probe / best * 1e3)}))
