#!/usr/bin/env python3
"""A/B harness of round 2: C3 (GROUP BY 1e9 rows, 1 M groups) and C4 one-GPU (1e8 probe x 1e7 build, fused count/sum) device times for
the build selected by the CHGPU_TUNE_* environment (read once per process).  usage: bench_r02.py [c3|c4|both] [rows_c3] [label]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import clickhouse_amd as ch

what = sys.argv[1] if len(sys.argv) > 1 else "both"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
label = sys.argv[3] if len(sys.argv) > 3 else ""
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)
out = {"label": label, "env": {k: v for k, v in os.environ.items() if k.startswith("CHGPU_TUNE")}}


def timed(fn, reps=5, warmup=2):
    r = None
    for _ in range(warmup):
        r = fn()
    st.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        r = fn()
        e1.record(st)
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return min(ms), sum(ms) / len(ms), r


if what in ("c3", "both"):
    g = torch.Generator(device=dev).manual_seed(2)
    k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int32, device=dev, generator=g)
    v = torch.randint(-2**31, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
    kc = ctx.wrap(k.data_ptr(), np.uint32, rows, keepalive=k)
    vc = ctx.wrap(v.data_ptr(), np.int64, rows, keepalive=v)
    aggs = [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)]

    def run():
        A = ch.Aggregator(np.uint32, aggs, size_hint=1_000_000, ctx=ctx)
        A.execute_on_block(kc, [vc, None])
        return A

    best, avg, A = timed(run)
    gk, (gs, gc) = A.convert_to_block()
    assert os.environ.get('CHGPU_EXPERIMENT_TILES') or gk.shape[0] == len(A) and int(gc.sum()) == rows and int(gs.astype(np.uint64).sum(dtype=np.uint64)) == int(v.sum().item()) % 2**64
    out["C3"] = {"rows": rows, "groups": len(A), "best_ms": best, "avg_ms": avg, "frac": 12.0 * rows / (avg * 1e-3) / 8e12}
    del k, v, kc, vc, A
    ctx.trim()
    torch.cuda.empty_cache()

if what in ("c4", "both"):
    nb, npb = 10_000_000, 100_000_000
    g = torch.Generator(device=dev).manual_seed(5)
    bk = (torch.randperm(nb, device=dev, generator=g).to(torch.int64) + 1) * 2654435761
    pk = torch.where(torch.rand(npb, device=dev, generator=g) < 0.5, bk[torch.randint(0, nb, (npb,), device=dev, generator=g)],
                     torch.randint(0, 2**62, (npb,), dtype=torch.int64, device=dev, generator=g))
    bv = torch.randint(-2**40, 2**40, (nb,), dtype=torch.int64, device=dev, generator=g)
    bkc = ctx.wrap(bk.data_ptr(), np.uint64, nb, keepalive=bk)
    pkc = ctx.wrap(pk.data_ptr(), np.uint64, npb, keepalive=pk)
    bvc = ctx.wrap(bv.data_ptr(), np.int64, nb, keepalive=bv)

    def build():
        j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, ctx=ctx)
        j.add_block(bkc)
        j.finish_build()
        return j

    bb, ba, j = timed(build, reps=3, warmup=1)
    pb, pa, (cnt, sm) = timed(lambda: j.probe_count_sum(pkc, bvc))
    sbk, order = torch.sort(bk)
    pos = torch.searchsorted(sbk, pk).clamp_(max=nb - 1)
    hit = sbk[pos] == pk
    assert os.environ.get('CHGPU_EXPERIMENT_JOIN_LDS') or (cnt, sm % 2**64) == (int(hit.sum().item()), int(bv[order[pos[hit]]].sum().item()) % 2**64)
    out["C4"] = {"build_best_ms": bb, "build_avg_ms": ba, "probe_best_ms": pb, "probe_avg_ms": pa, "matches": cnt,
                 "frac": (8.0 * npb + 16.0 * nb + 8.0 * cnt) / ((ba + pa) * 1e-3) / 8e12}
print(json.dumps(out))
