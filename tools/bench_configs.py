#!/usr/bin/env python3
"""BASELINE.json configs C1, C3 and C4 (single-GPU shapes): GPU time through the C ABI, the CPU restatement (oracle/) timed
on a bounded sample of the same columns, and a parity check on that sample.  Prints one JSON object.
(C2 is bench.py itself.)  usage: bench_configs.py [rows_c3] [probe_rows_c4]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import clickhouse_amd as ch
import oracle as O

O.build()
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)
cores = max(1, min(16, len(os.sched_getaffinity(0))))
out = {"cores": cores}


def best(fn, reps):
    b, r = None, None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        b = dt if b is None else min(b, dt)
    return b, r


# ---------------- C1: 10 M rows, single CPU thread (plumbing) + the same on the GPU -----------------------------------
rng = np.random.Generator(np.random.PCG64(1))
a = rng.integers(0, 2**31, size=10_000_000, dtype=np.int64)
t_cpu, r_cpu = best(lambda: O.filter_sum_pipeline(a, O.LT, 214748365, threads=1), 5)
col = ctx.upload(a)
t_gpu, r_gpu = best(lambda: ch.filter_sum(col, ch.LT, 214748365), 5)
assert (int(r_gpu[0]), r_gpu[1]) == (int(r_cpu[0]), r_cpu[1])
out["C1"] = {"rows": 10_000_000, "cpu_1thread_rows_per_s": 1e7 / t_cpu, "gpu_rows_per_s_incl_launch_and_readback": 1e7 / t_gpu,
             "result": [int(r_cpu[0]), r_cpu[1]], "parity": "bit-exact"}

# ---------------- C3: GROUP BY UInt32 key (1 M groups), sum(Int64) + count ------------------------------------------------
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
g = torch.Generator(device=dev).manual_seed(2)
k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int32, device=dev, generator=g)
v = torch.randint(-2**31, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
kc = ctx.wrap(k.data_ptr(), np.uint32, rows, keepalive=k)
vc = ctx.wrap(v.data_ptr(), np.int64, rows, keepalive=v)
aggs = [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)]


def gpu_groupby(n):
    A = ch.Aggregator(np.uint32, aggs, size_hint=1_000_000, ctx=ctx)
    A.execute_on_block(kc, [vc, None], 0, n)
    return A


t_gpu, A = best(lambda: gpu_groupby(rows), 3)
sample = min(rows, 50_000_000)
ks = k[:sample].cpu().numpy().view(np.uint32)
vs = v[:sample].cpu().numpy()


def cpu_groupby(threads):
    parts = [None] * threads

    def work(t):
        lo, hi = sample * t // threads, sample * (t + 1) // threads
        ag = O.Aggregator(np.uint32, aggs)
        for b in range(lo, hi, O.DEFAULT_BLOCK_SIZE):
            e = min(hi, b + O.DEFAULT_BLOCK_SIZE)
            ag.execute_on_block(ks[b:e], [vs[b:e], None])
        parts[t] = ag
    th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    [x.start() for x in th]
    [x.join() for x in th]
    for p in parts[1:]:
        parts[0].merge(p)       # mergeDataImpl (bucket-wise when two-level)
    return parts[0]


t1, ag1 = best(lambda: cpu_groupby(1), 1)
tN, agN = best(lambda: cpu_groupby(cores), 2)
As = gpu_groupby(sample)
gk, (gs, gc) = As.convert_to_block()
ok, (os_, oc) = agN.convert_to_block()
gi, oi = np.argsort(gk), np.argsort(ok)
assert np.array_equal(gk[gi], ok[oi]) and np.array_equal(gs[gi], os_[oi]) and np.array_equal(gc[gi], oc[oi])
out["C3"] = {"rows": rows, "groups": len(A), "gpu_ms": t_gpu * 1e3, "gpu_rows_per_s": rows / t_gpu, "algorithmic_GBps": 12 * rows / t_gpu / 1e9,
             "roofline_frac": 12 * rows / t_gpu / 8e12, "cpu_sample_rows": sample, "cpu_1thread_rows_per_s": sample / t1,
             f"cpu_{cores}threads_rows_per_s": sample / tN, "parity": "bit-exact on the sample (keys, sums, counts)"}
del k, v, kc, vc, A, As

# ---------------- C4 (one GPU): 1e8-row probe JOIN 1e7-row build on UInt64, INNER ALL, payload gathered --------------------
nb = 10_000_000
npb = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
g = torch.Generator(device=dev).manual_seed(5)
bk = (torch.randperm(nb, device=dev, generator=g).to(torch.int64) + 1) * 2654435761
pk = torch.where(torch.rand(npb, device=dev, generator=g) < 0.5, bk[torch.randint(0, nb, (npb,), device=dev, generator=g)],
                 torch.randint(0, 2**62, (npb,), dtype=torch.int64, device=dev, generator=g))
bv = torch.randint(-2**40, 2**40, (nb,), dtype=torch.int64, device=dev, generator=g)
bkc = ctx.wrap(bk.data_ptr(), np.uint64, nb, keepalive=bk)
pkc = ctx.wrap(pk.data_ptr(), np.uint64, npb, keepalive=pk)
bvc = ctx.wrap(bv.data_ptr(), np.int64, nb, keepalive=bv)


def gpu_build():
    j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, ctx=ctx)
    j.add_block(bkc)
    j.finish_build()
    return j


def gpu_probe(j, pcol):
    r = j.probe_columns(pcol)
    payload = bvc.index(r["right_rowid"], default_for_missing=True)
    s, c = ch.filter_sum(payload, ch.GE, -2**62)   # count(), sum(bv) checksum form
    return r["n_out"], int(s), c


t_build, j = best(gpu_build, 3)
t_probe, (n_out, chk, cnt) = best(lambda: gpu_probe(j, pkc), 3)
psample = min(npb, 20_000_000)
bk_h, pk_h, bv_h = bk.cpu().numpy().view(np.uint64), pk[:psample].cpu().numpy().view(np.uint64), bv.cpu().numpy()
t0 = time.perf_counter()
oj = O.HashJoin(O.JOIN_INNER, O.STRICT_ALL)
for b in range(0, nb, O.DEFAULT_BLOCK_SIZE):
    oj.add_block(bk_h[b:b + O.DEFAULT_BLOCK_SIZE])
t_cb = time.perf_counter() - t0
t0 = time.perf_counter()
cpu_rows, cpu_chk = 0, 0
blk = O.DEFAULT_BLOCK_SIZE
for b in range(0, psample, blk):
    l, rb, rr, c = oj.joined_pairs(pk_h[b:b + blk])
    cpu_rows += l.shape[0]
    cpu_chk += int(bv_h[rb * blk + rr].sum())     # fillFromBlocksAndRowNumbers: (block, row) -> build row
t_cp = time.perf_counter() - t0
n_s, chk_s, cnt_s = gpu_probe(j, pkc.cut(0, psample))
assert (n_s, chk_s) == (cpu_rows, cpu_chk & 0xFFFFFFFFFFFFFFFF if cpu_chk >= 0 else cpu_chk) or (n_s == cpu_rows and (chk_s - cpu_chk) % 2**64 == 0)
out["C4_one_gpu"] = {"build_rows": nb, "probe_rows": npb, "matches": n_out, "gpu_build_ms": t_build * 1e3, "gpu_probe_gather_sum_ms": t_probe * 1e3,
                     "gpu_probe_rows_per_s": npb / t_probe, "algorithmic_GBps": (8 * npb + 16 * nb + 8 * n_out) / (t_build + t_probe) / 1e9,
                     "cpu_1thread_build_rows_per_s": nb / t_cb, "cpu_1thread_probe_rows_per_s": psample / t_cp, "cpu_probe_sample_rows": psample,
                     "parity": "match count and sum(payload) bit-exact on the sample"}
print(json.dumps(out))
