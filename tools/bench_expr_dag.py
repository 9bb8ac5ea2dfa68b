#!/usr/bin/env python3
"""Run-time-compiled expression DAG kernels (csrc/expr_jit.hip) next to the hand-written kernels they generalise, over
HBM-resident columns.  usage: bench_expr_dag.py [rows]   -> one JSON object on stdout"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import clickhouse_amd as ch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)
g = torch.Generator(device=dev).manual_seed(3)


def best_of(fn, reps=10):
    best, out = None, None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, out


def line(name, dt, bytes_per_row, extra=None):
    d = {"case": name, "ms": dt * 1e3, "rows_per_s": rows / dt, "algorithmic_B_per_row": bytes_per_row,
         "GBps": bytes_per_row * rows / dt / 1e9, "roofline_frac": bytes_per_row * rows / dt / 8e12}
    d.update(extra or {})
    return d


res = []
# --- C2: SELECT sum(a), count() WHERE a < C over Int64 ---------------------------------------------------------------
a = torch.randint(0, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
ca = ctx.wrap(a.data_ptr(), np.int64, rows, keepalive=a)
d = ch.ActionsDAG()
ia = d.add_input(0, np.int64)
p = d.add_function("less", ia, d.add_column(214748365, np.uint32))
ex = d.compile()
t0 = time.perf_counter(); ex.filter_sum(ctx, [ca.cut(0, 1024)], p, ia); jit_first = time.perf_counter() - t0
dt_j, (s_j, c_j) = best_of(lambda: ex.filter_sum(ctx, [ca], p, ia))
dt_h, (s_h, c_h) = best_of(lambda: ch.filter_sum(ca, ch.LT, 214748365))
assert (int(s_j), c_j) == (int(s_h), c_h)
res.append(line("C2 filter+sum Int64: JIT DAG", dt_j, 8, {"first_call_incl_hiprtc_ms": jit_first * 1e3}))
res.append(line("C2 filter+sum Int64: hand-written k_filter_sum", dt_h, 8))
# materialised mask (comparison) and product
dt_m, outs = best_of(lambda: ex.execute(ctx, [ca], [p]), reps=5)
dt_c, m2 = best_of(lambda: ch.cmp_const(ca, ch.LT, 214748365), reps=5)
res.append(line("less(a, C) -> UInt8 mask: JIT DAG", dt_m, 9))
res.append(line("less(a, C) -> UInt8 mask: hand-written k_cmp_mask", dt_c, 9))
# WHERE a < C (10 % pass) + projection (a, a * 3): fused k_fcount + k_femit against mask -> filter_columns -> arithmetic
d2 = ch.ActionsDAG()
ja = d2.add_input(0, np.int64)
jp = d2.add_function("less", ja, d2.add_column(214748365, np.uint32))
jv = d2.add_function("multiply", ja, d2.add_column(3, np.uint8))
ex2 = d2.compile()
ex2.filter_execute(ctx, [ca.cut(0, 4096)], jp, [ja, jv])
dt_f, (fo, frows) = best_of(lambda: ex2.filter_execute(ctx, [ca], jp, [ja, jv]), reps=5)
def unfused_project():
    mk = ch.cmp_const(ca, ch.LT, 214748365)
    fa = ch.filter_columns([ca], mk)[0]
    return fa, ex2.execute(ctx, [fa], [jv])[0]
dt_u2, (ua, uv) = best_of(unfused_project, reps=5)
assert frows == ua.size() and int(ch.sum_add_many(fo[1])[0]) == int(ch.sum_add_many(uv)[0])
res.append(line("WHERE a < C + projection (a, a*3), 10 % pass: JIT k_fcount + k_femit", dt_f, 8 + 1.6, {"rows_out": frows}))
res.append(line("WHERE a < C + projection: k_cmp_mask + filter_columns + JIT multiply", dt_u2, 8 + 1.6))
del outs, m2, a, ca, fo, ua, uv
ctx.trim()
torch.cuda.empty_cache()
# --- SSB Q1.1 over the real widths (UInt32, UInt8, UInt8, UInt32: 10 B/row) ------------------------------------------------
od = (torch.randint(0, 70000, (rows,), dtype=torch.int32, device=dev, generator=g) + 19920101)
disc = torch.randint(0, 11, (rows,), dtype=torch.int32, device=dev, generator=g).to(torch.uint8)
qty = torch.randint(1, 51, (rows,), dtype=torch.int32, device=dev, generator=g).to(torch.uint8)
price = torch.randint(90_000, 10_000_000, (rows,), dtype=torch.int32, device=dev, generator=g)
ts = [od, disc, qty, price]
cols = [ctx.wrap(t.data_ptr(), np.uint32 if t.dtype == torch.int32 else np.uint8, rows, keepalive=t) for t in ts]
d = ch.ActionsDAG()
iod, idisc, iqty, iprice = d.add_input(0, np.uint32), d.add_input(1, np.uint8), d.add_input(2, np.uint8), d.add_input(3, np.uint32)
c = lambda v, t: d.add_column(v, t)
f = d.add_function("and", d.add_function("greaterOrEquals", iod, c(19930101, np.uint32)), d.add_function("lessOrEquals", iod, c(19931231, np.uint32)))
f = d.add_function("and", f, d.add_function("greaterOrEquals", idisc, c(1, np.uint8)))
f = d.add_function("and", f, d.add_function("lessOrEquals", idisc, c(3, np.uint8)))
f = d.add_function("and", f, d.add_function("less", iqty, c(25, np.uint8)))
v = d.add_function("multiply", iprice, idisc)
ex = d.compile()
preds = [(0, ch.GE, 19930101), (0, ch.LE, 19931231), (1, ch.GE, 1), (1, ch.LE, 3), (2, ch.LT, 25)]
dt_j, (s_j, c_j) = best_of(lambda: ex.filter_sum(ctx, cols, f, v))
dt_h, (s_h, c_h) = best_of(lambda: ch.expr_filter_sum(cols, preds, ch.VAL_MUL, 3, 1))
assert (int(s_j), c_j) == (int(s_h), c_h), (s_j, c_j, s_h, c_h)
res.append(line("SSB Q1.1 (U32,U8,U8,U32): JIT DAG", dt_j, 10))
res.append(line("SSB Q1.1 (U32,U8,U8,U32): hand-written k_expr_filter_sum", dt_h, 10))
# the unfused reference shape on the GPU: 5 comparisons + 4 ands + multiply + filter of the product + sum
def unfused():
    m = None
    for (ci, op, val) in preds:
        k = ch.cmp_const(cols[ci], op, val)
        m = k if m is None else ch.and_(m, k)
    prod = ex.execute(ctx, cols, [v])[0]
    return ch.sum_add_many_conditional(prod, m)
dt_u, s_u = best_of(unfused, reps=3)
assert int(s_u[0]) == int(s_j)
res.append(line("SSB Q1.1: one kernel per action (the reference's shape, on the GPU)", dt_u, 10))
print(json.dumps({"rows": rows, "results": res}))
