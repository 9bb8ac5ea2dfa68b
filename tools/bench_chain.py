#!/usr/bin/env python3
"""chgpu_join_probe_chain on the SSB Q4.1 shape (750 M fact rows; supplier / part key sets of 2 M keys, customer 30 M, date 2556): the chain
call alone, tables built beforehand.  usage: bench_chain.py [rows] [reps]  (run under rocprofv3 --kernel-trace --stats for per-kernel times)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

import clickhouse_amd as ch
import ssb

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 750_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
C, S, P = 30_000_000, 2_000_000, 2_000_000
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)
dims = ssb.gen_dims(C, S, P)
lo_t = ssb.gen_lineorder_torch(rows, C, S, P, dev)
torch.cuda.synchronize()
lo = {k: ctx.wrap(v.data_ptr(), np.uint32, rows, keepalive=v) for k, v in lo_t.items()}
up = ctx.upload
j_c = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, key_dtype=np.uint32, ctx=ctx)
j_c.add_block(up(dims["c_custkey"][dims["c_region"] == ssb.AMERICA]))
j_s = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_SEMI, key_dtype=np.uint32, ctx=ctx)
j_s.add_block(up(dims["s_suppkey"][dims["s_region"] == ssb.AMERICA]))
j_p = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_SEMI, key_dtype=np.uint32, ctx=ctx)
j_p.add_block(up(dims["p_partkey"][dims["p_mfgr"] <= 2]))
j_d = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, key_dtype=np.uint32, ctx=ctx)
j_d.add_block(up(dims["d_datekey"]))
for j in (j_c, j_s, j_p, j_d):
    j.finish_build()


def run():
    return ch.join_probe_chain([j_s, j_p, j_c, j_d], [lo["lo_suppkey"], lo["lo_partkey"], lo["lo_custkey"], lo["lo_orderdate"]],
                               right_rows=[False, False, True, True], carry=[lo["lo_revenue"], lo["lo_supplycost"]], want_indexes=False)


r = run()
ctx.synchronize()
best = None
for _ in range(reps):
    ctx.synchronize()
    t0 = time.perf_counter()
    r = run()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print(json.dumps({"rows": rows, "kept": r["kept"], "chain_ms_best": best * 1e3, "GBps_16B_per_row": 16 * rows / best / 1e9}))
