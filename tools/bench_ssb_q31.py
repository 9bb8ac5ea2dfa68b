#!/usr/bin/env python3
"""SSB Q3.1 with String keys from compressed column files on one GPU's lineorder share (tools/ssb_q31.py): end-to-end time including
the decode of the compressed fact columns, the String dictionary encoding of the dimensions, joins, GROUP BY and ORDER BY; the numpy
restatement on the same box beside it.  usage: bench_ssb_q31.py [rows] [customers] [suppliers]  -> one JSON object"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import clickhouse_amd as ch
import ssb_q31 as Q
from oracle import compression as OC

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000
customers = int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000
suppliers = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
ctx = ch.Context(0)
dims, lo = Q.gen(rows, customers, suppliers)
files = Q.compress_lineorder(OC, lo)
dtypes = {k: v.dtype for k, v in lo.items()}
raw_bytes = sum(v.nbytes for v in lo.values())
comp_bytes = sum(len(b) for b in files.values())
best, got, phases = None, None, None
for _ in range(3):
    ctx.synchronize()
    t0 = time.perf_counter()
    times = {}
    got, kept = Q.q31_gpu(ch, ctx, dims, files, dtypes, times)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    if best is None or dt < best:
        best, phases = dt, times
t0 = time.perf_counter()
want, passed = Q.q31_cpu(dims, lo)
t_cpu = time.perf_counter() - t0
assert sorted(got) == sorted(want)
print(json.dumps({"query": "SSB Q3.1, String c_nation / s_nation / regions, Date lo_orderdate, compressed lineorder columns (LZ4, Delta+LZ4)",
                  "lineorder_rows": rows, "customers": customers, "suppliers": suppliers, "raw_fact_bytes": raw_bytes, "compressed_fact_bytes": comp_bytes,
                  "groups": len(got), "rows_after_year_filter": kept, "rows_joined": passed,
                  "gpu_ms_end_to_end_incl_upload_decode_dimension_encoding": best * 1e3, "gpu_phase_ms": phases, "gpu_rows_per_s": rows / best,
                  "cpu_numpy_restatement_ms_uncompressed_host_arrays_1thread": t_cpu * 1e3, "cpu_rows_per_s": rows / t_cpu,
                  "parity": "result rows equal"}))
