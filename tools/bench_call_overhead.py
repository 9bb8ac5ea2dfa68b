#!/usr/bin/env python3
"""Host cost of the harness: what one call through ctypes + the Python wrappers costs, next to the kernels it launches.  (The SSB plan in
tools/ssb.py is ~40 such calls: its fixed part is mostly this.)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clickhouse_amd as ch

ctx = ch.Context(0)
small = ctx.upload(np.arange(1024, dtype=np.uint32))
u8 = ctx.upload((np.arange(1024) % 5).astype(np.uint8))
out = {}


def timeit(name, fn, reps=500):
    fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    out[name + "_us"] = round((time.perf_counter() - t0) / reps * 1e6, 2)


timeit("col_size (no device work)", lambda: small.size())
timeit("cmp_const on 1024 rows (1 launch, no sync)", lambda: ch.cmp_const(u8, ch.EQ, 1))
timeit("count_bytes_in_filter (1 launch + read-back)", lambda: ch.count_bytes_in_filter(u8))
timeit("filter (count + scan + scatter + read-back)", lambda: small.filter(u8))


def join_cycle():
    j = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_SEMI, key_dtype=np.uint32, ctx=ctx)
    j.add_block(small)
    j.finish_build()
    j.close()


timeit("HashJoin create + add_block + finish + free", join_cycle)


def agg_cycle():
    a = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.uint32), (ch.AGG_COUNT, None)], ctx=ctx)
    a.execute_on_block(small, [small, None])
    a.convert_to_block()
    a.close()


timeit("Aggregator create + add 1024 rows + convert_to_block + free", agg_cycle, reps=200)
print(json.dumps(out))
