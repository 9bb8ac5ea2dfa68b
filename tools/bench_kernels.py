#!/usr/bin/env python3
"""Exercise every operator once or twice at a large size so `rocprofv3 --kernel-trace --stats` yields per-kernel durations;
`tools/kernel_roofline.py` then turns the stats CSV into a roofline table.  usage: bench_kernels.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import clickhouse_amd as ch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randint(0, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
b = torch.randint(0, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
k32 = torch.randint(0, 1_000_000, (rows,), dtype=torch.int32, device=dev, generator=g)
ac = ctx.wrap(a.data_ptr(), np.int64, rows, keepalive=a)
bc = ctx.wrap(b.data_ptr(), np.int64, rows, keepalive=b)
kc = ctx.wrap(k32.data_ptr(), np.uint32, rows, keepalive=k32)
k64 = (k32 % 1000).to(torch.int64)
k64c = ctx.wrap(k64.data_ptr(), np.int64, rows, keepalive=k64)
THR = 214748365
for rep in range(2):
    ch.filter_sum(ac, ch.LT, THR)                       # k_filter_sum 1 column
    ch.filter_sum(bc, ch.LT, THR, ac)                   # k_filter_sum 2 columns
    mask = ch.cmp_const(ac, ch.LT, THR)                 # k_cmp_mask
    ch.count_bytes_in_filter(mask)                      # k_filter_sum<u8>
    out = ac.filter(mask)                               # k_mask_chunk_counts, scan, k_filter_scatter
    ch.sum_add_many(out)
    sel = ch.hash_to_selector(ac, 8)                    # k_selector
    wh = ac.get_weak_hash32()                           # k_weak_hash32
    parts, counts = ch.partition_by_hash(ac, 8, [ac, bc])   # k_selector, k_part_hist, scan, k_part_scatter
    idx = ctx.upload(np.random.default_rng(1).integers(0, rows, size=rows // 4, dtype=np.uint64))
    ac.index(idx)                                       # k_index (random gather)
    offs = ctx.upload(np.cumsum(np.random.default_rng(2).integers(0, 3, size=rows // 8)).astype(np.uint64))
    ac.cut(0, rows // 8).replicate(offs)                # k_replicate
    A = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], size_hint=1_000_000, ctx=ctx)
    A.execute_on_block(kc, [ac, None])                  # k_gb_hist, k_gb_scatter, k_agg_part_lds
    A.convert_to_block()                                # k_occupied_mask + filter kernels
    B = ch.Aggregator(np.int64, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], ctx=ctx)
    B.execute_on_block(k64c, [ac, None])                # k_agg_part_lds<u64> in RANGE mode (1000 groups, LDS-staged)
    dim = ctx.upload(np.arange(1, 400_001, dtype=np.uint32) * 5)
    sj = ch.HashJoin(ch.JOIN_LEFT, ch.STRICT_SEMI, key_dtype=np.uint32, ctx=ctx)
    sj.add_block(dim)
    sj.finish_build()
    sj.probe_columns(kc, need_right_rows=False)         # k_join_probe_filter<true> (dense prefilter bitmap)
    D = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], size_hint=100_000, ctx=ctx)
    D.execute_on_block(kc, [ac, None], 0, 2_000_000)    # k_agg_rows_direct
    nb = 10_000_000
    j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, ctx=ctx)
    au = ctx.wrap(a.data_ptr(), np.uint64, rows, keepalive=a)
    bu = ctx.wrap(b.data_ptr(), np.uint64, rows, keepalive=b)
    j.add_block(au.cut(0, nb))
    j.finish_build()                                    # k_join_stage_keys, k_join_insert, scan, k_join_fill, k_join_root_first
    r = j.probe_columns(bu.cut(0, rows // 4))           # k_join_probe_count, scan, k_join_cut, k_join_emit
    od = [ctx.wrap(t.data_ptr(), np.uint32, rows * 2, keepalive=t) for t in (a, b)]
    ch.expr_filter_sum(od, [(0, ch.GE, 5), (1, ch.LT, 2**31)], ch.VAL_MUL, 0, 1)  # k_expr_filter_sum<u32>: 2 columns of 2*rows UInt32
ctx.synchronize()
print("done", rows)
