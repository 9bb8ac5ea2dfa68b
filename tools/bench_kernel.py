#!/usr/bin/env python3
"""Quick product-level timing of the fused filter+sum kernel through the C ABI (no CPU baseline): GB/s over N rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import clickhouse_amd as ch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
two = len(sys.argv) > 3 and sys.argv[3] == "two"
dev = torch.device("cuda", 0)
a = torch.randint(0, 2**31, (n,), dtype=torch.int64, device=dev)
b = torch.randint(0, 2**31, (n,), dtype=torch.int64, device=dev) if two else None
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)
res = torch.zeros(2, dtype=torch.int64, device=dev)
col = ctx.wrap(a.data_ptr(), np.int64, n, keepalive=a)
colb = ctx.wrap(b.data_ptr(), np.int64, n, keepalive=b) if two else None
slot = ctx.wrap(res.data_ptr(), np.uint64, 2, keepalive=res)
for _ in range(3):
    ch.filter_sum_async(colb if two else col, ch.LT, 214748365, col if two else None, slot)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for s, e in ev:
    s.record(st); ch.filter_sum_async(colb if two else col, ch.LT, 214748365, col if two else None, slot); e.record(st)
torch.cuda.synchronize()
ms = sorted(s.elapsed_time(e) for s, e in ev)
bytes_ = 8.0 * n * (2 if two else 1)
print(f"rows={n} cols={'2' if two else '1'} med={ms[len(ms)//2]:.4f} ms min={ms[0]:.4f} ms  {bytes_/ms[len(ms)//2]/1e6:.1f} GB/s (med)  {bytes_/ms[0]/1e6:.1f} GB/s (best)")
