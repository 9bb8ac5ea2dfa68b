#!/usr/bin/env python3
"""Host -> HBM upload rate of chgpu_col_upload (pageable source, hipMemcpyAsync + sync) next to a pinned torch copy: on the MI355X boxes of this
pool both reach ~56 GB/s, i.e. the runtime's pageable path already saturates the link and a pinned staging ring would add nothing."""
import time, numpy as np, sys
sys.path.insert(0, '.')
import clickhouse_amd as ch
ctx = ch.Context(0)
a = np.ones(1 << 28, dtype=np.int64)  # 2 GiB
for _ in range(3):
    t0 = time.perf_counter(); c = ctx.upload(a); ctx.synchronize(); dt = time.perf_counter() - t0
    print("pageable upload GB/s", a.nbytes / dt / 1e9); del c
import torch
p = torch.empty(1 << 28, dtype=torch.int64).pin_memory()
d = torch.empty(1 << 28, dtype=torch.int64, device="cuda")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(p, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pinned copy GB/s", p.numel() * 8 / dt / 1e9)
