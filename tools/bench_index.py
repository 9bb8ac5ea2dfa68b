import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import clickhouse_amd as ch
ctx = ch.Context(0)
dev = torch.device('cuda:0')
for rows, lim in ((400_000_000, 100_000_000), (1_000_000, 100_000_000), (30_000_000, 100_000_000)):
    g = torch.Generator(device=dev).manual_seed(1)
    data = torch.randint(0, 2**40, (rows,), dtype=torch.int64, device=dev, generator=g)
    idx = torch.randint(0, rows, (lim,), dtype=torch.int64, device=dev, generator=g)
    dc = ctx.wrap(data.data_ptr(), np.int64, rows, keepalive=data); ic = ctx.wrap(idx.data_ptr(), np.uint64, lim, keepalive=idx)
    best = 1e9
    for _ in range(4):
        ctx.synchronize(); t0 = time.perf_counter(); o = dc.index(ic); ctx.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    want = data[idx[:1000]].cpu().numpy()
    assert np.array_equal(o.numpy()[:1000], want)
    print(rows, lim, round(best, 3), "ms", round(lim / best / 1e6, 1), "G gathers/s")
    del o
