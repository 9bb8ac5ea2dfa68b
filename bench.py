#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X: rows/sec (+ achieved HBM GB/s) of the block-processing hot path.

Workload at every N: BASELINE.json configs[1] — `SELECT sum(a), count() FROM t WHERE a < 214748365` (~10 % pass) over
1 000 000 000 Int64 rows per GPU, uniform [0, 2^31), resident in HBM before the timed region (synthetic, generated on
device).  One "step" = one pass of the fused HIP filter+sum kernel over the whole column plus the no-key state merge
(mergeWithoutKeyDataImpl): at N>1 every rank scans its own 1 B rows (weak scaling, no data-path collective) and the
16-byte {sum,count} states are summed with one RCCL all-reduce per step, issued asynchronously.

Contract: `python bench.py --gpus N --steps K --warmup W`; N>1 is launched by torch.distributed.run (one rank per GPU).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

THRESHOLD = 214748365  # ~10.0 % of uniform [0, 2^31)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000_000, help="rows per GPU (default: the 1 B rows of configs[1])")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even at world size 1 (exercises the RCCL code path on one GPU)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the measured configuration) or gloo (rehearsal of the N>1 code path on one GPU)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import clickhouse_amd as ch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if args.backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())  # rehearsal only: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    _saved_stdout_fd = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # stdout carries exactly one JSON line: RCCL prints its version banner to stdout when the communicator is created
        # (NCCL_DEBUG=VERSION is exported on the GPU boxes), so fd 1 points at stderr until the result is printed
        sys.stdout.flush()
        _saved_stdout_fd = os.dup(1)
        os.dup2(2, 1)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    n = args.rows
    K, W = args.steps, args.warmup

    # ---- synthetic input, resident in HBM ----
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    a = torch.randint(0, 2**31, (n,), dtype=torch.int64, device=dev, generator=g)
    results = torch.zeros((K + W, 2), dtype=torch.int64, device=dev)  # one {sum, count} state slot per step
    torch.cuda.synchronize()

    # the C-ABI context launches on this torch stream so the HIP events below bracket exactly its kernels
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx = ch.Context(local_rank, stream.cuda_stream)
    col = ctx.wrap(a.data_ptr(), np.int64, n, keepalive=a)
    slots = [ctx.wrap(results[i].data_ptr(), np.uint64, 2, keepalive=results) for i in range(K + W)]

    works = []

    def merge_states(i):
        """mergeWithoutKeyDataImpl across ranks: one 16-byte all-reduce, asynchronous under RCCL"""
        if dist is None:
            return
        if args.backend == "nccl":
            works.append(dist.all_reduce(results[i], op=dist.ReduceOp.SUM, async_op=True))
        else:  # gloo rehearsal: stage through the host
            stream.synchronize()
            h = results[i].cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            results[i].copy_(h)

    def step(i):
        ch.filter_sum_async(col, ch.LT, THRESHOLD, None, slots[i])  # HIP kernels via the C ABI, no host sync
        merge_states(i)

    for i in range(W):
        step(i)
    for w in works:
        w.wait()
    works.clear()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        ev[i][0].record(stream)
        ch.filter_sum_async(col, ch.LT, THRESHOLD, None, slots[W + i])
        ev[i][1].record(stream)
        merge_states(W + i)
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    kern_ms = [s.elapsed_time(e) for s, e in ev]  # HIP events on the launch stream: filter+sum kernel (+ its 1-block finish)
    kern_avg_ms = sum(kern_ms) / len(kern_ms)

    if dist is not None:
        red_dev = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, kern_avg_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # MAX over ranks
        elapsed, kern_avg_ms = float(t[0].item()), float(t[1].item())

    # ---- sanity: every step produced the same, correct state (checked outside the timed region) ----
    res = results.cpu().numpy()
    assert (res == res[W]).all(), "steps disagree"
    if world == 1:
        want_sum = int(a[a < THRESHOLD].sum().item())
        want_cnt = int((a < THRESHOLD).sum().item())
        assert (int(res[W][0]), int(res[W][1])) == (want_sum, want_cnt), (res[W], want_sum, want_cnt)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_rows = n * world * K
    value = total_rows / elapsed
    algo_bytes = 8.0 * n  # 8 B/row, one launch scans the rank's whole column
    achieved = algo_bytes / (kern_avg_ms * 1e-3) / 1e9

    traffic = None
    tpath = os.path.join(REPO, "profiles", "traffic.json")  # PMC-derived bytes/launch recorded by profiles/collect.sh
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("rows") == n:
                traffic = tj.get("k_filter_sum_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": json.load(open(os.path.join(REPO, "BASELINE.json")))["metric"],
        "value": value,
        "unit": "rows/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE.json configs[1]: SELECT sum(a), count() WHERE a < 214748365 (~10% pass) over Int64 rows, "
                        "HBM-resident, fused HIP filter+sum kernel",
            "rows_per_gpu": n,
            "global_rows_per_step": n * world,
            "parallelism": f"row-range shards x{world}, 16-byte state all-reduce" if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "kernel": "k_filter_sum<long,2,true,false,IntRangePred>",
            "kernel_avg_ms": kern_avg_ms,
            "algorithmic_bytes_per_launch": algo_bytes,
        },
    }

    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a, args.cpu_sample_rows, ctx, ch)

    if _saved_stdout_fd is not None:
        sys.stdout.flush()
        os.dup2(_saved_stdout_fd, 1)  # the real stdout is back for the one JSON line
        os.close(_saved_stdout_fd)
    print(json.dumps(out), flush=True)
    if dist is not None:
        os.dup2(2, 1)  # anything RCCL says while shutting down goes to stderr again
        dist.destroy_process_group()


def cpu_baseline(a, sample_rows, ctx, ch):
    """The reference CPU path restated (oracle/, kind 'port') timed on this box's host cores over a bounded sample of the
    same column: per-Block (65 409 rows) compare -> UInt8 mask -> countBytesInFilter -> IColumn::filter -> sum addMany.
    The oracle is only the thing timed/compared here, never part of the GPU path."""
    import numpy as np

    import oracle

    oracle.build()
    m = min(sample_rows, a.shape[0])
    host = a[:m].cpu().numpy()
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    best1, bestN = None, None
    r1 = rN = None
    for _ in range(5):
        t0 = time.perf_counter()
        r1 = oracle.filter_sum_pipeline(host, oracle.LT, THRESHOLD, threads=1)
        dt = time.perf_counter() - t0
        best1 = dt if best1 is None else min(best1, dt)
    for _ in range(20):
        t0 = time.perf_counter()
        rN = oracle.filter_sum_pipeline(host, oracle.LT, THRESHOLD, threads=cores)
        dt = time.perf_counter() - t0
        bestN = dt if bestN is None else min(bestN, dt)
    # parity on the sample: the HIP path must give the CPU path's answer bit for bit
    s, c = ch.filter_sum(ctx.wrap(a.data_ptr(), np.int64, m, keepalive=a), ch.LT, THRESHOLD)
    assert (int(s), c) == (int(r1[0]), r1[1]) == (int(rN[0]), rN[1]), "GPU result differs from the CPU restatement"
    return {
        "value": m / bestN,
        "unit": "rows/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {m} rows of the same column, Blocks of 65409 rows, best of 20 ({cores} threads) ; "
                  f"single thread: {m / best1:.4g} rows/s (best of 5); about 7 s of CPU work in all",
        "single_thread_value": m / best1,
    }


if __name__ == "__main__":
    main()
