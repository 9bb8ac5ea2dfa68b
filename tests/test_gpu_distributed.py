"""Two ranks sharing the one GPU of the test box: the REAL per-GPU operators (HIP kernels through the C ABI, via
clickhouse_amd.distributed.LocalEngine) under the sharding orchestration, with gloo as transport (RCCL needs one device per
rank; on the 8-GPU node the same code runs with backend nccl)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, init_file, out_dir):
    sys.path.insert(0, REPO)
    import clickhouse_amd as ch
    from clickhouse_amd import distributed as D

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        eng = D.LocalEngine(device_index=0)
        rng = np.random.Generator(np.random.PCG64(321))
        n = 400_000
        keys_all = rng.integers(0, 50_000, size=n, dtype=np.uint64)
        vals_all = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)
        lo, hi = rank * n // world, (rank + 1) * n // world
        dev = torch.device("cuda", 0)

        g = D.ShardedGroupBy(eng, np.uint64, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)])
        g.add_block(torch.from_numpy(keys_all[lo:hi].view(np.int64)).to(dev), [torch.from_numpy(vals_all[lo:hi]).to(dev), None])
        k, (s, c) = g.finish()
        sel = ch.hash_to_selector(eng.ctx.upload(k), world).numpy()
        assert (sel == rank).all()
        gathered = [None] * world
        dist.all_gather_object(gathered, (k, s, c))
        if rank == 0:
            kk = np.concatenate([x[0] for x in gathered])
            ss = np.concatenate([x[1] for x in gathered])
            cc = np.concatenate([x[2] for x in gathered])
            uk, inv = np.unique(keys_all, return_inverse=True)
            ws = np.zeros(uk.shape[0], dtype=np.int64)
            np.add.at(ws, inv, vals_all)
            order = np.argsort(kk)
            assert np.array_equal(kk[order], uk) and np.array_equal(ss[order], ws) and np.array_equal(cc[order], np.bincount(inv).astype(np.uint64))

        bk_all = rng.integers(0, 30_000, size=90_000, dtype=np.uint64)
        pk_all = rng.integers(0, 40_000, size=200_000, dtype=np.uint64)
        j = D.ShardedHashJoin(eng, ch.JOIN_INNER, ch.STRICT_ALL)
        blo, bhi = rank * 90_000 // world, (rank + 1) * 90_000 // world
        plo, phi = rank * 200_000 // world, (rank + 1) * 200_000 // world
        j.add_build_rows(torch.from_numpy(bk_all[blo:bhi].view(np.int64)).to(dev))
        left, right = j.probe(torch.from_numpy(pk_all[plo:phi].view(np.int64)).to(dev))
        gathered = [None] * world
        dist.all_gather_object(gathered, (left, right))
        if rank == 0:
            def glob(ids, total):
                return (ids >> 40) * (total // world) + (ids & ((1 << 40) - 1))
            got = np.concatenate([np.stack([glob(l, 200_000), glob(r, 90_000)], axis=1) for l, r in gathered])
            order = np.argsort(bk_all, kind="stable")
            sk = bk_all[order]
            lo_i = np.searchsorted(sk, pk_all, "left")
            hi_i = np.searchsorted(sk, pk_all, "right")
            want_rows = int((hi_i - lo_i).sum())
            assert got.shape[0] == want_rows
            assert (pk_all[got[:, 0]] == bk_all[got[:, 1]]).all()
            assert len({(int(a), int(b)) for a, b in got}) == want_rows
        dist.barrier()
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_real_kernels_under_sharding():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, os.path.join(d, "rdv"), d), nprocs=world, join=True)
        assert all(os.path.exists(os.path.join(d, f"ok{r}")) for r in range(world))
