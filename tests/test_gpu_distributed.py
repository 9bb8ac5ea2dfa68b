"""Two ranks sharing the one GPU of the test box: the REAL per-GPU operators (HIP kernels through the C ABI, via
clickhouse_amd.distributed.LocalEngine) under the sharding orchestration, with gloo as transport (RCCL needs one device per
rank; on the 8-GPU node the same code runs with backend nccl)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, init_file, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import clickhouse_amd as ch
    from clickhouse_amd import distributed as D
    from cpu_engine import GlooExchange

    class Engine(GlooExchange, D.LocalEngine):
        """the real per-GPU operators; only the transport is swapped: partitions are staged through the host and moved by gloo"""

        def __init__(self, ctx):
            D.LocalEngine.__init__(self, ctx, None)

        world = property(lambda self: dist.get_world_size(), lambda self, v: None)
        rank = property(lambda self: dist.get_rank(), lambda self, v: None)

        def all_to_all(self, col, counts, recv_counts):
            return self.ctx.upload(self._a2a_host(col.numpy(), counts, recv_counts))

    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        ctx = ch.Context(0)
        eng = Engine(ctx)
        rng = np.random.Generator(np.random.PCG64(321))
        n = 400_000
        keys_all = rng.integers(0, 50_000, size=n, dtype=np.uint64)
        vals_all = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)
        lo, hi = rank * n // world, (rank + 1) * n // world

        g = D.ShardedGroupBy(eng, np.uint64, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)])
        g.add_block(ctx.upload(keys_all[lo:hi]), [ctx.upload(vals_all[lo:hi]), None])
        k, (s, c) = g.finish()
        sel = ch.hash_to_selector(ctx.upload(k), world).numpy()
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

        nb, npb = 90_000, 200_000
        bk_all = rng.integers(0, 30_000, size=nb, dtype=np.uint64)
        bv_all = rng.integers(-2**50, 2**50, size=nb, dtype=np.int64)
        pk_all = rng.integers(0, 40_000, size=npb, dtype=np.uint64)
        j = D.ShardedHashJoin(eng, ch.JOIN_INNER, ch.STRICT_ALL)
        blo, bhi = rank * nb // world, (rank + 1) * nb // world
        plo, phi = rank * npb // world, (rank + 1) * npb // world
        for b in range(blo, bhi, 20_000):  # several right Blocks per rank: flat row ordinals over the routed blocks
            e = min(bhi, b + 20_000)
            j.add_build_rows(ctx.upload(bk_all[b:e]), [ctx.upload(np.arange(b, e, dtype=np.int64)), ctx.upload(bv_all[b:e])])
        j.finish_build()
        n_out, left, right = j.probe(ctx.upload(pk_all[plo:phi]), [ctx.upload(np.arange(plo, phi, dtype=np.int64))])
        lk, lid, rid, rv = left[0].numpy(), left[1].numpy(), right[0].numpy(), right[1].numpy()   # results are device columns
        assert lk.shape[0] == n_out == rid.shape[0]
        assert (ch.hash_to_selector(ctx.upload(lk), world).numpy() == rank).all() if n_out else True
        cnt, sm = j.probe_count_sum(ctx.upload(pk_all[plo:phi]), payload_index=1)
        gathered = [None] * world
        dist.all_gather_object(gathered, (lid, rid, rv))
        if rank == 0:
            got = np.concatenate([np.stack([l, r], axis=1) for l, r, _ in gathered])
            for _, r, v in gathered:
                assert np.array_equal(v, bv_all[r])              # the payload travelled with its row and was gathered on the device
            order = np.argsort(bk_all, kind="stable")
            sk = bk_all[order]
            lo_i = np.searchsorted(sk, pk_all, "left")
            hi_i = np.searchsorted(sk, pk_all, "right")
            want_rows = int((hi_i - lo_i).sum())
            assert got.shape[0] == want_rows
            assert (pk_all[got[:, 0]] == bk_all[got[:, 1]]).all()
            assert len({(int(a), int(b)) for a, b in got}) == want_rows
            assert cnt == want_rows and sm == int(bv_all[got[:, 1]].astype(np.uint64).sum(dtype=np.uint64))
        # BASELINE.json configs[4] sharded, real kernels: the join chain + gather on this rank's lineorder rows, dimensions replicated, the
        # (year, nation) partial states exchanged and merged by their owners; the union of the ranks' groups == the oracle plan over all rows
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import oracle as O
        import ssb
        dims = ssb.gen_dims(300_000, 20_000, 20_000)
        lo_all = ssb.gen_lineorder_numpy(2_400_000, 300_000, 20_000, 20_000)
        rlo, rhi = rank * 2_400_000 // world, (rank + 1) * 2_400_000 // world
        mine = ssb.q41_sharded_gpu(ch, D, eng, dims, {k: ctx.upload(v[rlo:rhi]) for k, v in lo_all.items()})
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        if rank == 0:
            want = ssb.q41_cpu(O, dims, lo_all)
            union = {}
            for g_ in gathered:
                assert not (set(g_) & set(union)), "a group lives on two ranks"
                union.update(g_)
            assert union == want and len(want) == 35
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
