"""N>1 path on CPU: world_size-2 gloo runs of the sharding orchestration in clickhouse_amd/distributed.py (the local
operators are replaced by the oracle-backed CPU engine; on the GPU box the same code drives the HIP kernels)."""
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
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle as O
    from clickhouse_amd import distributed as D
    from cpu_engine import CpuEngine

    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        eng = CpuEngine()
        rng = np.random.Generator(np.random.PCG64(123))
        n = 40_000
        keys_all = rng.integers(0, 5000, size=n, dtype=np.uint64)
        keys_all[:5] = 0
        vals_all = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)
        lo, hi = rank * n // world, (rank + 1) * n // world

        # 1. no-key states: mergeWithoutKeyDataImpl == one all-reduce (integer sum wraps modulo 2^64)
        local = vals_all[lo:hi]
        got = eng.all_reduce_u64([int(local[local < 0].sum()), int((local < 0).sum())])
        want = [int(vals_all[vals_all < 0].sum()) % 2**64, int((vals_all < 0).sum())]
        assert got == want

        # 2. sharded GROUP BY: rows split by range, partial states routed to owner = bucket & (world-1)
        g = D.ShardedGroupBy(eng, np.uint64, [(O.AGG_SUM, np.int64), (O.AGG_COUNT, None)])
        for b in range(lo, hi, 7001):
            e = min(hi, b + 7001)
            g.add_block(keys_all[b:e], [vals_all[b:e], None])
        k, (s, c) = g.finish()
        owners = O.hash_to_selector(np.ascontiguousarray(k), world)
        assert (owners == rank).all(), "a rank holds groups it does not own"
        gathered = [None] * world
        dist.all_gather_object(gathered, (k, s, c))
        if rank == 0:
            kk = np.concatenate([x[0] for x in gathered])
            ss = np.concatenate([x[1] for x in gathered])
            cc = np.concatenate([x[2] for x in gathered])
            assert len(set(kk.tolist())) == kk.shape[0], "a group lives on two ranks"
            uk, inv = np.unique(keys_all, return_inverse=True)
            ws = np.zeros(uk.shape[0], dtype=np.int64)
            np.add.at(ws, inv, vals_all)
            order = np.argsort(kk)
            assert np.array_equal(kk[order], uk) and np.array_equal(ss[order].view(np.int64), ws)
            assert np.array_equal(cc[order], np.bincount(inv).astype(np.uint64))

        # 3. sharded hash join (parallel_hash routing): build rows travel with their payload (here: the global build row number and a
        #    value), probe rows with their global row number; the union of the per-rank results == the single-node join
        bk_all = rng.integers(0, 3000, size=9000, dtype=np.uint64)
        bv_all = rng.integers(-2**50, 2**50, size=9000, dtype=np.int64)
        pk_all = rng.integers(0, 4000, size=20_000, dtype=np.uint64)
        j = D.ShardedHashJoin(eng, O.JOIN_INNER, O.STRICT_ALL)
        blo, bhi = rank * 9000 // world, (rank + 1) * 9000 // world
        plo, phi = rank * 20_000 // world, (rank + 1) * 20_000 // world
        for b in range(blo, bhi, 2000):  # the build side arrives Block by Block
            e = min(bhi, b + 2000)
            j.add_build_rows(bk_all[b:e], [np.arange(b, e, dtype=np.int64), bv_all[b:e]])
        j.finish_build()
        n_out, left, right = j.probe(pk_all[plo:phi], [np.arange(plo, phi, dtype=np.int64)])
        assert left[0].shape[0] == n_out == right[0].shape[0]
        assert (O.hash_to_selector(np.ascontiguousarray(left[0]), world) == rank).all(), "joined rows must stay on the rank that owns their key"
        cnt, sm = j.probe_count_sum(pk_all[plo:phi], payload_index=1)   # the fused form: global count(), sum(bv)
        gathered = [None] * world
        dist.all_gather_object(gathered, (left[1], right[0], right[1]))
        if rank == 0:
            pairs = set()
            for l, r_, v in gathered:
                pairs |= set(zip(l.tolist(), r_.tolist()))
                assert np.array_equal(v, bv_all[r_])
            one = O.HashJoin(O.JOIN_INNER, O.STRICT_ALL)
            one.add_block(bk_all)
            ol, ob, orow, _ = one.joined_pairs(pk_all)
            assert pairs == set(zip(ol.tolist(), orow.tolist()))
            assert sum(x[0].shape[0] for x in gathered) == ol.shape[0]
            assert cnt == ol.shape[0] and sm == int(bv_all[orow].astype(np.uint64).sum(dtype=np.uint64))
        # 4. BASELINE.json configs[4] sharded: lineorder rows split by range, dimensions replicated, every rank runs the SSB Q4.1 plan over
        #    its rows, the (year, nation) partial states meet at their owners -- the union of the ranks' groups == the one-process plan
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import ssb
        dims = ssb.gen_dims(30_000, 2_000, 2_000)
        lo_all = ssb.gen_lineorder_numpy(200_000, 30_000, 2_000, 2_000)
        rlo, rhi = rank * 200_000 // world, (rank + 1) * 200_000 // world
        mine = ssb.q41_sharded_cpu(O, D, eng, dims, {k: v[rlo:rhi] for k, v in lo_all.items()})
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        if rank == 0:
            want = ssb.q41_cpu(O, dims, lo_all)
            union = {}
            for g_ in gathered:
                assert not (set(g_) & set(union)), "a group lives on two ranks"
                union.update(g_)
            assert union == want and len(want) > 20
            assert all(len(g_) > 0 for g_ in gathered), "every rank should own some of the 35 groups"
        dist.barrier()
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_gloo_sharded_groupby_join_and_state_merge():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "rendezvous")
        mp.spawn(_worker, args=(world, init_file, d), nprocs=world, join=True)
        assert all(os.path.exists(os.path.join(d, f"ok{r}")) for r in range(world))


@pytest.mark.timeout(300)
def test_world4_gloo_sharded_groupby_join_and_state_merge():
    """the same orchestration with four ranks (the owner rule, the count exchange and the all-to-all see more than one peer)"""
    world = 4
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "init")
        mp.spawn(_worker, args=(world, init_file, d), nprocs=world, join=True)
        assert all(os.path.exists(os.path.join(d, f"ok{r}")) for r in range(world))


def test_non_power_of_two_world_is_rejected():
    from clickhouse_amd import distributed as D
    assert D.world_is_power_of_two(8) and not D.world_is_power_of_two(6)
