"""A CPU stand-in for clickhouse_amd.distributed.LocalEngine, built on the oracle and gloo: lets the multi-rank orchestration
(exchange of counts, all-to-all of hash partitions, owner-side merge, routed build / probe) run with world_size 2 without a GPU.
"Columns" are numpy arrays.  Test infrastructure.  GlooExchange is the transport half alone: tests/test_gpu_distributed.py mixes it
into the real LocalEngine to run the HIP kernels under two ranks that share one GPU (RCCL needs a device per rank)."""
import numpy as np
import torch
import torch.distributed as dist

import oracle as O


class GlooExchange:
    """exchange_counts / all_to_all / all_reduce_u64 over torch.distributed (gloo), host arrays in and out"""

    @property
    def world(self):
        return dist.get_world_size()

    @property
    def rank(self):
        return dist.get_rank()

    def exchange_counts(self, counts):
        send = torch.as_tensor(np.asarray(counts, dtype=np.int64))
        recv = torch.empty(self.world, dtype=torch.int64)
        dist.all_to_all_single(recv, send)
        return [int(x) for x in recv]

    def _a2a_host(self, arr: np.ndarray, counts, recv_counts) -> np.ndarray:
        raw = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1))
        es = arr.dtype.itemsize
        recv = torch.empty(int(sum(recv_counts)) * es, dtype=torch.uint8)
        dist.all_to_all_single(recv, raw, output_split_sizes=[int(c) * es for c in recv_counts], input_split_sizes=[int(c) * es for c in counts])
        return recv.numpy().view(arr.dtype)

    def exchange(self, parts, counts):
        """the engine's one-exchange-per-Block entry: counts first, then every column"""
        recv_counts = self.exchange_counts(counts)
        return [self.all_to_all(p, counts, recv_counts) for p in parts], recv_counts

    def all_reduce_u64(self, values):
        t = torch.from_numpy(np.array([int(v) % 2**64 for v in values], dtype=np.uint64).view(np.int64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)  # two's complement: the wrap-around sum
        return [int(x) for x in t.numpy().view(np.uint64)]


class _CpuAgg:
    def __init__(self, key_dtype, aggs):
        assert all(k in (O.AGG_SUM, O.AGG_COUNT) for k, _ in aggs), "CPU test engine: sum/count states only"
        self.key_dtype, self.aggs = np.dtype(key_dtype), aggs
        self.rows = O.Aggregator(key_dtype, aggs)
        # merged partial states are sums of 8-byte words whatever the function
        self.states = O.Aggregator(key_dtype, [(O.AGG_SUM, np.uint64)] * len(aggs))

    def result(self):
        k1, r1 = self.rows.convert_to_block()
        k2, r2 = self.states.convert_to_block()
        keys = np.concatenate([k1, k2])
        uk, inv = np.unique(keys, return_inverse=True)
        out = []
        for j in range(len(self.aggs)):
            acc = np.zeros(uk.shape[0], dtype=np.uint64)
            np.add.at(acc, inv, np.concatenate([r1[j].view(np.uint64), r2[j].view(np.uint64)]))
            out.append(acc.view(r1[j].dtype))
        return uk.astype(self.key_dtype), out


class _CpuJoin:
    def __init__(self, kind, strictness):
        self.j = O.HashJoin(kind, strictness)
        self.block_rows = []


class CpuEngine(GlooExchange):
    def partition_by_hash(self, keys, cols, n_shards):
        sel = O.hash_to_selector(np.ascontiguousarray(keys), n_shards)
        order = np.argsort(sel, kind="stable")  # stable within a shard, like chgpu_partition_by_hash
        counts = np.bincount(sel.astype(np.int64), minlength=n_shards).astype(np.uint64)
        return [np.ascontiguousarray(c[order]) for c in cols], [int(c) for c in counts]

    def all_to_all(self, col, counts, recv_counts):
        return self._a2a_host(col, counts, recv_counts)

    def Aggregator(self, key_dtype, aggs, size_hint=0):
        return _CpuAgg(key_dtype, aggs)

    def HashJoin(self, kind, strictness, key_dtype=np.uint64):
        return _CpuJoin(kind, strictness)

    def agg_add(self, agg, keys, args):
        agg.rows.execute_on_block(keys, args)

    def agg_export(self, agg):
        k, res = agg.result()
        return k, [np.ascontiguousarray(r).view(np.uint64) for r in res], k.shape[0]

    def agg_merge_states(self, agg, keys, words):
        agg.states.execute_on_block(keys, list(words))

    def agg_result(self, agg):
        return agg.result()

    def agg_finalize(self, agg):
        k, res = agg.result()
        return k, res, k.shape[0]

    def to_column(self, x, dtype):
        return np.ascontiguousarray(np.asarray(x) if dtype is None else np.asarray(x, dtype=dtype))

    def cut(self, col, begin, rows):
        return col[begin:begin + rows]

    def rows(self, col):
        return col.shape[0]

    def join_add(self, join, keys):
        join.j.add_block(keys.astype(np.uint64))
        join.block_rows.append(keys.shape[0])

    def join_finish(self, join):
        pass

    def concat(self, cols):
        return np.concatenate(cols) if cols else None

    def _flat(self, join, blk, row):
        starts = np.concatenate([[0], np.cumsum(join.block_rows)[:-1]]).astype(np.int64)
        return starts[blk] + row

    def join_count_sum(self, join, keys, payload):
        _, blk, row, consumed = join.j.joined_pairs(keys.astype(np.uint64))
        assert consumed == keys.shape[0]
        hit = blk >= 0
        s = int(payload[self._flat(join, blk[hit], row[hit])].astype(np.uint64).sum(dtype=np.uint64)) if payload is not None else 0
        return blk.shape[0], s

    def join_materialize(self, join, keys, left_cols, right_cols):
        left, blk, row, consumed = join.j.joined_pairs(keys.astype(np.uint64))
        assert consumed == keys.shape[0]
        hit = blk >= 0
        flat = np.where(hit, self._flat(join, np.where(hit, blk, 0), np.where(hit, row, 0)), 0)
        right = [np.where(hit, c[flat], 0).astype(c.dtype) for c in right_cols]
        return left.shape[0], [c[left] for c in left_cols], right
