"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

What the reference does on one host with threads, and where (file:line in the reference checkout):
  * no-key aggregation: per-thread states folded by mergeWithoutKeyDataImpl (src/Interpreters/Aggregator.cpp:2584-2628)
        -> `merge_without_key`: one 16-byte all-reduce of {sum, count};
  * GROUP BY: per-thread tables, two-level buckets `(crc32c(key) >> 24) & 0xFF` merged bucket-wise
    (src/Common/HashTable/TwoLevelHashTable.h:53, src/Processors/Transforms/AggregatingTransform.cpp:120-136)
        -> `ShardedGroupBy`: each rank pre-aggregates its rows, routes partial states to owner = bucket & (world-1) with ONE
           all-to-all, owners merge (mergeBucketImpl) and keep their shard of the result;
  * parallel_hash join: rows routed to shard `getBucketFromHash(h) & (slots-1)` (src/Interpreters/ConcurrentHashJoin.cpp:426-440,
    538-565) -> `ShardedHashJoin`: build and probe rows are routed by the same rule with an all-to-all each, joined locally.

The exchange is the only collective; xGMI is point-to-point (7 links per GPU), so an all-to-all of hash partitions drives
all links at once.  Everything here is host-side orchestration over a `LocalEngine` (the HIP kernels through the C ABI);
tests substitute a CPU engine to run it under gloo with world_size 2.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

_NP2T = {np.dtype(np.int64): torch.int64, np.dtype(np.uint64): torch.int64, np.dtype(np.uint32): torch.int32,
         np.dtype(np.int32): torch.int32, np.dtype(np.float64): torch.float64, np.dtype(np.uint8): torch.uint8,
         np.dtype(np.uint16): torch.int16, np.dtype(np.int16): torch.int16, np.dtype(np.int8): torch.int8, np.dtype(np.float32): torch.float32}


def world_is_power_of_two(world: int) -> bool:
    return world >= 1 and (world & (world - 1)) == 0


def merge_without_key(states: torch.Tensor, group=None, async_op: bool = False):
    """mergeWithoutKeyDataImpl across ranks: `states` holds 8-byte state words ({sum bits, count}); integer sums wrap."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    return dist.all_reduce(states, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def _backend_is_device(group=None) -> bool:
    return dist.get_backend(group) == "nccl"


def exchange_counts(counts, group=None, device="cpu") -> np.ndarray:
    """counts[r] rows this rank sends to rank r -> rows it receives from every rank."""
    world = dist.get_world_size(group)
    dev = device if _backend_is_device(group) else "cpu"
    send = torch.as_tensor(np.asarray(counts, dtype=np.int64), device=dev)
    recv = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv, send, group=group)
    return recv.cpu().numpy()


def all_to_all_rows(send: torch.Tensor, send_counts, recv_counts, group=None) -> torch.Tensor:
    """Variable-size all-to-all of a 1-D tensor laid out shard after shard (the output of a hash partition)."""
    send_counts = [int(x) for x in send_counts]
    recv_counts = [int(x) for x in recv_counts]
    dev = send.device
    staged = send if _backend_is_device(group) or dev.type == "cpu" else send.cpu()   # gloo moves host memory
    recv = torch.empty(sum(recv_counts), dtype=send.dtype, device=staged.device)
    dist.all_to_all_single(recv, staged.contiguous(), output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
    return recv if recv.device == dev else recv.to(dev)


class LocalEngine:
    """The per-GPU operators the sharded algorithms need.  The default implementation runs the HIP kernels through the
    C ABI on torch CUDA tensors; tests pass their own engine with the same methods."""

    def __init__(self, ctx=None, device_index: int = 0):
        import clickhouse_amd as ch
        self.ch = ch
        self.device = torch.device("cuda", device_index)
        if ctx is None:
            # one stream for both worlds: the C-ABI context launches on a torch stream that is made current, so torch's
            # copies/collectives and the HIP kernels are ordered without host synchronisation
            self.stream = torch.cuda.Stream(device=self.device)
            torch.cuda.set_stream(self.stream)
            ctx = ch.Context(device_index, self.stream.cuda_stream)
        self.ctx = ctx

    # -- tensors <-> device columns (zero copy both ways) ---------------------------------------
    def col(self, t: torch.Tensor, dtype):
        assert t.is_cuda and t.is_contiguous()
        return self.ctx.wrap(t.data_ptr(), dtype, t.shape[0], keepalive=t)

    def tensor(self, col, dtype) -> torch.Tensor:
        """Device column -> torch tensor that owns its memory (one device-to-device copy on the shared stream)."""
        n = col.size()
        out = torch.empty(n, dtype=_NP2T[np.dtype(dtype)], device=self.device)
        if n:
            nbytes = n * np.dtype(dtype).itemsize
            out.view(torch.uint8).copy_(_borrow(col.device_ptr, nbytes, self.device, keepalive=col))
        return out

    def partition_by_hash(self, keys: torch.Tensor, key_dtype, cols, dtypes, n_shards: int):
        kc = self.col(keys, key_dtype)
        outs, counts = self.ch.partition_by_hash(kc, n_shards, [self.col(c, d) for c, d in zip(cols, dtypes)])
        self.ctx.synchronize()
        return [self.tensor(o, d) for o, d in zip(outs, dtypes)], counts

    def Aggregator(self, key_dtype, aggs, size_hint=0):
        return self.ch.Aggregator(key_dtype, aggs, size_hint=size_hint, ctx=self.ctx)

    def HashJoin(self, kind, strictness, key_dtype=np.uint64):
        return self.ch.HashJoin(kind, strictness, key_dtype=key_dtype, ctx=self.ctx)

    def agg_add(self, agg, keys, key_dtype, args, arg_dtypes):
        agg.execute_on_block(self.col(keys, key_dtype), [self.col(a, d) if a is not None else None for a, d in zip(args, arg_dtypes)])

    def agg_export(self, agg, key_dtype):
        keys, words, rows = agg.export_state_columns()
        return self.tensor(keys, key_dtype), [self.tensor(w, np.uint64) for w in words], rows

    def agg_merge_states(self, agg, keys, key_dtype, words):
        agg.merge_states(self.col(keys, key_dtype), [self.col(w, np.uint64) for w in words], keys.shape[0])

    def agg_result(self, agg):
        return agg.convert_to_block()

    def join_add(self, join, keys, key_dtype):
        join.add_block(self.col(keys, key_dtype))

    def join_pairs(self, join, keys, key_dtype):
        return join.joined_pairs(self.col(keys, key_dtype))


class _Borrowed:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can view it without a copy."""

    def __init__(self, ptr, nbytes, keepalive):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        self._keepalive = keepalive


def _borrow(ptr: int, nbytes: int, device, keepalive=None) -> torch.Tensor:
    return torch.as_tensor(_Borrowed(ptr, nbytes, keepalive), device=device)


class ShardedGroupBy:
    """GROUP BY across ranks: local pre-aggregation -> one all-to-all of partial states -> owner-side merge."""

    def __init__(self, engine, key_dtype, aggs, group=None, size_hint: int = 0):
        self.e, self.key_dtype, self.aggs, self.group = engine, np.dtype(key_dtype), aggs, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if not world_is_power_of_two(self.world):
            raise ValueError("the bucket rule `bucket & (world-1)` needs a power-of-two world size (ConcurrentHashJoin.cpp:158)")
        self.local = engine.Aggregator(key_dtype, aggs, size_hint=size_hint)
        self.arg_dtypes = [d for _, d in aggs]

    def add_block(self, keys, args):
        """executeOnBlock on this rank's rows"""
        self.e.agg_add(self.local, keys, self.key_dtype, args, self.arg_dtypes)

    def finish(self):
        """-> (keys, [result columns]) of the groups this rank owns."""
        if self.world == 1:
            return self.e.agg_result(self.local)
        keys, words, rows = self.e.agg_export(self.local, self.key_dtype)       # convertToBlockImplNotFinal
        parts, counts = self.e.partition_by_hash(keys, self.key_dtype, [keys] + words, [self.key_dtype] + [np.uint64] * len(words), self.world)
        recv_counts = exchange_counts(counts, self.group, device=parts[0].device if hasattr(parts[0], "device") else "cpu")
        got = [all_to_all_rows(p, counts, recv_counts, self.group) for p in parts]   # THE exchange step
        owner = self.e.Aggregator(self.key_dtype, self.aggs, size_hint=int(sum(recv_counts)))
        self.e.agg_merge_states(owner, got[0], self.key_dtype, got[1:])          # mergeBucketImpl on the owner
        self.owner = owner
        return self.e.agg_result(owner)


class ShardedHashJoin:
    """parallel_hash across ranks: build rows and probe rows are routed to `bucket(key) & (world-1)`; each rank builds and
    probes its own shard.  Row identities travel with the rows as (origin rank << 40 | origin row) payloads."""

    def __init__(self, engine, kind, strictness, key_dtype=np.uint64, group=None):
        self.e, self.kind, self.strictness, self.key_dtype, self.group = engine, kind, strictness, np.dtype(key_dtype), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if not world_is_power_of_two(self.world):
            raise ValueError("power-of-two world size required")
        self.join = engine.HashJoin(kind, strictness, key_dtype=key_dtype)
        self.build_ids = []

    def _route(self, keys, ids):
        if self.world == 1:
            return keys, ids
        parts, counts = self.e.partition_by_hash(keys, self.key_dtype, [keys, ids], [self.key_dtype, np.int64], self.world)
        recv_counts = exchange_counts(counts, self.group, device=parts[0].device if hasattr(parts[0], "device") else "cpu")
        return tuple(all_to_all_rows(p, counts, recv_counts, self.group) for p in parts)

    def _ids(self, n, like):
        base = torch.arange(n, dtype=torch.int64, device=like.device)
        return base + (self.rank << 40)

    def add_build_rows(self, keys):
        k, ids = self._route(keys, self._ids(keys.shape[0], keys))
        self.e.join_add(self.join, k, self.key_dtype)
        self.build_ids.append(ids)

    def probe(self, keys):
        """-> (left global ids, right global ids or -1) of the joined rows found on this rank's shard."""
        k, lids = self._route(keys, self._ids(keys.shape[0], keys))
        left, rblock, rrow, consumed = self.e.join_pairs(self.join, k, self.key_dtype)
        assert consumed == k.shape[0]
        lids_np = lids.cpu().numpy()
        out_left = lids_np[left]
        out_right = np.full(left.shape[0], -1, dtype=np.int64)
        hit = rblock >= 0
        if hit.any():
            ids = [b.cpu().numpy() for b in self.build_ids]
            sizes = np.array([i.shape[0] for i in ids])
            starts = np.concatenate([[0], np.cumsum(sizes)[:-1]])
            flat = np.concatenate(ids) if ids else np.zeros(0, dtype=np.int64)
            out_right[hit] = flat[starts[rblock[hit]] + rrow[hit]]
        return out_left, out_right
