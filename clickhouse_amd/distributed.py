"""Multi-GPU sharding of the hot path: one process per GPU, the exchange over RCCL (xGMI) through the C ABI (chgpu_comm_*).

What the reference does on one host with threads, and where (file:line in the reference checkout):
  * no-key aggregation: per-thread states folded by mergeWithoutKeyDataImpl (src/Interpreters/Aggregator.cpp:2584-2628)
        -> `Engine.all_reduce_u64`: one 16-byte all-reduce of {sum, count};
  * GROUP BY: per-thread tables, two-level buckets `(crc32c(key) >> 24) & 0xFF` merged bucket-wise
    (src/Common/HashTable/TwoLevelHashTable.h:53, src/Processors/Transforms/AggregatingTransform.cpp:120-136)
        -> `ShardedGroupBy`: each rank pre-aggregates its rows, routes partial states to owner = bucket & (world-1) with ONE
           all-to-all, owners merge (mergeBucketImpl) and keep their shard of the result;
  * parallel_hash join: rows routed to shard `getBucketFromHash(h) & (slots-1)` (src/Interpreters/ConcurrentHashJoin.cpp:426-440,
    538-565) -> `ShardedHashJoin`: build and probe rows are routed by the same rule with an all-to-all each, joined where they
    land; joined rows stay on the owner (the next operator consumes them there), only scalar aggregates are all-reduced.

These classes are the Python mirror of GpuShardedAggregator / GpuConcurrentHashJoin (host/chgpu_shim.hpp) and contain only
orchestration: every data step is an `Engine` method.  `LocalEngine` runs them on device columns through the C ABI -- partition
kernels, chgpu_all_to_all over RCCL, aggregation / join kernels -- with no host round trip between routing and result.  Tests
substitute an engine of the same shape built on the CPU oracle and gloo to run the orchestration with world_size 2 on CPU.
The exchange is the only collective; xGMI is point-to-point (7 links per GPU), so an all-to-all of hash partitions drives all
links at once.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K


def world_is_power_of_two(world: int) -> bool:
    return world >= 1 and (world & (world - 1)) == 0


class Comm:
    """chgpu_comm: one rank of the node's exchange, bound to one Context (device + stream)."""

    def __init__(self, ctx, rank: int, world: int, unique_id: bytes):
        assert len(unique_id) == 128
        self.ctx, self.rank, self.world = ctx, rank, world
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        K.check(K.lib().chgpu_comm_init(ctx._h, rank, world, buf, C.byref(h)))
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        K.check(K.lib().chgpu_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_env(cls, ctx, rank: int, world: int, broadcast):
        """broadcast(obj_or_None) -> obj: the host pipeline's control plane (rank 0 passes the id, the others None)"""
        uid = broadcast(cls.unique_id() if rank == 0 else None)
        return cls(ctx, rank, world, uid)

    def close(self):
        if getattr(self, "_h", None):
            K.lib().chgpu_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def all_to_all_counts(self, send_counts):
        snd = (C.c_uint64 * self.world)(*[int(x) for x in send_counts])
        rcv = (C.c_uint64 * self.world)()
        K.check(K.lib().chgpu_all_to_all_counts(self._h, snd, rcv))
        return [int(x) for x in rcv]

    def all_to_all(self, col, send_counts, recv_counts):
        from .columns import Column
        snd = (C.c_uint64 * self.world)(*[int(x) for x in send_counts])
        rcv = (C.c_uint64 * self.world)(*[int(x) for x in recv_counts])
        h = C.c_void_p()
        K.check(K.lib().chgpu_all_to_all(self._h, col._h, snd, rcv, C.byref(h)))
        return Column(self.ctx, h)

    def all_to_all_multi(self, cols, send_counts):
        """every column of one partitioned Block in one exchange: -> ([received Columns], recv_counts)"""
        from .columns import Column
        n = len(cols)
        snd = (C.c_uint64 * self.world)(*[int(x) for x in send_counts])
        rcv = (C.c_uint64 * self.world)()
        hin = (C.c_void_p * max(1, n))(*[c._h for c in cols])
        hout = (C.c_void_p * max(1, n))()
        K.check(K.lib().chgpu_all_to_all_multi(self._h, n, hin, snd, rcv, hout))
        return [Column(self.ctx, C.c_void_p(hout[k])) for k in range(n)], [int(x) for x in rcv]

    def all_reduce_u64(self, values):
        n = len(values)
        buf = (C.c_uint64 * max(1, n))(*[int(v) % 2**64 for v in values])
        K.check(K.lib().chgpu_all_reduce_u64_host(self._h, buf, n))
        return [int(buf[i]) for i in range(n)]

    def all_reduce_column(self, col):
        K.check(K.lib().chgpu_all_reduce_u64(self._h, col._h))

    def barrier(self):
        K.check(K.lib().chgpu_comm_barrier(self._h))

    def stats(self):
        out = (C.c_uint64 * 3)()
        K.check(K.lib().chgpu_comm_stats(self._h, out))
        return dict(bytes_sent=int(out[0]), bytes_received=int(out[1]), collectives=int(out[2]))


class LocalEngine:
    """The per-GPU operators and the exchange the sharded algorithms need, on device columns through the C ABI."""

    def __init__(self, ctx, comm: Comm | None = None):
        import clickhouse_amd as ch
        self.ch, self.ctx, self.comm = ch, ctx, comm
        self.world = comm.world if comm is not None else 1
        self.rank = comm.rank if comm is not None else 0

    # -- the exchange ---------------------------------------------------------------------------
    def partition_by_hash(self, keys, cols, n_shards: int):
        return self.ch.partition_by_hash(keys, n_shards, cols)      # -> ([Column shards back to back], counts)

    def exchange_counts(self, counts):
        return self.comm.all_to_all_counts(counts)

    def all_to_all(self, col, counts, recv_counts):
        return self.comm.all_to_all(col, counts, recv_counts)

    def exchange(self, parts, counts):
        """the whole partitioned Block in ONE exchange (chgpu_all_to_all_multi): -> ([received columns], recv_counts)"""
        return self.comm.all_to_all_multi(parts, counts)

    def all_reduce_u64(self, values):
        return self.comm.all_reduce_u64(values) if self.comm is not None else [int(v) % 2**64 for v in values]

    # -- local operators ------------------------------------------------------------------------
    def Aggregator(self, key_dtype, aggs, size_hint=0):
        return self.ch.Aggregator(key_dtype, aggs, size_hint=size_hint, ctx=self.ctx)

    def HashJoin(self, kind, strictness, key_dtype=np.uint64):
        return self.ch.HashJoin(kind, strictness, key_dtype=key_dtype, ctx=self.ctx)

    def agg_add(self, agg, keys, args):
        agg.execute_on_block(keys, args)

    def agg_export(self, agg):
        return agg.export_state_columns()                            # (keys, [state words], rows)

    def agg_merge_states(self, agg, keys, words):
        agg.merge_states(keys, words, keys.size())

    def agg_result(self, agg):
        return agg.convert_to_block()

    def agg_finalize(self, agg):
        keys, cols = agg.finalize_columns()                          # (keys, [result columns], rows), resident in HBM
        return keys, cols, keys.size()

    def deserialize_states(self, kind, data, rows):
        return self.ch.deserialize_states(self.ctx, kind, data, [rows])

    def to_column(self, x, dtype):
        if isinstance(x, self.ch.Column):
            return x
        return self.ctx.column(np.ascontiguousarray(np.asarray(x) if dtype is None else np.asarray(x, dtype=dtype)))

    def cut(self, col, begin, rows):
        return col.cut(begin, rows)

    def rows(self, col) -> int:
        return col.size()

    def join_add(self, join, keys):
        join.add_block(keys)

    def join_finish(self, join):
        join.finish_build()

    def concat(self, cols):
        return cols[0] if len(cols) == 1 else self.ch.concat(cols)

    def join_count_sum(self, join, keys, payload):
        return join.probe_count_sum(keys, payload)                   # (count, sum) on this shard

    def join_materialize(self, join, keys, left_cols, right_cols):
        """joinBlock on this shard, all on the device: -> (n_out, [left columns replicated / filtered], [right columns gathered])"""
        r = join.probe_columns(keys)
        assert r["consumed"] == keys.size()
        flat = join.flatten_rowids(r["right_rowid"])
        if r["offsets"] is not None:
            left = [c.replicate(r["offsets"]) for c in left_cols]
        elif r["filter"] is not None:
            left = [c.filter(r["filter"]) for c in left_cols]
        else:
            left = list(left_cols)
        right = [c.index(flat, default_for_missing=True) for c in right_cols]
        return r["n_out"], left, right


class ShardedGroupBy:
    """GROUP BY across ranks: local pre-aggregation -> one all-to-all of partial states -> owner-side merge."""

    def __init__(self, engine, key_dtype, aggs, size_hint: int = 0):
        self.e, self.key_dtype, self.aggs = engine, np.dtype(key_dtype), aggs
        self.world = engine.world
        if not world_is_power_of_two(self.world):
            raise ValueError("the bucket rule `bucket & (world-1)` needs a power-of-two world size (ConcurrentHashJoin.cpp:158)")
        self.size_hint = size_hint
        self.local = engine.Aggregator(key_dtype, aggs, size_hint=size_hint)
        self.owner = None

    def add_block(self, keys, args):
        """executeOnBlock on this rank's rows"""
        self.e.agg_add(self.local, keys, args)

    def finish_columns(self):
        """-> the aggregator holding the groups this rank owns (results stay on the device)"""
        if self.world == 1:
            return self.local
        keys, words, rows = self.e.agg_export(self.local)                       # convertToBlockImplNotFinal
        parts, counts = self.e.partition_by_hash(keys, [keys] + words, self.world)
        got, recv_counts = self.e.exchange(parts, counts)                         # THE exchange step: every column in one group
        self.owner = self.e.Aggregator(self.key_dtype, self.aggs, size_hint=max(int(sum(recv_counts)), 1))
        self.e.agg_merge_states(self.owner, got[0], got[1:])                      # mergeBucketImpl on the owner
        return self.owner

    def finish(self):
        """-> (keys, [result columns]) of the groups this rank owns, on the host."""
        return self.e.agg_result(self.finish_columns())


class ShardedHashJoin:
    """parallel_hash across ranks: build rows and probe rows are routed to `bucket(key) & (world-1)`; each rank builds and probes its
    own shard.  Payload columns travel with their rows; joined rows stay on the rank that owns their key."""

    def __init__(self, engine, kind, strictness, key_dtype=np.uint64):
        self.e, self.kind, self.strictness, self.key_dtype = engine, kind, strictness, np.dtype(key_dtype)
        self.world, self.rank = engine.world, engine.rank
        if not world_is_power_of_two(self.world):
            raise ValueError("power-of-two world size required")
        self.join = engine.HashJoin(kind, strictness, key_dtype=key_dtype)
        self.payload_parts = None
        self.payload = None

    def _route(self, keys, cols):
        """dispatchBlock: -> (keys, [cols]) of the rows this rank owns"""
        if self.world == 1:
            return keys, list(cols)
        parts, counts = self.e.partition_by_hash(keys, [keys] + list(cols), self.world)
        got, _ = self.e.exchange(parts, counts)
        return got[0], got[1:]

    def add_build_rows(self, keys, payload_cols=()):
        k, pay = self._route(keys, payload_cols)
        self.e.join_add(self.join, k)
        if self.payload_parts is None:
            self.payload_parts = [[] for _ in pay]
        for lst, c in zip(self.payload_parts, pay):
            lst.append(c)

    def finish_build(self):
        self.e.join_finish(self.join)
        self.payload = [self.e.concat(lst) for lst in (self.payload_parts or [])]

    def probe_count_sum(self, keys, payload_index: int = 0):
        """SELECT count(), sum(right payload) over the WHOLE join: route the left keys, fused probe + aggregate on the owner, one
        16-byte all-reduce.  Integer payloads (wrap-around sums are order-independent)."""
        if self.payload is None:
            self.finish_build()
        k, _ = self._route(keys, [])
        cnt, sm = self.e.join_count_sum(self.join, k, self.payload[payload_index] if self.payload else None)
        c, s = self.e.all_reduce_u64([cnt, (sm or 0)])
        return c, s

    def probe(self, keys, left_cols=()):
        """joinBlock across ranks -> (n_out, [left columns], [right payload columns]) of the joined rows THIS rank owns, as the
        engine's columns (device columns under LocalEngine)"""
        if self.payload is None:
            self.finish_build()
        k, left = self._route(keys, left_cols)
        return self.e.join_materialize(self.join, k, [k] + left, self.payload)
