"""A CPU stand-in for clickhouse_amd.distributed.LocalEngine, built on the oracle: lets the multi-rank orchestration
(exchange of counts, all-to-all of hash partitions, owner-side merge) run under gloo without a GPU.  Test infrastructure."""
import numpy as np
import torch

import oracle as O


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


class CpuEngine:
    device = torch.device("cpu")

    @staticmethod
    def _np(t, dtype):
        return t.numpy().view(np.dtype(dtype)) if isinstance(t, torch.Tensor) else np.asarray(t, dtype=dtype)

    def partition_by_hash(self, keys, key_dtype, cols, dtypes, n_shards):
        k = self._np(keys, key_dtype)
        sel = O.hash_to_selector(np.ascontiguousarray(k), n_shards)
        order = np.argsort(sel, kind="stable")
        counts = np.bincount(sel.astype(np.int64), minlength=n_shards).astype(np.uint64)
        return [torch.from_numpy(np.ascontiguousarray(self._np(c, d)[order]).view(np.int64 if np.dtype(d).itemsize == 8 else np.int32 if np.dtype(d).itemsize == 4 else np.uint8))
                for c, d in zip(cols, dtypes)], counts

    def Aggregator(self, key_dtype, aggs, size_hint=0):
        return _CpuAgg(key_dtype, aggs)

    def HashJoin(self, kind, strictness, key_dtype=np.uint64):
        return O.HashJoin(kind, strictness)

    def agg_add(self, agg, keys, key_dtype, args, arg_dtypes):
        agg.rows.execute_on_block(self._np(keys, key_dtype), [self._np(a, d) if a is not None else None for a, d in zip(args, arg_dtypes)])

    def agg_export(self, agg, key_dtype):
        k, res = agg.result()
        return (torch.from_numpy(k.view(np.int64 if k.dtype.itemsize == 8 else np.int32)),
                [torch.from_numpy(np.ascontiguousarray(r).view(np.int64)) for r in res], k.shape[0])

    def agg_merge_states(self, agg, keys, key_dtype, words):
        agg.states.execute_on_block(self._np(keys, key_dtype), [self._np(w, np.uint64) for w in words])

    def agg_result(self, agg):
        return agg.result()

    def join_add(self, join, keys, key_dtype):
        join.add_block(self._np(keys, key_dtype).astype(np.uint64))

    def join_pairs(self, join, keys, key_dtype):
        return join.joined_pairs(self._np(keys, key_dtype).astype(np.uint64))
