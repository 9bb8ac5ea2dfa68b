"""Aggregator mirror (src/Interpreters/Aggregator.h:179-265) over the C ABI: one instance == one
AggregatedDataVariants living in HBM.  execute_on_block / merge / convert_to_block follow executeOnBlock /
mergeDataImpl / convertToBlockImplFinal."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K
from .columns import NP_OF, TAG_OF, Column, Context, sum_result_dtype


class Aggregator:
    def __init__(self, key_dtype, aggs, two_level_threshold: int = 100000, size_hint: int = 0, ctx: Context | None = None):
        """aggs: list of (kind, arg_dtype or None).  key_dtype None = without_key.  two_level_threshold is accepted for
        interface parity (Aggregator::Params) — the device table is single-level."""
        self.ctx = ctx if ctx is not None else Context(0)
        self.key_tag = -1 if key_dtype is None else TAG_OF[np.dtype(key_dtype)]
        self.aggs = [(k, (TAG_OF[np.dtype(d)] if d is not None else K.U64)) for k, d in aggs]
        kinds = (C.c_int * max(1, len(self.aggs)))(*[k for k, _ in self.aggs])
        types = (C.c_int * max(1, len(self.aggs)))(*[t for _, t in self.aggs])
        h = C.c_void_p()
        K.check(K.lib().chgpu_agg_create(self.ctx._h, self.key_tag, len(self.aggs), kinds, types, size_hint, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            K.lib().chgpu_agg_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def execute_on_block(self, keys, args, row_begin: int = 0, row_end: int | None = None, filter=None):
        """Aggregator::executeOnBlock(columns, row_begin, row_end, result, key_columns, aggregate_columns, ...).
        filter: a UInt8 WHERE mask over the same rows (a FilterTransform fused in front of the aggregation)."""
        kcol = self.ctx.column(keys) if keys is not None else None
        acols = [self.ctx.column(a) if a is not None else None for a in args]
        fcol = self.ctx.column(filter) if filter is not None else None
        n = kcol.size() if kcol is not None else (fcol.size() if fcol is not None else next(a.size() for a in acols if a is not None))
        row_end = n if row_end is None else row_end
        ptrs = (C.c_void_p * max(1, len(acols)))(*[(a._h if a is not None else None) for a in acols])
        if fcol is None:
            K.check(K.lib().chgpu_agg_add_block(self._h, kcol._h if kcol is not None else None, ptrs, row_begin, row_end))
        else:
            K.check(K.lib().chgpu_agg_add_block_filtered(self._h, kcol._h if kcol is not None else None, ptrs, row_begin, row_end, fcol._h))

    def merge(self, other: "Aggregator"):
        K.check(K.lib().chgpu_agg_merge(self._h, other._h))

    def merge_states(self, keys: Column | None, state_cols, rows: int):
        ptrs = (C.c_void_p * max(1, len(state_cols)))(*[c._h for c in state_cols])
        K.check(K.lib().chgpu_agg_merge_states(self._h, keys._h if keys is not None else None, ptrs, rows))

    def __len__(self):
        n = C.c_uint64(0)
        K.check(K.lib().chgpu_agg_size(self._h, C.byref(n)))
        return int(n.value)

    @property
    def n_words(self):
        return sum(2 if k in (K.AGG_AVG, K.AGG_ANY) else 1 for k, _ in self.aggs)   # (min / max: one order-key word; any: claim + value)

    def result_dtypes(self):
        out = []
        for kind, t in self.aggs:
            out.append(np.uint64 if kind == K.AGG_COUNT else np.float64 if kind == K.AGG_AVG else NP_OF[t] if kind in (K.AGG_MIN, K.AGG_MAX, K.AGG_ANY) else sum_result_dtype(t))
        return out

    def finalize_columns(self):
        """-> (keys Column or None, [result Columns]) resident in HBM."""
        kh = C.c_void_p()
        res = (C.c_void_p * max(1, len(self.aggs)))()
        n = C.c_uint64(0)
        K.check(K.lib().chgpu_agg_finalize(self._h, C.byref(kh), res, C.byref(n)))
        keys = Column(self.ctx, kh) if kh.value else None
        return keys, [Column(self.ctx, C.c_void_p(res[j])) for j in range(len(self.aggs))]

    def export_state_columns(self):
        kh = C.c_void_p()
        nw = self.n_words
        res = (C.c_void_p * max(1, nw))()
        n = C.c_uint64(0)
        K.check(K.lib().chgpu_agg_export_states(self._h, C.byref(kh), res, C.byref(n)))
        keys = Column(self.ctx, kh) if kh.value else None
        return keys, [Column(self.ctx, C.c_void_p(res[w])) for w in range(nw)], int(n.value)

    def export_state_columns_two_level(self):
        """the partial states ordered by two-level bucket number -> (keys, [state Columns], groups, bucket_counts[256])"""
        kh = C.c_void_p()
        nw = self.n_words
        res = (C.c_void_p * max(1, nw))()
        n = C.c_uint64(0)
        counts = (C.c_uint64 * 256)()
        K.check(K.lib().chgpu_agg_export_states_two_level(self._h, C.byref(kh), res, C.byref(n), counts))
        return Column(self.ctx, kh), [Column(self.ctx, C.c_void_p(res[w])) for w in range(nw)], int(n.value), [int(x) for x in counts]

    def convert_to_block(self):
        """Aggregator::convertToBlocks(final=true) downloaded: (keys ndarray or None, [result ndarrays])."""
        keys, res = self.finalize_columns()
        got = Column.numpy_many(([keys] if keys is not None else []) + res)   # one wait for the whole result Block
        return (got[0] if keys is not None else None), got[(1 if keys is not None else 0):]


def serialize_states(ctx: Context, kind: int, word0: Column, word1: Column | None = None):
    """the ColumnAggregateFunction wire bytes of one aggregate function's states (IAggregateFunction::serialize per row: sum = 8 bytes,
    count = VarUInt, avg = 8 bytes + VarUInt) -> (bytes Column, row offsets Column[rows + 1])"""
    bh, oh = C.c_void_p(), C.c_void_p()
    K.check(K.lib().chgpu_agg_serialize_states(ctx._live(), kind, word0._h, word1._h if word1 is not None else None, C.byref(bh), C.byref(oh)))
    return Column(ctx, bh), Column(ctx, oh)


def deserialize_states(ctx: Context, kind: int, data: Column, stream_rows, stream_byte_begin=None):
    """the inverse, for mergeOnBlock: `stream_rows[s]` states from byte `stream_byte_begin[s]` of every stream -> (word0, word1 or None)"""
    n = len(stream_rows)
    rows = (C.c_uint64 * n)(*[int(r) for r in stream_rows])
    begin = (C.c_uint64 * n)(*[int(b) for b in stream_byte_begin]) if stream_byte_begin is not None else None
    h0, h1 = C.c_void_p(), C.c_void_p()
    K.check(K.lib().chgpu_agg_deserialize_states(ctx._live(), kind, data._h, n, begin, rows, C.byref(h0), C.byref(h1) if kind == K.AGG_AVG else None))
    return Column(ctx, h0), (Column(ctx, h1) if kind == K.AGG_AVG else None)


class NullableKeyAggregator:
    """GROUP BY a Nullable(T) key: AggregationDataWithNullKey (src/Interpreters/AggregatedData.h:71-95) keeps the NULL group's state
    out of the hash table (has_null_key_data / null_key_data), and the key extraction sends rows whose null-map byte is set there
    (ColumnsHashingImpl.h:196-240) whatever their nested value is.  Here: the hash table aggregates the rows whose null-map byte is 0
    (the WHERE-fused form of add_block, mask = NOT null), an aggregation without key takes the rows whose byte is set."""

    def __init__(self, key_dtype, aggs, ctx: Context | None = None, size_hint: int = 0):
        self.ctx = ctx if ctx is not None else Context(0)
        self.keyed = Aggregator(key_dtype, aggs, size_hint=size_hint, ctx=self.ctx)
        self.null_group = Aggregator(None, aggs, ctx=self.ctx)
        self.has_null_key_data = False
        from .expression import ActionsDAG
        d = ActionsDAG()
        d.add_function("not", d.add_input(0, np.uint8))
        self._not = d.compile()

    def execute_on_block(self, keys, null_map, args):
        """keys: the nested column of the ColumnNullable, null_map: its UInt8 null map (ColumnNullable.h)"""
        from .columns import count_bytes_in_filter
        k = self.ctx.column(keys)
        nm = self.ctx.column(null_map)
        acols = [self.ctx.column(a) if a is not None else None for a in args]
        not_null = self._not.execute(self.ctx, [nm], [1])[0]
        self.keyed.execute_on_block(k, acols, filter=not_null)
        if count_bytes_in_filter(nm):
            self.has_null_key_data = True
            self.null_group.execute_on_block(None, acols, filter=nm)

    def __len__(self):
        return len(self.keyed) + (1 if self.has_null_key_data else 0)

    def convert_to_block(self):
        """-> (keys ndarray, key null map ndarray[uint8], [result ndarrays]); the NULL group, when present, is the last row (the
        reference appends it the same way: insertDefault into the key column + 1 in the null map)"""
        keys, res = self.keyed.convert_to_block()
        nulls = np.zeros(keys.shape[0], dtype=np.uint8)
        if self.has_null_key_data:
            _, nres = self.null_group.convert_to_block()
            keys = np.concatenate([keys, np.zeros(1, dtype=keys.dtype)])
            nulls = np.concatenate([nulls, np.ones(1, dtype=np.uint8)])
            res = [np.concatenate([r, n.astype(r.dtype)]) for r, n in zip(res, nres)]
        return keys, nulls, res


def group_by_min_max(ctx: Context, keys: Column, values: Column):
    """SELECT key, min(value), max(value) GROUP BY key ORDER BY key: min / max states in the hash aggregator (CHGPU_AGG_MIN / _MAX, round 3 --
    rounds 1-2 took a detour over two stable sorts), the groups then ordered by key.
    -> (keys Column, min Column, max Column), one row per group, ascending by key."""
    from .columns import sort_permutation
    if keys.size() == 0:
        return ctx.alloc(keys.dtype, 0), ctx.alloc(values.dtype, 0), ctx.alloc(values.dtype, 0)
    agg = Aggregator(keys.dtype, [(K.AGG_MIN, values.dtype), (K.AGG_MAX, values.dtype)], ctx=ctx)
    agg.execute_on_block(keys, [values, values])
    gk, (mn, mx) = agg.finalize_columns()
    perm = sort_permutation(gk)
    return gk.index(perm), mn.index(perm), mx.index(perm)
