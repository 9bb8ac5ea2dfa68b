"""LowCardinality keys: ColumnLowCardinality = dictionary + index column
(src/Columns/ColumnLowCardinality.h:27-69), each Block with its own dictionary.

The reference's low_cardinality_key* aggregation methods (AggregatedDataVariants.h:119-127) resolve every dictionary entry
once per block and walk the rows through a per-position cache (HashMethodSingleLowCardinalityColumn,
ColumnsHashing.h:82-260).  Here `LowCardinalityDictionary` is that resolution on the host — the block's few thousand
dictionary entries against the query-wide dictionary — and `chgpu_lc_remap` walks the rows on the device, producing an
ordinary UInt32 key column for GROUP BY / join / sharding.  Strings never reach the device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K
from .aggregator import Aggregator
from .columns import Column, Context


class ColumnLowCardinality:
    """dictionary: sequence of values (str / bytes / numbers), position -> value; indexes: UInt8/16/32/64 Column in HBM"""

    def __init__(self, dictionary, indexes: Column):
        # the list object itself identifies the dictionary: Blocks cut / filtered from one column share it, and LowCardinalityDictionary
        # resolves a shared dictionary once (the reference keys its cache by the dictionary's hash, ColumnsHashing.h:155-183)
        self.dictionary = dictionary if isinstance(dictionary, list) else list(dictionary)
        self.indexes = indexes

    def size(self) -> int:
        return self.indexes.size()

    @staticmethod
    def from_values(ctx: Context, values, index_dtype=None) -> "ColumnLowCardinality":
        """Build from a full host column (IColumn -> ColumnLowCardinality::insertRangeFromFullColumn): dictionary in order of
        first appearance, the narrowest index type that holds it (ColumnLowCardinality::Index::check / expandType)."""
        values = np.asarray(values)
        uniq, first, inv = np.unique(values, return_index=True, return_inverse=True)
        order = np.argsort(first, kind="stable")
        rank = np.empty_like(order)
        rank[order] = np.arange(order.shape[0])
        d = uniq[order].tolist()
        if index_dtype is None:
            index_dtype = np.uint8 if len(d) <= 2**8 else np.uint16 if len(d) <= 2**16 else np.uint32
        return ColumnLowCardinality(d, ctx.upload(rank[inv].astype(index_dtype)))

    def filter(self, filt: Column) -> "ColumnLowCardinality":
        """ColumnLowCardinality::filter: the indexes are filtered, the dictionary is shared"""
        return ColumnLowCardinality(self.dictionary, self.indexes.filter(filt))

    def cut(self, start: int, length: int) -> "ColumnLowCardinality":
        return ColumnLowCardinality(self.dictionary, self.indexes.cut(start, length))

    def convert_to_full_column(self) -> list:
        """convertToFullColumn (:53) on the host, for tests"""
        idx = self.indexes.numpy()
        return [self.dictionary[int(i)] for i in idx]


class ColumnString:
    """ColumnString (src/Columns/ColumnString.h:40-49) in HBM: chars (every value followed by a zero byte) + cumulative offsets."""

    def __init__(self, offsets: Column, chars: Column, host_values=None):
        self.offsets = offsets
        self.chars = chars
        self._host_values = host_values  # the caller's Block, when it is at hand (saves the download in dictionary())

    @staticmethod
    def from_values(ctx: Context, values) -> "ColumnString":
        vals = [v.encode() if isinstance(v, str) else bytes(v) for v in values]
        lens = np.fromiter((len(v) + 1 for v in vals), dtype=np.uint64, count=len(vals))
        chars = np.frombuffer(b"".join(v + b"\0" for v in vals), dtype=np.uint8) if vals else np.zeros(0, dtype=np.uint8)
        return ColumnString(ctx.upload(np.cumsum(lens, dtype=np.uint64)), ctx.upload(chars), vals)

    def size(self) -> int:
        return self.offsets.size()

    def value_at(self, rows) -> list:
        if self._host_values is not None:
            return [self._host_values[int(r)] for r in rows]
        # ColumnString::index over a few rows, on the device: only the requested values cross PCIe
        rows = np.asarray(rows, dtype=np.uint64)
        if rows.shape[0] == 0:
            return []
        ctx = self.offsets.ctx
        ends = self.offsets.index(ctx.upload(rows)).numpy()
        prev = np.where(rows > 0, rows - np.uint64(1), np.uint64(0))
        begins = np.where(rows > 0, self.offsets.index(ctx.upload(prev)).numpy(), np.uint64(0))
        lens = (ends - begins - np.uint64(1)).astype(np.int64)
        flat = np.repeat(begins.astype(np.int64) - np.concatenate(([0], np.cumsum(lens)[:-1])), lens) + np.arange(int(lens.sum()), dtype=np.int64)
        data = self.chars.index(ctx.upload(flat.astype(np.uint64))).numpy().tobytes() if flat.shape[0] else b""
        cuts = np.concatenate(([0], np.cumsum(lens)))
        return [data[int(cuts[k]):int(cuts[k + 1])] for k in range(rows.shape[0])]

    def filter(self, filt: Column) -> "ColumnString":
        """ColumnString::filter (ColumnString.cpp:270-290): the kept values, in order"""
        oo, oc = C.c_void_p(), C.c_void_p()
        rows = C.c_uint64(0)
        ctx = self.offsets.ctx
        K.check(K.lib().chgpu_string_filter(ctx._h, self.offsets._h, self.chars._h, filt._h, C.byref(oo), C.byref(oc), C.byref(rows)))
        return ColumnString(Column(ctx, oo), Column(ctx, oc))

    def to_list(self) -> list:
        """the values on the host (tests)"""
        offs = self.offsets.numpy()
        chars = self.chars.numpy().tobytes()
        return [chars[(int(offs[i - 1]) if i else 0):int(offs[i]) - 1] for i in range(offs.shape[0])]

    def dictionary_encode(self) -> "ColumnLowCardinality":
        """String -> LowCardinality(String) on the device: ids by first appearance; the dictionary's strings are read on the host"""
        ids, rows = C.c_void_p(), C.c_void_p()
        n = C.c_uint64(0)
        ctx = self.offsets.ctx
        K.check(K.lib().chgpu_string_dictionary_encode(ctx._h, self.offsets._h, self.chars._h, C.byref(ids), C.byref(rows), C.byref(n)))
        first_rows = Column(ctx, rows).numpy()
        return ColumnLowCardinality(self.value_at(first_rows), Column(ctx, ids))


class LowCardinalityDictionary:
    """The query-wide dictionary: value -> global id (insertion order; ids are dense, so they are also ideal GROUP BY keys
    for the LDS-staged RANGE strategy)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.values = []
        self._ids = {}
        self._cache = {}  # id(block dictionary list) -> (len, remap Column): Blocks of one part share their dictionary

    def __len__(self):
        return len(self.values)

    def remap_table(self, dictionary) -> Column:
        key = id(dictionary)
        hit = self._cache.get(key)
        if hit is not None and hit[0] is dictionary:
            return hit[1]
        out = np.empty(len(dictionary), dtype=np.uint32)
        for pos, v in enumerate(dictionary):
            g = self._ids.get(v)
            if g is None:
                g = len(self.values)
                self._ids[v] = g
                self.values.append(v)
            out[pos] = g
        col = self.ctx.upload(out)
        self._cache[key] = (dictionary, col)  # keeps the list alive: id() stays unique
        return col

    def map_block(self, col: ColumnLowCardinality) -> Column:
        """-> UInt32 Column of global ids, one per row"""
        remap = self.remap_table(col.dictionary)
        h = C.c_void_p()
        K.check(K.lib().chgpu_lc_remap(self.ctx._h, col.indexes._h, remap._h, C.byref(h)))
        return Column(self.ctx, h)

    def decode(self, ids: np.ndarray) -> list:
        return [self.values[int(i)] for i in ids]


class LowCardinalityAggregator:
    """Aggregator over one LowCardinality key column (the low_cardinality_key_string variant): same interface as Aggregator,
    keys come back as dictionary values."""

    def __init__(self, aggs, ctx: Context | None = None, size_hint: int = 0):
        self.ctx = ctx if ctx is not None else Context(0)
        self.dictionary = LowCardinalityDictionary(self.ctx)
        self.agg = Aggregator(np.uint32, aggs, size_hint=size_hint, ctx=self.ctx)

    def execute_on_block(self, keys: ColumnLowCardinality, args, row_begin: int = 0, row_end: int | None = None, filter=None):
        self.agg.execute_on_block(self.dictionary.map_block(keys), args, row_begin, row_end, filter=filter)

    def __len__(self):
        return len(self.agg)

    def convert_to_block(self):
        ids, res = self.agg.convert_to_block()
        return self.dictionary.decode(ids), res


class PackedKeysAggregator:
    """GROUP BY several key columns, any of them LowCardinality / dictionary-encoded String — the keys16/32/64 variants
    (chooseAggregationMethod, Aggregator.cpp:773-778: all keys fixed-width and <= 8 bytes together -> packFixed<UInt64>) with
    LowCardinality keys contributing their dictionary positions, which is how the reference's `low_cardinality_keys128/256` treat them
    (AggregatedDataVariants.h:128-131, HashMethodKeysFixed with has_low_cardinality).  A LowCardinality key takes 2 bytes of the packed
    key (query-wide ids cast to UInt16 by a run-time compiled kernel); more than 65 536 distinct values or more than 8 key bytes ->
    CHGPU_ERR_NOT_IMPLEMENTED (the caller keeps its CPU method).  SSB Q3.1's `GROUP BY c_nation, s_nation, d_year` is 2 + 2 + 2 bytes."""

    def __init__(self, key_kinds, aggs, ctx: Context | None = None, size_hint: int = 0):
        """key_kinds: one entry per key column, either "lc" or a numpy dtype"""
        from .expression import ActionsDAG
        self.ctx = ctx if ctx is not None else Context(0)
        self.kinds = [k if k == "lc" else np.dtype(k) for k in key_kinds]
        self.dicts = [LowCardinalityDictionary(self.ctx) if k == "lc" else None for k in self.kinds]
        self.widths = [2 if k == "lc" else k.itemsize for k in self.kinds]
        if sum(self.widths) > 8:
            raise K.ChgpuError(K.ERR_NOT_IMPLEMENTED, "more than 8 key bytes: keys128/256 stay on the CPU")
        d = ActionsDAG()
        d.add_function("toUInt16", d.add_input(0, np.uint32))
        self._narrow = d.compile()
        self.agg = Aggregator(np.uint64, aggs, size_hint=size_hint, ctx=self.ctx)

    def execute_on_block(self, keys, args, filter=None):
        from .columns import pack_fixed_keys
        cols = []
        for k, kind, dic in zip(keys, self.kinds, self.dicts):
            if kind == "lc":
                ids = dic.map_block(k)
                if len(dic) > 65536:
                    raise K.ChgpuError(K.ERR_NOT_IMPLEMENTED, "a LowCardinality key with more than 65536 values does not fit its 2 key bytes")
                cols.append(self._narrow.execute(self.ctx, [ids], [1])[0])
            else:
                cols.append(self.ctx.column(k))
        self.agg.execute_on_block(pack_fixed_keys(cols), args, filter=filter)

    def __len__(self):
        return len(self.agg)

    def convert_to_block(self):
        """-> ([one list / ndarray per key column], [result ndarrays])"""
        from .columns import unpack_fixed_key
        keys, res = self.agg.finalize_columns()
        out, off = [], 0
        for kind, dic, w in zip(self.kinds, self.dicts, self.widths):
            part = unpack_fixed_key(keys, off, np.uint16 if kind == "lc" else kind).numpy()
            out.append(dic.decode(part) if kind == "lc" else part)
            off += w
        return out, [r.numpy() for r in res]
