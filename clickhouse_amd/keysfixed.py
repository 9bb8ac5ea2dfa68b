"""keys128 / keys256: GROUP BY and joins over several fixed-width key columns that pack into more than 8 bytes
(AggregatedDataVariants.h:70-71,83-84; HashMethodKeysFixed, src/Common/ColumnsHashing/HashMethod.h:236-410; packFixed,
src/Interpreters/AggregationCommon.h:91-158).  A device-resident exact dictionary (chgpu_keydict) turns the packed keys into dense UInt32
ids; the 8-byte-key operators run on the ids; the key columns of a result come back from the dictionary (insertKeyIntoColumns)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K
from .aggregator import Aggregator
from .columns import TAG_OF, Column, Context
from .hashjoin import HashJoin

NO_ID = 0xFFFFFFFF


class KeyDict:
    def __init__(self, key_dtypes, ctx: Context | None = None, size_hint: int = 0):
        self.ctx = ctx if ctx is not None else Context(0)
        self.key_dtypes = [np.dtype(d) for d in key_dtypes]
        total = sum(d.itemsize for d in self.key_dtypes)
        if total > 32:
            raise K.ChgpuError(K.ERR_NOT_IMPLEMENTED, f"{total} key bytes: beyond keys256 (the reference serializes such keys): CPU path")
        self.key_bytes = 16 if total <= 16 else 32  # keys128 when they fit, else keys256 (Aggregator.cpp:773-778)
        self.offsets = np.concatenate([[0], np.cumsum([d.itemsize for d in self.key_dtypes])[:-1]]).astype(int).tolist()
        h = C.c_void_p()
        K.check(K.lib().chgpu_keydict_create(self.ctx._live(), self.key_bytes, size_hint, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            K.lib().chgpu_keydict_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = C.c_uint64(0)
        K.check(K.lib().chgpu_keydict_size(self._h, C.byref(n)))
        return int(n.value)

    def encode(self, key_cols, insert: bool = True, row_begin: int = 0, row_end: int | None = None) -> Column:
        cols = [self.ctx.column(np.ascontiguousarray(c).astype(d, copy=False) if not isinstance(c, Column) else c) for c, d in zip(key_cols, self.key_dtypes)]
        assert len(cols) == len(self.key_dtypes)
        n = cols[0].size()
        ptrs = (C.c_void_p * len(cols))(*[c._h for c in cols])
        h = C.c_void_p()
        K.check(K.lib().chgpu_keydict_encode(self._h, len(cols), ptrs, row_begin, n if row_end is None else row_end, int(insert), C.byref(h)))
        return Column(self.ctx, h)

    def key_columns(self, ids: Column):
        """-> one Column per original key column for a column of ids"""
        out = []
        for d, off in zip(self.key_dtypes, self.offsets):
            h = C.c_void_p()
            K.check(K.lib().chgpu_keydict_key_column(self._h, ids._h, off, TAG_OF[d], C.byref(h)))
            out.append(Column(self.ctx, h))
        return out

    def selector(self, ids: Column, num_shards: int) -> Column:
        h = C.c_void_p()
        K.check(K.lib().chgpu_keydict_selector(self._h, ids._h, num_shards, C.byref(h)))
        return Column(self.ctx, h)


class KeysFixedAggregator:
    """Aggregator over keys128 / keys256: executeOnBlock packs and encodes the key columns, the inner aggregator groups by id."""

    def __init__(self, key_dtypes, aggs, size_hint: int = 0, ctx: Context | None = None):
        self.ctx = ctx if ctx is not None else Context(0)
        self.dict = KeyDict(key_dtypes, self.ctx, size_hint)
        self.inner = Aggregator(np.uint32, aggs, size_hint=size_hint, ctx=self.ctx)

    def execute_on_block(self, key_cols, args, row_begin: int = 0, row_end: int | None = None):
        ids = self.dict.encode(key_cols, True, row_begin, row_end)
        acols = [self.ctx.column(a) if a is not None else None for a in args]
        if row_begin or row_end is not None:
            n = ids.size()
            acols = [a.cut(row_begin, n) if a is not None else None for a in acols]
        self.inner.execute_on_block(ids, acols)

    def __len__(self):
        return len(self.inner)

    def finalize_columns(self):
        ids, res = self.inner.finalize_columns()
        return self.dict.key_columns(ids), res

    def convert_to_block(self):
        """-> ([key ndarrays], [result ndarrays]); row order unspecified, as in the reference"""
        keys, res = self.finalize_columns()
        return [k.numpy() for k in keys], [r.numpy() for r in res]


class KeysFixedHashJoin:
    """HashJoin over keys128 / keys256: the build side is encoded with emplace, the probe side with find (a key the build side does not
    hold gets an id no build row has, i.e. it misses)."""

    def __init__(self, key_dtypes, kind: int, strictness: int, ctx: Context | None = None):
        self.ctx = ctx if ctx is not None else Context(0)
        self.dict = KeyDict(key_dtypes, self.ctx)
        self.join = HashJoin(kind, strictness, key_dtype=np.uint32, ctx=self.ctx)

    def add_block(self, key_cols, null_map=None, join_mask=None):
        return self.join.add_block(self.dict.encode(key_cols, True), null_map, join_mask)

    def finish_build(self):
        self.join.finish_build()

    def probe_columns(self, key_cols, **kw):
        return self.join.probe_columns(self.dict.encode(key_cols, False), **kw)

    def joined_pairs(self, key_cols, **kw):
        return self.join.joined_pairs(self.dict.encode(key_cols, False), **kw)

    def probe_count_sum(self, key_cols, payload=None):
        return self.join.probe_count_sum(self.dict.encode(key_cols, False), payload)


class ColumnFixedString:
    """ColumnFixedString (src/Columns/ColumnFixedString.h): `chars` = rows x n raw bytes in HBM."""

    def __init__(self, chars: Column, n: int):
        assert chars.size() % n == 0
        self.chars, self.n, self.ctx = chars, n, chars.ctx

    @classmethod
    def from_numpy(cls, ctx: Context, values: np.ndarray):
        """values: an array of dtype 'S<n>' (numpy pads with zero bytes exactly like FixedString)"""
        values = np.ascontiguousarray(values)
        assert values.dtype.kind == "S"
        return cls(ctx.upload(values.view(np.uint8).reshape(-1)), values.dtype.itemsize)

    def size(self):
        return self.chars.size() // self.n

    @property
    def n_words(self):
        return (self.n + 7) // 8

    def words(self):
        """the value's 8-byte words as UInt64 Columns (chgpu_fixed_string_word)"""
        out = []
        for w in range(self.n_words):
            h = C.c_void_p()
            K.check(K.lib().chgpu_fixed_string_word(self.ctx._live(), self.chars._h, self.n, w, C.byref(h)))
            out.append(Column(self.ctx, h))
        return out

    @classmethod
    def from_words(cls, ctx: Context, words, n: int):
        ptrs = (C.c_void_p * len(words))(*[w._h for w in words])
        h = C.c_void_p()
        K.check(K.lib().chgpu_fixed_string_from_words(ctx._live(), len(words), ptrs, n, C.byref(h)))
        return cls(Column(ctx, h), n)

    def numpy(self):
        return self.chars.numpy().view(f"S{self.n}")


class FixedStringAggregator:
    """GROUP BY one FixedString(N) key (AggregatedDataVariants::key_fixed_string): N <= 8 -> the UInt64 aggregator over the value's single
    word, N <= 32 -> keys128 / keys256 over its words."""

    def __init__(self, n: int, aggs, size_hint: int = 0, ctx: Context | None = None):
        self.ctx = ctx if ctx is not None else Context(0)
        self.n = n
        if n > 32:
            raise K.ChgpuError(K.ERR_NOT_IMPLEMENTED, f"FixedString({n}) key: beyond keys256, CPU path")
        self.wide = n > 8
        self.inner = (KeysFixedAggregator([np.uint64] * ((n + 7) // 8), aggs, size_hint, self.ctx) if self.wide
                      else Aggregator(np.uint64, aggs, size_hint=size_hint, ctx=self.ctx))

    def execute_on_block(self, key: ColumnFixedString, args):
        assert key.n == self.n
        words = key.words()
        self.inner.execute_on_block(words if self.wide else words[0], args)

    def __len__(self):
        return len(self.inner)

    def convert_to_block(self):
        """-> (keys as an 'S<n>' ndarray, [result ndarrays])"""
        keys, res = self.inner.finalize_columns()
        words = keys if self.wide else [keys]
        return ColumnFixedString.from_words(self.ctx, words, self.n).numpy(), [r.numpy() for r in res]
