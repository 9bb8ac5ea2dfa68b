"""HashJoin mirror (IJoin: src/Interpreters/IJoin.h:80-142; HashJoin key64) over the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K
from .columns import TAG_OF, Column, Context

NO_ROW = 0xFFFFFFFFFFFFFFFF


class HashJoin:
    def __init__(self, kind: int, strictness: int, any_take_last_row: bool = False, key_dtype=np.uint64, ctx: Context | None = None):
        self.ctx = ctx if ctx is not None else Context(0)
        self.kind, self.strictness = kind, strictness
        self.key_dtype = np.dtype(key_dtype)
        h = C.c_void_p()
        K.check(K.lib().chgpu_join_create(self.ctx._h, TAG_OF[self.key_dtype], kind, strictness, int(any_take_last_row), 0, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            K.lib().chgpu_join_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def need_replication(self):  # JoinFeatures.h:28: is_all_join || (is_any_join && right) || (is_semi_join && right)
        return self.strictness == K.STRICT_ALL or (self.kind == K.JOIN_RIGHT and self.strictness in (K.STRICT_ANY, K.STRICT_SEMI))

    @property
    def need_filter(self):  # JoinFeatures.h:32
        return not self.need_replication and (self.kind in (K.JOIN_INNER, K.JOIN_RIGHT) or self.strictness in (K.STRICT_SEMI, K.STRICT_ANTI))

    def _col(self, x, dtype=None):
        if isinstance(x, Column):
            return x
        x = np.ascontiguousarray(x)
        if dtype is not None and x.dtype != dtype:
            x = x.astype(dtype)
        return self.ctx.upload(x)

    def add_block(self, keys, null_map=None, join_mask=None) -> int:
        """IJoin::addBlockToJoin"""
        k = self._col(keys, self.key_dtype)
        nm = self._col(null_map, np.uint8) if null_map is not None else None
        jm = self._col(join_mask, np.uint8) if join_mask is not None else None
        idx = C.c_uint32(0)
        K.check(K.lib().chgpu_join_add_block(self._h, k._h, nm._h if nm else None, jm._h if jm else None, C.byref(idx)))
        return int(idx.value)

    def finish_build(self):
        """IJoin::onBuildPhaseFinish"""
        K.check(K.lib().chgpu_join_finish_build(self._h))

    @property
    def total_rows(self):
        r = C.c_uint64(0)
        K.check(K.lib().chgpu_join_total_rows(self._h, C.byref(r), None))
        return int(r.value)

    @property
    def n_keys(self):
        r, k = C.c_uint64(0), C.c_uint64(0)
        K.check(K.lib().chgpu_join_total_rows(self._h, C.byref(r), C.byref(k)))
        return int(k.value)

    def probe_columns(self, keys, null_map=None, max_joined_block_rows: int = 0, need_right_rows: bool = True):
        """joinBlock's joinRightColumns -> dict(consumed, n_out, filter, offsets, right_rowid) of device Columns.
        need_right_rows=False (LEFT SEMI / LEFT ANTI only): the right side contributes no columns, only the filter is built."""
        k = self._col(keys, self.key_dtype)
        nm = self._col(null_map, np.uint8) if null_map is not None else None
        fh, oh, rh = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n_out, consumed = C.c_uint64(0), C.c_uint64(0)
        K.check(K.lib().chgpu_join_probe(self._h, k._h, nm._h if nm else None, max_joined_block_rows, C.byref(fh), C.byref(oh),
                                         C.byref(rh) if need_right_rows else None, C.byref(n_out), C.byref(consumed)))
        return dict(consumed=int(consumed.value), n_out=int(n_out.value),
                    filter=Column(self.ctx, fh) if fh.value else None,
                    offsets=Column(self.ctx, oh) if oh.value else None,
                    right_rowid=Column(self.ctx, rh) if need_right_rows else None)

    def probe_count_sum(self, keys, payload=None, null_map=None):
        """joinBlock + `SELECT count(), sum(payload)` behind it, fused (chgpu_join_probe_agg): -> (count, sum).  payload: the right
        side's column over all right blocks in insertion order (None: count only, sum is None).  The sum has SumSimple's type:
        a Python int for integer payloads (signed for signed types), a float for Float32/64."""
        k = self._col(keys, self.key_dtype)
        nm = self._col(null_map, np.uint8) if null_map is not None else None
        p = self._col(payload) if payload is not None else None
        cnt = C.c_uint64(0)
        bits = C.c_uint64(0)
        K.check(K.lib().chgpu_join_probe_agg(self._h, k._h, nm._h if nm else None, p._h if p else None, C.byref(cnt),
                                             C.byref(bits) if p else None))
        if p is None:
            return int(cnt.value), None
        kind = np.dtype(p.dtype).kind
        raw = np.array([bits.value], dtype=np.uint64)
        if kind == "f":
            return int(cnt.value), float(raw.view(np.float64)[0])
        return int(cnt.value), int(raw.view(np.int64)[0]) if kind == "i" else int(bits.value)

    def non_joined_rows(self):
        """IJoin::getNonJoinedBlocks for RIGHT / FULL joins: (right_block, right_row) of the build rows no left row matched,
        in insertion order (call after the last probe)"""
        h = C.c_void_p()
        n = C.c_uint64(0)
        K.check(K.lib().chgpu_join_non_joined_rows(self._h, C.byref(h), C.byref(n)))
        rid = Column(self.ctx, h).numpy()
        return (rid >> np.uint64(32)).astype(np.int64), (rid & np.uint64(0xFFFFFFFF)).astype(np.int64)

    def flatten_rowids(self, right_rowid: Column) -> Column:
        """(block << 32 | row) -> ordinal over all right blocks (index into concatenated payload columns)"""
        h = C.c_void_p()
        K.check(K.lib().chgpu_join_flatten_rowids(self._h, right_rowid._h, C.byref(h)))
        return Column(self.ctx, h)

    def probe(self, keys, null_map=None, max_joined_block_rows: int = 0):
        r = self.probe_columns(keys, null_map, max_joined_block_rows)
        rid = r["right_rowid"].numpy()
        miss = rid == np.uint64(NO_ROW)
        block = np.where(miss, -1, (rid >> np.uint64(32)).astype(np.int64))
        row = np.where(miss, -1, (rid & np.uint64(0xFFFFFFFF)).astype(np.int64))
        return dict(consumed=r["consumed"], filter=r["filter"].numpy() if r["filter"] is not None else None,
                    offsets=r["offsets"].numpy() if r["offsets"] is not None else None,
                    added_block=block.astype(np.int64), added_row=row.astype(np.int64))

    def joined_pairs(self, keys, null_map=None, max_joined_block_rows: int = 0):
        """Canonical observable result: (left_row, right_block, right_row) per joined row, in output order."""
        r = self.probe(keys, null_map, max_joined_block_rows)
        c = r["consumed"]
        if self.need_replication:
            counts = np.diff(np.concatenate([[0], r["offsets"]])).astype(np.int64)
            left = np.repeat(np.arange(c, dtype=np.int64), counts)
        elif self.need_filter:
            left = np.nonzero(r["filter"])[0].astype(np.int64)
        else:
            left = np.arange(c, dtype=np.int64)
        assert left.shape[0] == r["added_block"].shape[0]
        return left, r["added_block"], r["added_row"], c


def join_probe_chain(joins, keys, null_maps=None, right_rows=None, carry=None, want_indexes=True, want_filter=False, right_cols=None):
    """A chain of filter-form JoiningTransforms (LEFT SEMI / LEFT ANTI / ALL over unique build keys) answered in one sweep over the left
    key columns, before any left column is copied (chgpu_join_probe_chain): keys[s] is probed against joins[s].
    right_rows[s] truthy: also return joins[s]'s matched build row per survivor; carry: left Columns to gather at the survivors.
    right_cols[s] (a Column of joins[s]'s one build block): return that column's values at the matched rows instead of the row ids
    (chgpu_join_probe_chain_columns).
    -> dict(kept, indexes, right_rowid=[Column | None per step: row ids, or the right column's values], carry=[Column ...], filter)"""
    assert len(joins) == len(keys) and len(joins) >= 1
    ctx = joins[0].ctx
    n = len(joins)
    cols = [j._col(k, j.key_dtype) for j, k in zip(joins, keys)]
    nms = [None] * n if null_maps is None else [j._col(m, np.uint8) if m is not None else None for j, m in zip(joins, null_maps)]
    carry = list(carry or [])
    jh = (C.c_void_p * n)(*[j._h for j in joins])
    kh = (C.c_void_p * n)(*[c._h for c in cols])
    nh = (C.c_void_p * n)(*[(m._h if m is not None else None) for m in nms])
    want = (C.c_int * n)(*[int(bool(right_rows[s])) if right_rows is not None else 0 for s in range(n)])
    rh = (C.c_void_p * n)()
    nc = len(carry)
    ch_in = (C.c_void_p * max(nc, 1))(*[c._h for c in carry])
    ch_out = (C.c_void_p * max(nc, 1))()
    ih, fh = C.c_void_p(), C.c_void_p()
    kept = C.c_uint64(0)
    if right_cols is not None and any(c is not None for c in right_cols):
        ph = (C.c_void_p * n)(*[(c._h if c is not None else None) for c in right_cols])
        K.check(K.lib().chgpu_join_probe_chain_columns(n, jh, kh, nh if null_maps is not None else None, want, ph, nc, ch_in if nc else None,
                                                       C.byref(ih) if want_indexes else None, rh, ch_out if nc else None,
                                                       C.byref(fh) if want_filter else None, C.byref(kept)))
    else:
        K.check(K.lib().chgpu_join_probe_chain(n, jh, kh, nh if null_maps is not None else None, want, nc, ch_in if nc else None,
                                               C.byref(ih) if want_indexes else None, rh, ch_out if nc else None,
                                               C.byref(fh) if want_filter else None, C.byref(kept)))
    return dict(kept=int(kept.value), indexes=Column(ctx, ih) if want_indexes else None,
                right_rowid=[Column(ctx, C.c_void_p(rh[s])) if rh[s] else None for s in range(n)],
                carry=[Column(ctx, C.c_void_p(ch_out[c])) for c in range(nc)],
                filter=Column(ctx, fh) if want_filter else None)


class AsofJoin:
    """ASOF INNER / LEFT join (JoinStrictness::Asof; RowRefs.cpp SortedLookupVector, HashJoinMethodsImpl.h:462-478) over the C ABI:
    one fixed-width integer key + one numeric asof column; `inequality` as ASOFJoinInequality with the LEFT value on the left."""

    def __init__(self, kind: int, inequality: int = K.ASOF_GREATER_OR_EQUALS, key_dtype=np.uint64, asof_dtype=np.uint32, ctx: Context | None = None):
        self.ctx = ctx if ctx is not None else Context(0)
        self.kind, self.inequality = kind, inequality
        self.key_dtype, self.asof_dtype = np.dtype(key_dtype), np.dtype(asof_dtype)
        h = C.c_void_p()
        K.check(K.lib().chgpu_asof_create(self.ctx._h, TAG_OF[self.key_dtype], TAG_OF[self.asof_dtype], kind, inequality, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            K.lib().chgpu_asof_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _col(self, x, dtype):
        return x if isinstance(x, Column) else self.ctx.upload(np.ascontiguousarray(x, dtype=dtype))

    def add_block(self, keys, asof, null_map=None, join_mask=None) -> int:
        bi = C.c_uint64(0)
        k, a = self._col(keys, self.key_dtype), self._col(asof, self.asof_dtype)
        nm = self._col(null_map, np.uint8) if null_map is not None else None
        jm = self._col(join_mask, np.uint8) if join_mask is not None else None
        K.check(K.lib().chgpu_asof_add_block(self._h, k._h, a._h, nm._h if nm is not None else None, jm._h if jm is not None else None, C.byref(bi)))
        return int(bi.value)

    @property
    def total_rows(self) -> int:
        n = C.c_uint64(0)
        K.check(K.lib().chgpu_asof_total_rows(self._h, C.byref(n)))
        return int(n.value)

    def joined_pairs(self, keys, asof, null_map=None):
        """-> (left_row, right_block, right_row) per output row; LEFT: one per left row, (-1, -1) for the default row"""
        k, a = self._col(keys, self.key_dtype), self._col(asof, self.asof_dtype)
        nm = self._col(null_map, np.uint8) if null_map is not None else None
        fh, rh = C.c_void_p(), C.c_void_p()
        n = C.c_uint64(0)
        K.check(K.lib().chgpu_asof_probe(self._h, k._h, a._h, nm._h if nm is not None else None, C.byref(fh), C.byref(rh), C.byref(n)))
        filt = Column(self.ctx, fh).numpy()
        rid = Column(self.ctx, rh).numpy()
        left = np.nonzero(filt)[0].astype(np.int64) if self.kind == K.JOIN_INNER else np.arange(filt.shape[0], dtype=np.int64)
        assert rid.shape[0] == left.shape[0] == n.value
        miss = rid == NO_ROW
        block = np.where(miss, -1, (rid >> np.uint64(32)).astype(np.int64))
        row = np.where(miss, -1, (rid & np.uint64(0xFFFFFFFF)).astype(np.int64))
        return left, block, row
