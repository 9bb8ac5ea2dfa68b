"""Host-side mirror of the reference's column / function / aggregate interfaces for the hot path, over the C ABI.

Names and argument meaning follow the reference (src/Columns/IColumn.h, src/Functions/FunctionsComparison.h,
src/AggregateFunctions/IAggregateFunction.h) so parity tests read like the reference's own tests.  Everything here
runs on the MI355X through libchgpu.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K

NP_OF = {K.I64: np.int64, K.U32: np.uint32, K.U64: np.uint64, K.F64: np.float64, K.U8: np.uint8, K.I32: np.int32,
         K.U16: np.uint16, K.I16: np.int16, K.I8: np.int8, K.F32: np.float32}
TAG_OF = {np.dtype(v): k for k, v in NP_OF.items()}


def sum_result_dtype(tag: int):
    """SumSimple (src/AggregateFunctions/AggregateFunctionSum.cpp:19-28)."""
    if tag in (K.I64, K.I32, K.I16, K.I8):
        return np.int64
    if tag in (K.U64, K.U32, K.U16, K.U8):
        return np.uint64
    return np.float64


_OPTION_ENV_PREFIXES = ("CHGPU_TUNE_", "CHGPU_EXPERIMENT_", "CHGPU_TEST_", "CHGPU_AGG_NO_PARTITION", "CHGPU_DEBUG")


def options_from_env(environ=None) -> dict:
    """The A/B scripts under tools/ select plans with CHGPU_TUNE_* variables.  The library takes options only through chgpu_ctx_set_option;
    this harness turns such variables into those calls when a Context is made (CHGPU_TUNE_GB_NO_TILED=1 -> "tune_gb_no_tiled" = 1)."""
    import os
    env = os.environ if environ is None else environ
    out = {}
    for k, v in env.items():
        if k.startswith(_OPTION_ENV_PREFIXES):
            try:
                out[k[len("CHGPU_"):].lower()] = int(v) if v != "" else 1
            except ValueError:
                out[k[len("CHGPU_"):].lower()] = 1
    return out


def set_default_option(name: str, value: int = 1):
    """process-wide default (chgpu_ctx_set_option with no context): what context-less code such as the expression compiler reads"""
    K.check(K.lib().chgpu_ctx_set_option(None, name.encode(), int(value)))


class Context:
    """One device + one HIP stream (one per pipeline thread, IProcessor.h:176-193)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        h = C.c_void_p()
        K.check(K.lib().chgpu_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h
        self._closed = False
        self.device = device
        for name, value in options_from_env().items():   # developer harness only: the library itself never reads the environment
            try:
                self.set_option(name, value)
                if name == "tune_jit_unroll":          # read by the context-less source generator
                    set_default_option(name, value)
            except K.ChgpuError as e:
                import sys
                print(f"clickhouse_amd: environment option ignored: {e}", file=sys.stderr)

    def set_option(self, name: str, value: int = 1):
        """chgpu_ctx_set_option: a developer option of this context (plan-level A/B switches, launch geometry; tools/README.md)"""
        K.check(K.lib().chgpu_ctx_set_option(self._h, name.encode(), int(value)))

    def close(self):
        """chgpu_ctx_destroy: columns / aggregations / joins made on this context keep the C context alive until the last of them is
        freed (they also keep this object alive), so closing first is safe in any garbage-collection order; making NEW objects on a
        closed context is refused."""
        if getattr(self, "_h", None) and not self._closed:
            K.lib().chgpu_ctx_destroy(self._h)
            self._closed = True

    def _live(self):
        if self._closed:
            raise K.ChgpuError(K.ERR_LOGICAL, "the context has been closed")
        return self._h

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        K.check(K.lib().chgpu_ctx_synchronize(self._h))

    def trim(self):
        """release the cached column pool and scratch arena"""
        K.check(K.lib().chgpu_ctx_trim(self._h))

    def counters(self):
        arr = (C.c_uint64 * K.N_COUNTERS)()
        K.check(K.lib().chgpu_ctx_counters(self._h, arr))
        names = ["FilterTransformPassedRows", "FilterTransformPassedBytes", "JoinBuildTableRowCount", "JoinProbeTableRowCount",
                 "JoinResultRowCount", "AggregatedRows", "KernelLaunches", "TableRehashes"]
        return dict(zip(names, [int(x) for x in arr]))

    def timer_start(self):
        K.check(K.lib().chgpu_timer_start(self._h))

    def timer_stop_ms(self) -> float:
        ms = C.c_double(0)
        K.check(K.lib().chgpu_timer_stop_ms(self._h, C.byref(ms)))
        return ms.value

    # -- column factories -------------------------------------------------------------------
    def upload(self, arr: np.ndarray) -> "Column":
        arr = np.ascontiguousarray(arr)
        h = C.c_void_p()
        K.check(K.lib().chgpu_col_upload(self._live(), TAG_OF[arr.dtype], arr.ctypes.data_as(C.c_void_p), arr.shape[0], C.byref(h)))
        return Column(self, h)

    def alloc(self, dtype, rows: int) -> "Column":
        h = C.c_void_p()
        K.check(K.lib().chgpu_col_alloc(self._live(), TAG_OF[np.dtype(dtype)], rows, C.byref(h)))
        return Column(self, h)

    def wrap(self, device_ptr: int, dtype, rows: int, keepalive=None) -> "Column":
        """Non-owning view of HBM the caller manages (e.g. a torch tensor's data_ptr())."""
        h = C.c_void_p()
        K.check(K.lib().chgpu_col_wrap(self._live(), TAG_OF[np.dtype(dtype)], C.c_void_p(device_ptr), rows, C.byref(h)))
        c = Column(self, h)
        c._keepalive = keepalive
        return c

    def column(self, x) -> "Column":
        return x if isinstance(x, Column) else self.upload(x)


class Column:
    """ColumnVector<T> resident in HBM (src/Columns/ColumnVector.h:28-320)."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self._h = handle
        self._keepalive = None

    def free(self):
        if getattr(self, "_h", None):
            K.lib().chgpu_col_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def size(self) -> int:
        return int(K.lib().chgpu_col_rows(self._h))

    __len__ = size

    @property
    def tag(self) -> int:
        return int(K.lib().chgpu_col_type(self._h))

    @property
    def dtype(self):
        return np.dtype(NP_OF[self.tag])

    @property
    def device_ptr(self) -> int:
        return int(K.lib().chgpu_col_device_ptr(self._h) or 0)

    def numpy(self, rows: int | None = None) -> np.ndarray:
        rows = self.size() if rows is None else rows
        out = np.empty(rows, dtype=self.dtype)
        K.check(K.lib().chgpu_col_download(self.ctx._h, self._h, out.ctypes.data_as(C.c_void_p), rows))
        return out

    @staticmethod
    def numpy_many(cols) -> list:
        """every column of a result Block downloaded with one wait (chgpu_col_download_many)"""
        cols = list(cols)
        if not cols:
            return []
        outs = [np.empty(c.size(), dtype=c.dtype) for c in cols]
        hs = (C.c_void_p * len(cols))(*[c._h for c in cols])
        ps = (C.c_void_p * len(cols))(*[o.ctypes.data for o in outs])
        K.check(K.lib().chgpu_col_download_many(cols[0].ctx._h, len(cols), hs, ps))
        return outs

    def cut(self, start: int, length: int) -> "Column":
        """IColumn::cut (IColumn.h:118-121) as a non-owning view."""
        h = C.c_void_p()
        K.check(K.lib().chgpu_col_slice(self.ctx._h, self._h, start, length, C.byref(h)))
        c = Column(self.ctx, h)
        c._keepalive = self
        return c

    def filter(self, filt: "Column", result_size_hint: int = 0) -> "Column":
        """IColumn::filter (IColumn.h:313-314)."""
        h = C.c_void_p()
        n = C.c_uint64(0)
        K.check(K.lib().chgpu_filter(self.ctx._h, self._h, filt._h, result_size_hint, C.byref(h), C.byref(n)))
        return Column(self.ctx, h)

    def index(self, indexes: "Column", limit: int = 0, default_for_missing: bool = False) -> "Column":
        """IColumn::index (IColumn.h:331)."""
        h = C.c_void_p()
        K.check(K.lib().chgpu_index(self.ctx._h, self._h, indexes._h, limit, int(default_for_missing), C.byref(h)))
        return Column(self.ctx, h)

    def replicate(self, offsets: "Column") -> "Column":
        """IColumn::replicate (IColumn.h:440)."""
        h = C.c_void_p()
        K.check(K.lib().chgpu_replicate(self.ctx._h, self._h, offsets._h, C.byref(h)))
        return Column(self.ctx, h)

    def scatter(self, num_columns: int, selector: "Column"):
        """IColumn::scatter (IColumn.h:448)."""
        outs = (C.c_void_p * num_columns)()
        K.check(K.lib().chgpu_scatter(self.ctx._h, self._h, selector._h, num_columns, outs))
        return [Column(self.ctx, C.c_void_p(outs[i])) for i in range(num_columns)]

    def get_weak_hash32(self, hash_col: "Column | None" = None) -> "Column":
        """IColumn::getWeakHash32 (IColumn.h:302): CRC32-C chained over hash_col (initialised to 0xFFFFFFFF)."""
        if hash_col is None:
            hash_col = self.ctx.upload(np.full(self.size(), 0xFFFFFFFF, dtype=np.uint32))
        K.check(K.lib().chgpu_weak_hash32(self.ctx._h, self._h, hash_col._h))
        return hash_col


def _scalar_buf(tag: int, value):
    return np.array([value], dtype=NP_OF[tag])


def cmp_const(col: Column, op: int, scalar, scalar_tag: int | None = None) -> Column:
    """FunctionComparison with a constant right argument -> UInt8 column (FunctionsComparison.h:204-245)."""
    st = col.tag if scalar_tag is None else scalar_tag
    s = _scalar_buf(st, scalar)
    h = C.c_void_p()
    K.check(K.lib().chgpu_cmp_const(col.ctx._h, col._h, op, st, s.ctypes.data_as(C.c_void_p), C.byref(h)))
    return Column(col.ctx, h)


def count_bytes_in_filter(filt: Column) -> int:
    n = C.c_uint64(0)
    K.check(K.lib().chgpu_count_bytes_in_filter(filt.ctx._h, filt._h, C.byref(n)))
    return int(n.value)


def filter_description_nullable(data: Column, null_map: Column) -> Column:
    h = C.c_void_p()
    K.check(K.lib().chgpu_filter_description_nullable(data.ctx._h, data._h, null_map._h, C.byref(h)))
    return Column(data.ctx, h)


def sum_add_many(col: Column, row_begin: int = 0, row_end: int | None = None, state: np.ndarray | None = None) -> np.ndarray:
    """IAggregateFunction::addBatchSinglePlace for sum (AggregateFunctionSum.h:494-512)."""
    row_end = col.size() if row_end is None else row_end
    st = np.zeros(1, dtype=sum_result_dtype(col.tag)) if state is None else state
    K.check(K.lib().chgpu_sum_add_many(col.ctx._h, col._h, row_begin, row_end, st.ctypes.data_as(C.c_void_p)))
    return st


def sum_add_many_conditional(col: Column, cond: Column, row_begin: int = 0, row_end: int | None = None,
                             state: np.ndarray | None = None) -> np.ndarray:
    row_end = col.size() if row_end is None else row_end
    st = np.zeros(1, dtype=sum_result_dtype(col.tag)) if state is None else state
    K.check(K.lib().chgpu_sum_add_many_conditional(col.ctx._h, col._h, cond._h, row_begin, row_end, st.ctypes.data_as(C.c_void_p)))
    return st


def filter_sum(pred: Column, op: int, scalar, val: Column | None = None, scalar_tag: int | None = None):
    """`SELECT sum(val), count() WHERE pred <op> scalar` fused in one HBM pass -> (sum, count)."""
    val = pred if val is None else val
    st = pred.tag if scalar_tag is None else scalar_tag
    s = _scalar_buf(st, scalar)
    out = np.zeros(1, dtype=sum_result_dtype(val.tag))
    cnt = C.c_uint64(0)
    K.check(K.lib().chgpu_filter_sum(pred.ctx._h, pred._h, op, st, s.ctypes.data_as(C.c_void_p), val._h,
                                     out.ctypes.data_as(C.c_void_p), C.byref(cnt)))
    return out[0], int(cnt.value)


def filter_sum_async(pred: Column, op: int, scalar, val: Column | None, result: Column, scalar_tag: int | None = None):
    """Same without host synchronisation: `result` is a 2-row UInt64 device column {sum bits, count}."""
    val = pred if val is None else val
    st = pred.tag if scalar_tag is None else scalar_tag
    s = _scalar_buf(st, scalar)
    K.check(K.lib().chgpu_filter_sum_async(pred.ctx._h, pred._h, op, st, s.ctypes.data_as(C.c_void_p), val._h, result._h))


def filter_columns(cols, filt: Column, result_size_hint: int = 0):
    """Every column of a Block filtered by one mask (FilterTransform's loop over the chunk's columns): the mask is counted and
    scanned once, one host synchronisation for all columns."""
    n = len(cols)
    cp = (C.c_void_p * max(1, n))(*[c._h for c in cols])
    outs = (C.c_void_p * max(1, n))()
    rows = C.c_uint64(0)
    K.check(K.lib().chgpu_filter_columns(filt.ctx._h, n, cp, filt._h, result_size_hint, outs, C.byref(rows)))
    return [Column(filt.ctx, C.c_void_p(outs[k])) for k in range(n)]


def replicate_columns(cols, offsets: Column):
    """Every column of a Block replicated by one offsets_to_replicate (joinBlock's loop over the left columns): one host synchronisation,
    columns of one width share a kernel."""
    n = len(cols)
    cp = (C.c_void_p * max(1, n))(*[c._h for c in cols])
    outs = (C.c_void_p * max(1, n))()
    K.check(K.lib().chgpu_replicate_columns(offsets.ctx._h, n, cp, offsets._h, outs))
    return [Column(offsets.ctx, C.c_void_p(outs[k])) for k in range(n)]


def hash_to_selector(keys: Column, num_shards: int) -> Column:
    h = C.c_void_p()
    K.check(K.lib().chgpu_hash_to_selector(keys.ctx._h, keys._h, num_shards, C.byref(h)))
    return Column(keys.ctx, h)


def partition_by_hash(keys: Column, num_shards: int, cols):
    """-> (list of partitioned columns, counts[num_shards]); shard s occupies [sum(counts[:s]), +counts[s])."""
    n = len(cols)
    ins = (C.c_void_p * n)(*[c._h for c in cols])
    outs = (C.c_void_p * n)()
    counts = (C.c_uint64 * num_shards)()
    K.check(K.lib().chgpu_partition_by_hash(keys.ctx._h, keys._h, num_shards, n, ins, outs, counts))
    return [Column(keys.ctx, C.c_void_p(outs[i])) for i in range(n)], np.array(list(counts), dtype=np.uint64)


def pack_fixed_keys(cols) -> Column:
    """packFixed<UInt64> over several fixed-width key columns (AggregationCommon.h:91-158): one UInt64 key per row."""
    n = len(cols)
    ins = (C.c_void_p * n)(*[c._h for c in cols])
    h = C.c_void_p()
    K.check(K.lib().chgpu_pack_fixed_keys(cols[0].ctx._h, n, ins, C.byref(h)))
    return Column(cols[0].ctx, h)


def unpack_fixed_key(packed: Column, byte_offset: int, dtype) -> Column:
    h = C.c_void_p()
    K.check(K.lib().chgpu_unpack_fixed_key(packed.ctx._h, packed._h, byte_offset, TAG_OF[np.dtype(dtype)], C.byref(h)))
    return Column(packed.ctx, h)


def and_(a: Column, b: Column) -> Column:
    """FunctionAnd over two UInt8 columns (FunctionsLogical.h:82-96)."""
    h = C.c_void_p()
    K.check(K.lib().chgpu_and(a.ctx._h, a._h, b._h, C.byref(h)))
    return Column(a.ctx, h)


def arith(value_op: int, a: Column, b: Column) -> Column:
    """multiply / plus / minus with NumberTraits result types (multiply.cpp:10-28, NumberTraits.h:73-87)."""
    h = C.c_void_p()
    K.check(K.lib().chgpu_arith(a.ctx._h, value_op, a._h, b._h, C.byref(h)))
    return Column(a.ctx, h)


def scalar_bits(tag: int, value) -> int:
    return int.from_bytes(np.array([value], dtype=NP_OF[tag]).tobytes().ljust(8, b"\0"), "little")


def expr_filter_sum(cols, preds, value_op: int, val_a: int, val_b: int = 0):
    """Fused `SELECT sum(value), count() WHERE p0 AND p1 ...`; preds: (col_index, op, scalar[, scalar_tag]) -> (sum, count)."""
    n = len(cols)
    cp = (C.c_void_p * n)(*[c._h for c in cols])
    m = len(preds)
    pc = (C.c_uint32 * max(1, m))(*[p[0] for p in preds])
    po = (C.c_int * max(1, m))(*[p[1] for p in preds])
    tags = [(p[3] if len(p) > 3 and p[3] is not None else cols[p[0]].tag) for p in preds]
    ps = (C.c_int * max(1, m))(*tags)
    pb = (C.c_uint64 * max(1, m))(*[scalar_bits(t, p[2]) for p, t in zip(preds, tags)])
    rt = C.c_int(0)
    out = np.zeros(1, dtype=np.uint64)
    cnt = C.c_uint64(0)
    K.check(K.lib().chgpu_expr_filter_sum(cols[0].ctx._h, n, cp, m, pc, po, ps, pb, value_op, val_a, val_b, C.byref(rt),
                                          out.ctypes.data_as(C.c_void_p), C.byref(cnt)))
    return out.view(NP_OF[rt.value])[0], int(cnt.value)


def sort_permutation(col: Column, perm_in: Column | None = None, descending: bool = False, nan_direction_hint: int = 1) -> Column:
    """IColumn::getPermutation(direction, Stable, 0, nan_direction_hint) -> UInt64 permutation Column; perm_in composes a previous
    (less significant) sort: ORDER BY a, b == sort_permutation(a, perm_in=sort_permutation(b))"""
    h = C.c_void_p()
    K.check(K.lib().chgpu_sort_permutation(col.ctx._h, col._h, perm_in._h if perm_in is not None else None, 1 if descending else 0,
                                           nan_direction_hint, C.byref(h)))
    return Column(col.ctx, h)


def sort_permutation_limit(col: Column, limit: int, descending: bool = False, nan_direction_hint: int = 1) -> Column:
    """the first `limit` entries of the stable sorted permutation (ORDER BY ... LIMIT n over one column)"""
    h = C.c_void_p()
    K.check(K.lib().chgpu_sort_permutation_limit(col.ctx._h, col._h, 1 if descending else 0, nan_direction_hint, limit, C.byref(h)))
    return Column(col.ctx, h)


def filter_to_indices(filt: Column) -> Column:
    """filterToIndices: ascending row numbers of the non-zero filter bytes"""
    h = C.c_void_p()
    n = C.c_uint64(0)
    K.check(K.lib().chgpu_filter_to_indices(filt.ctx._h, filt._h, C.byref(h), C.byref(n)))
    return Column(filt.ctx, h)


def sort_block(columns, description, limit: int = 0):
    """sortBlock (src/Interpreters/sortBlock.cpp): description = [(position, descending, nan_direction_hint), ...] most significant
    first; every column of the block permuted (IColumn::permute == index), cut to `limit` rows when given."""
    description = list(description)
    if limit and len(description) == 1:
        pos, desc, hint = description[0]
        perm = sort_permutation_limit(columns[pos], limit, desc, hint)  # one sort column + LIMIT: only the candidate rows are sorted
        return [c.index(perm) for c in columns], perm
    perm = None
    for pos, desc, hint in reversed(description):
        perm = sort_permutation(columns[pos], perm, desc, hint)
    if limit:
        perm = perm.cut(0, min(limit, perm.size()))
    return [c.index(perm) for c in columns], perm


def concat(cols) -> Column:
    """glue columns of one type end to end (right-side Blocks -> one payload column)"""
    n = len(cols)
    ins = (C.c_void_p * n)(*[c._h for c in cols])
    h = C.c_void_p()
    K.check(K.lib().chgpu_col_concat(cols[0].ctx._h, n, ins, C.byref(h)))
    return Column(cols[0].ctx, h)
