"""CPU oracle — TEST INFRASTRUCTURE ONLY.

ctypes binding over oracle/libchoracle.so (the C restatement of the reference's hot path, oracle/ch_oracle.c)
and, when present, oracle/_ref/libchref_hash.so (the reference's own Hash.h compiled in place).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (clickhouse_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# type tags (same numeric values as include/chgpu.h)
I64, U32, U64, F64, U8, I32, U16, I16, I8, F32 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
EQ, NE, LT, GT, LE, GE = 0, 1, 2, 3, 4, 5
AGG_COUNT, AGG_SUM, AGG_AVG, AGG_MIN, AGG_MAX, AGG_ANY = 0, 1, 2, 3, 4, 5
JOIN_INNER, JOIN_LEFT = 0, 1
STRICT_ANY, STRICT_ALL, STRICT_SEMI, STRICT_ANTI = 0, 1, 2, 3
DEFAULT_BLOCK_SIZE = 65409

NP_OF = {I64: np.int64, U32: np.uint32, U64: np.uint64, F64: np.float64, U8: np.uint8, I32: np.int32, U16: np.uint16, I16: np.int16, I8: np.int8, F32: np.float32}
TAG_OF = {np.dtype(v): k for k, v in NP_OF.items()}


def tag_of(arr: np.ndarray) -> int:
    return TAG_OF[arr.dtype]


def sum_result_dtype(tag: int):
    """SumSimple result type (AggregateFunctionSum.cpp:19-28)."""
    if tag in (I64, I32, I16, I8):
        return np.int64
    if tag in (U64, U32, U16, U8):
        return np.uint64
    return np.float64


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "libchoracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("ch_oracle.c", "ch_oracle.h", "ch_hashtable.inc")
    ):
        subprocess.check_call(["make", "-C", _HERE, "libchoracle.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _p(arr):
    return arr.ctypes.data_as(C.c_void_p) if arr is not None else None


_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        so = os.environ.get("CHO_LIB", os.path.join(_HERE, "libchoracle.so"))  # CHO_LIB: sanitizer builds of the oracle
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        u64, sz, vp, i32 = C.c_uint64, C.c_size_t, C.c_void_p, C.c_int
        sig = {
            "cho_intHash64": (u64, [u64]),
            "cho_intHashCRC32": (u64, [u64]),
            "cho_intHashCRC32_seed": (u64, [u64, u64]),
            "cho_intHashCRC32_soft": (u64, [u64, u64]),
            "cho_intHash32": (C.c_uint32, [u64, u64]),
            "cho_sql_intHash64": (u64, [u64]),
            "cho_sql_intHash32": (C.c_uint32, [u64]),
            "cho_hash_crc32_batch": (None, [i32, vp, sz, vp]),
            "cho_weak_hash32": (None, [i32, vp, sz, vp]),
            "cho_two_level_bucket": (C.c_uint32, [u64]),
            "cho_crc32c_tables": (None, [vp, vp]),
            "cho_cmp_const": (i32, [i32, vp, sz, i32, i32, vp, vp]),
            "cho_bytes64MaskToBits64Mask": (u64, [vp]),
            "cho_countBytesInFilter": (sz, [vp, sz, sz]),
            "cho_filter": (C.c_int64, [i32, vp, sz, vp, sz, vp]),
            "cho_filter_description_nullable": (None, [vp, vp, sz, vp]),
            "cho_index": (None, [i32, vp, vp, sz, vp]),
            "cho_replicate": (None, [i32, vp, sz, vp, vp]),
            "cho_scatter": (None, [i32, vp, sz, vp, sz, vp, vp]),
            "cho_sum_add_many": (None, [i32, vp, vp, sz, sz]),
            "cho_sum_add_many_conditional": (None, [i32, vp, vp, vp, sz, sz]),
            "cho_avg_divide": (C.c_double, [i32, vp, u64]),
            "cho_filter_sum_pipeline": (i32, [i32, vp, vp, sz, i32, vp, sz, i32, vp, vp, vp, vp]),
            "cho_and_u8": (None, [vp, vp, sz, vp]),
            "cho_arith_result_type": (i32, [i32, i32, i32]),
            "cho_arith_sum_type": (i32, [i32, i32, i32]),
            "cho_arith": (i32, [i32, i32, vp, i32, vp, sz, vp]),
            "cho_expr_filter_sum_pipeline": (i32, [sz, vp, vp, sz, sz, vp, vp, vp, vp, i32, C.c_uint32, C.c_uint32, sz, i32, vp, vp]),
            "cho_hashmap_create": (vp, []),
            "cho_hashmap_free": (None, [vp]),
            "cho_hashmap_emplace": (i32, [vp, u64, C.POINTER(C.POINTER(u64))]),
            "cho_hashmap_find": (C.POINTER(u64), [vp, u64]),
            "cho_hashmap_reserve": (None, [vp, sz]),
            "cho_hashmap_size": (sz, [vp]),
            "cho_hashmap_buf_size": (sz, [vp]),
            "cho_hashmap_has_zero": (i32, [vp]),
            "cho_hashmap_dump": (sz, [vp, vp, vp]),
            "cho_agg_create": (vp, [i32, i32, vp, vp, u64]),
            "cho_agg_free": (None, [vp]),
            "cho_agg_execute_on_block": (i32, [vp, vp, vp, sz, sz]),
            "cho_agg_merge": (i32, [vp, vp]),
            "cho_agg_size": (sz, [vp]),
            "cho_agg_is_two_level": (i32, [vp]),
            "cho_agg_convert_to_block": (sz, [vp, vp, vp]),
            "cho_join_create": (vp, [i32, i32, i32]),
            "cho_join_free": (None, [vp]),
            "cho_join_add_block": (C.c_int64, [vp, vp, sz, vp, vp]),
            "cho_join_total_rows": (sz, [vp]),
            "cho_join_keys": (sz, [vp]),
            "cho_join_probe": (sz, [vp, vp, sz, vp, sz, vp, vp, vp, vp, sz, vp]),
            "cho_join_need_filter": (i32, [vp]),
            "cho_join_need_replication": (i32, [vp]),
            "cho_hash_to_selector": (None, [i32, vp, sz, sz, vp]),
            "cho_pack_fixed": (None, [sz, vp, vp, sz, sz, vp]),
            "cho_hash_keys_fixed": (u64, [vp, sz]),
            "cho_widemap_create": (vp, [sz]),
            "cho_widemap_free": (None, [vp]),
            "cho_widemap_size": (sz, [vp]),
            "cho_widemap_batch": (None, [vp, vp, sz, i32, vp]),
            "cho_widemap_keys": (None, [vp, vp]),
            "cho_groupby_pipeline": (vp, [i32, i32, vp, vp, vp, vp, sz, sz, i32, u64, vp]),
            "cho_join_count_sum_pipeline": (i32, [vp, vp, sz, vp, sz, sz, i32, vp, vp, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def ref_hash():
    """The reference's own Hash.h compiled in place (oracle/_ref); None when it was never built."""
    global _ref
    if _ref is None:
        so = os.path.join(_HERE, "_ref", "libchref_hash.so")
        if not os.path.exists(so):
            return None
        R = C.CDLL(so)
        u64 = C.c_uint64
        for name, res, args in [
            ("ref_intHash64", u64, [u64]),
            ("ref_intHashCRC32", u64, [u64]),
            ("ref_intHashCRC32_seed", u64, [u64, u64]),
            ("ref_intHash32_salt0", C.c_uint32, [u64]),
            ("ref_intHash32_sql", C.c_uint32, [u64]),
            ("ref_HashCRC32_UInt64", u64, [u64]),
            ("ref_HashCRC32_UInt32", u64, [C.c_uint32]),
            ("ref_HashCRC32_Int64", u64, [C.c_int64]),
            ("ref_hashCRC32_UInt64_seed", u64, [u64, u64]),
            ("ref_hashCRC32_UInt32_seed", u64, [C.c_uint32, u64]),
            ("ref_intHashCRC32_batch", None, [C.c_void_p, u64, C.c_void_p]),
            ("ref_intHash64_batch", None, [C.c_void_p, u64, C.c_void_p]),
            ("ref_UInt128HashCRC32", u64, [u64, u64]),
            ("ref_UInt256HashCRC32", u64, [u64, u64, u64, u64]),
        ]:
            fn = getattr(R, name)
            fn.restype = res
            fn.argtypes = args
        _ref = R
    return _ref


# ---------------------------------------------------------------------------------------------
# numpy-level helpers (what the tests call)
# ---------------------------------------------------------------------------------------------

def _scalar(tag: int, value):
    return np.array([value], dtype=NP_OF[tag])


def hash_crc32(keys: np.ndarray) -> np.ndarray:
    keys = np.ascontiguousarray(keys)
    out = np.empty(keys.shape[0], dtype=np.uint64)
    lib().cho_hash_crc32_batch(tag_of(keys), _p(keys), keys.shape[0], _p(out))
    return out


def weak_hash32(data: np.ndarray, seed: np.ndarray | None = None) -> np.ndarray:
    data = np.ascontiguousarray(data)
    h = np.full(data.shape[0], 0xFFFFFFFF, dtype=np.uint32) if seed is None else seed.astype(np.uint32).copy()
    lib().cho_weak_hash32(tag_of(data), _p(data), data.shape[0], _p(h))
    return h


def crc32c_tables():
    t = np.zeros((8, 256), dtype=np.uint32)
    c = np.zeros(1, dtype=np.uint32)
    lib().cho_crc32c_tables(_p(t), _p(c))
    return t, int(c[0])


def cmp_const(a: np.ndarray, op: int, scalar, scalar_tag: int | None = None) -> np.ndarray:
    a = np.ascontiguousarray(a)
    st = tag_of(a) if scalar_tag is None else scalar_tag
    s = _scalar(st, scalar)
    out = np.empty(a.shape[0], dtype=np.uint8)
    rc = lib().cho_cmp_const(tag_of(a), _p(a), a.shape[0], op, st, _p(s), _p(out))
    assert rc == 0
    return out


def count_bytes_in_filter(filt: np.ndarray) -> int:
    filt = np.ascontiguousarray(filt, dtype=np.uint8)
    return int(lib().cho_countBytesInFilter(_p(filt), 0, filt.shape[0]))


def filter_column(data: np.ndarray, filt: np.ndarray) -> np.ndarray:
    data = np.ascontiguousarray(data)
    filt = np.ascontiguousarray(filt, dtype=np.uint8)
    out = np.empty(data.shape[0] + 64, dtype=data.dtype)
    n = lib().cho_filter(data.dtype.itemsize, _p(data), data.shape[0], _p(filt), filt.shape[0], _p(out))
    if n < 0:
        raise ValueError("SIZES_OF_COLUMNS_DOESNT_MATCH")
    return out[:n].copy()


def index_column(data: np.ndarray, indexes: np.ndarray, limit: int | None = None) -> np.ndarray:
    data = np.ascontiguousarray(data)
    idx = np.ascontiguousarray(indexes, dtype=np.uint64)
    limit = idx.shape[0] if limit is None else limit
    out = np.empty(limit, dtype=data.dtype)
    lib().cho_index(data.dtype.itemsize, _p(data), _p(idx), limit, _p(out))
    return out


def replicate(data: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    data = np.ascontiguousarray(data)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    total = int(off[-1]) if off.shape[0] else 0
    out = np.empty(total, dtype=data.dtype)
    lib().cho_replicate(data.dtype.itemsize, _p(data), data.shape[0], _p(off), _p(out))
    return out


def scatter(data: np.ndarray, selector: np.ndarray, num_columns: int):
    data = np.ascontiguousarray(data)
    sel = np.ascontiguousarray(selector, dtype=np.uint64)
    out = np.empty(data.shape[0], dtype=data.dtype)
    sizes = np.zeros(num_columns, dtype=np.uint64)
    lib().cho_scatter(data.dtype.itemsize, _p(data), data.shape[0], _p(sel), num_columns, _p(out), _p(sizes))
    res, pos = [], 0
    for k in range(num_columns):
        res.append(out[pos:pos + int(sizes[k])].copy())
        pos += int(sizes[k])
    return res


def sum_add_many(data: np.ndarray, start: int = 0, end: int | None = None, state=None):
    data = np.ascontiguousarray(data)
    end = data.shape[0] if end is None else end
    st = np.zeros(1, dtype=sum_result_dtype(tag_of(data))) if state is None else state
    lib().cho_sum_add_many(tag_of(data), _p(st), _p(data), start, end)
    return st


def sum_add_many_conditional(data: np.ndarray, cond: np.ndarray, state=None):
    data = np.ascontiguousarray(data)
    cond = np.ascontiguousarray(cond, dtype=np.uint8)
    st = np.zeros(1, dtype=sum_result_dtype(tag_of(data))) if state is None else state
    lib().cho_sum_add_many_conditional(tag_of(data), _p(st), _p(data), _p(cond), 0, data.shape[0])
    return st


def filter_sum_pipeline(pred: np.ndarray, op: int, scalar, val: np.ndarray | None = None,
                        block_rows: int = DEFAULT_BLOCK_SIZE, threads: int = 1):
    """The C1/C2 query `SELECT sum(val), count() WHERE pred <op> scalar` through the per-Block pipeline."""
    pred = np.ascontiguousarray(pred)
    if val is not None:
        val = np.ascontiguousarray(val)
        assert val.dtype == pred.dtype and val.shape == pred.shape
    t = tag_of(pred)
    s = _scalar(t, scalar)
    out = np.zeros(1, dtype=sum_result_dtype(t))
    cnt = np.zeros(1, dtype=np.uint64)
    dropped = np.zeros(1, dtype=np.uint64)
    passed = np.zeros(1, dtype=np.uint64)
    rc = lib().cho_filter_sum_pipeline(t, _p(pred), _p(val), pred.shape[0], op, _p(s), block_rows, threads,
                                       _p(out), _p(cnt), _p(dropped), _p(passed))
    assert rc == 0
    return out[0], int(cnt[0]), int(dropped[0]), int(passed[0])


VAL_COL, VAL_MUL, VAL_PLUS, VAL_MINUS = 0, 1, 2, 3


def and_u8(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    out = np.empty_like(a)
    lib().cho_and_u8(_p(a), _p(b), a.shape[0], _p(out))
    return out


def arith(value_op, a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    rt = lib().cho_arith_result_type(value_op, tag_of(a), tag_of(b))
    assert rt >= 0
    out = np.empty(a.shape[0], dtype=NP_OF[rt])
    assert lib().cho_arith(value_op, tag_of(a), _p(a), tag_of(b), _p(b), a.shape[0], _p(out)) == 0
    return out


def scalar_bits(tag, value) -> int:
    """the 8 raw little-endian bytes of `value` typed as `tag` (how constants cross the C boundary)"""
    return int.from_bytes(np.array([value], dtype=NP_OF[tag]).tobytes().ljust(8, b"\0"), "little")


def expr_filter_sum_pipeline(cols, preds, value_op, val_a, val_b=0, block_rows=DEFAULT_BLOCK_SIZE, threads=1):
    """cols: list of ndarrays; preds: list of (col_index, op, scalar, scalar_tag or None) and-ed together."""
    cols = [np.ascontiguousarray(c) for c in cols]
    n = cols[0].shape[0]
    types = np.array([tag_of(c) for c in cols], dtype=np.int32)
    ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    pc = np.array([p[0] for p in preds], dtype=np.uint32)
    po = np.array([p[1] for p in preds], dtype=np.int32)
    ps = np.array([(p[3] if len(p) > 3 and p[3] is not None else tag_of(cols[p[0]])) for p in preds], dtype=np.int32)
    pb = np.array([scalar_bits(int(t), p[2]) for p, t in zip(preds, ps)], dtype=np.uint64)
    ta = tag_of(cols[val_a])
    rt = sum_result_dtype(ta) if value_op == VAL_COL else NP_OF[lib().cho_arith_sum_type(value_op, ta, tag_of(cols[val_b]))]
    out = np.zeros(1, dtype=rt)
    cnt = np.zeros(1, dtype=np.uint64)
    rc = lib().cho_expr_filter_sum_pipeline(len(cols), _p(types), ptrs, n, len(preds), _p(pc), _p(po), _p(ps), _p(pb), value_op, val_a, val_b,
                                            block_rows, threads, _p(out), _p(cnt))
    assert rc == 0
    return out[0], int(cnt[0])


class HashMap:
    """HashMap<UInt64, UInt64, HashCRC32<UInt64>> restated (gtest_hash_table scenarios)."""

    def __init__(self):
        self._h = lib().cho_hashmap_create()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().cho_hashmap_free(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def emplace(self, key: int, value: int | None = None) -> bool:
        mp = C.POINTER(C.c_uint64)()
        ins = lib().cho_hashmap_emplace(self._h, key, C.byref(mp))
        if ins and value is not None:
            mp[0] = value
        return bool(ins)

    def add(self, key: int, delta: int):
        mp = C.POINTER(C.c_uint64)()
        lib().cho_hashmap_emplace(self._h, key, C.byref(mp))
        mp[0] = (mp[0] + delta) & 0xFFFFFFFFFFFFFFFF

    def find(self, key: int):
        p = lib().cho_hashmap_find(self._h, key)
        return int(p[0]) if p else None

    def reserve(self, n: int):
        lib().cho_hashmap_reserve(self._h, n)

    def __len__(self):
        return int(lib().cho_hashmap_size(self._h))

    @property
    def buf_size(self):
        return int(lib().cho_hashmap_buf_size(self._h))

    @property
    def has_zero(self):
        return bool(lib().cho_hashmap_has_zero(self._h))

    def dump(self):
        n = len(self)
        k = np.empty(n, dtype=np.uint64)
        v = np.empty(n, dtype=np.uint64)
        m = lib().cho_hashmap_dump(self._h, _p(k), _p(v))
        assert m == n
        return k, v


class Aggregator:
    """Aggregator restated (executeOnBlock / merge / convertToBlocks). key_dtype None = without_key."""

    def __init__(self, key_dtype, aggs, two_level_threshold: int = 100000):
        # aggs: list of (kind, arg_dtype or None)
        self.key_tag = -1 if key_dtype is None else TAG_OF[np.dtype(key_dtype)]
        self.aggs = [(k, (TAG_OF[np.dtype(d)] if d is not None else I64)) for k, d in aggs]
        kinds = np.array([k for k, _ in self.aggs], dtype=np.int32)
        types = np.array([t for _, t in self.aggs], dtype=np.int32)
        self._h = lib().cho_agg_create(self.key_tag, len(self.aggs), _p(kinds), _p(types), two_level_threshold)
        assert self._h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().cho_agg_free(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def execute_on_block(self, keys, args, row_begin: int = 0, row_end: int | None = None):
        n = (keys.shape[0] if keys is not None else next(a for a in args if a is not None).shape[0])
        row_end = n if row_end is None else row_end
        keep = [np.ascontiguousarray(a) if a is not None else None for a in args]
        ptrs = (C.c_void_p * max(1, len(keep)))(*[(a.ctypes.data if a is not None else None) for a in keep])
        k = np.ascontiguousarray(keys) if keys is not None else None
        rc = lib().cho_agg_execute_on_block(self._h, _p(k), ptrs, row_begin, row_end)
        assert rc == 0

    def merge(self, other: "Aggregator"):
        assert lib().cho_agg_merge(self._h, other._h) == 0

    def __len__(self):
        return int(lib().cho_agg_size(self._h))

    @property
    def is_two_level(self):
        return bool(lib().cho_agg_is_two_level(self._h))

    def result_dtypes(self):
        out = []
        for kind, t in self.aggs:
            if kind == AGG_COUNT:
                out.append(np.uint64)
            elif kind == AGG_AVG:
                out.append(np.float64)
            elif kind in (AGG_MIN, AGG_MAX, AGG_ANY):
                out.append(NP_OF[t])      # the argument's own type (AggregateFunctionsMinMax.cpp)
            else:
                out.append(sum_result_dtype(t))
        return out

    def convert_to_block(self):
        n = len(self)
        keys = np.empty(n, dtype=NP_OF[self.key_tag]) if self.key_tag >= 0 else None
        # the C side emits 8 bytes per state: min / max arrive widened (sign- / zero-extended, Float32 as Float64) and are narrowed here
        sv = (AGG_MIN, AGG_MAX, AGG_ANY)
        wide = [np.float64 if (k in sv and np.dtype(d).kind == "f") else np.int64 if (k in sv and np.dtype(d).kind == "i")
                else np.uint64 if k in sv else d for (k, _), d in zip(self.aggs, self.result_dtypes())]
        res = [np.zeros(n, dtype=d) for d in wide]
        ptrs = (C.c_void_p * max(1, len(res)))(*[r.ctypes.data for r in res])
        m = lib().cho_agg_convert_to_block(self._h, _p(keys), ptrs)
        assert m == n
        return keys, [r.astype(d) if r.dtype != np.dtype(d) else r for r, d in zip(res, self.result_dtypes())]


class HashJoin:
    """HashJoin (key64) restated: addBlockToJoin / joinBlock's joinRightColumns."""

    def __init__(self, kind: int, strictness: int, any_take_last_row: bool = False):
        self._h = lib().cho_join_create(kind, strictness, int(any_take_last_row))
        if not self._h:
            raise NotImplementedError("unsupported join kind/strictness")
        self.kind, self.strictness = kind, strictness

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().cho_join_free(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    @property
    def need_filter(self):
        return bool(lib().cho_join_need_filter(self._h))

    @property
    def need_replication(self):
        return bool(lib().cho_join_need_replication(self._h))

    def add_block(self, keys, null_map=None, join_mask=None) -> int:
        k = np.ascontiguousarray(keys).astype(np.uint64, copy=False)
        nm = np.ascontiguousarray(null_map, dtype=np.uint8) if null_map is not None else None
        jm = np.ascontiguousarray(join_mask, dtype=np.uint8) if join_mask is not None else None
        return int(lib().cho_join_add_block(self._h, _p(k), k.shape[0], _p(nm), _p(jm)))

    @property
    def total_rows(self):
        return int(lib().cho_join_total_rows(self._h))

    @property
    def n_keys(self):
        return int(lib().cho_join_keys(self._h))

    def probe(self, keys, null_map=None, max_joined_block_rows: int = 0):
        """returns dict(consumed, filter, offsets, added_block, added_row)."""
        k = np.ascontiguousarray(keys).astype(np.uint64, copy=False)
        rows = k.shape[0]
        nm = np.ascontiguousarray(null_map, dtype=np.uint8) if null_map is not None else None
        filt = np.zeros(rows, dtype=np.uint8)
        offs = np.zeros(rows, dtype=np.uint64)
        cap = max(16, rows * 2)
        while True:
            ab = np.empty(cap, dtype=np.int64)
            ar = np.empty(cap, dtype=np.int64)
            n_added = C.c_size_t(0)
            # non-ALL strictness appends <= rows entries < cap, so the (flag-mutating) INNER ANY probe never retries
            consumed = lib().cho_join_probe(self._h, _p(k), rows, _p(nm), max_joined_block_rows, _p(filt), _p(offs),
                                            _p(ab), _p(ar), cap, C.byref(n_added))
            if n_added.value <= cap:
                break
            cap = n_added.value
        n = n_added.value
        return dict(consumed=int(consumed), filter=filt[:consumed] if self.need_filter else None,
                    offsets=offs[:consumed] if self.need_replication else None,
                    added_block=ab[:n].copy(), added_row=ar[:n].copy())

    def joined_pairs(self, keys, null_map=None, max_joined_block_rows: int = 0):
        """Canonical observable result: (left_row, right_block, right_row) per joined row, in output order."""
        r = self.probe(keys, null_map, max_joined_block_rows)
        c = r["consumed"]
        if self.need_replication:
            off = r["offsets"]
            counts = np.diff(np.concatenate([[0], off])).astype(np.int64)
            left = np.repeat(np.arange(c, dtype=np.int64), counts)
        elif self.need_filter:
            left = np.nonzero(r["filter"])[0].astype(np.int64)
        else:
            left = np.arange(c, dtype=np.int64)
        assert left.shape[0] == r["added_block"].shape[0], (left.shape, r["added_block"].shape)
        return left, r["added_block"], r["added_row"], c


def pack_fixed(key_cols, key_bytes: int | None = None) -> np.ndarray:
    """packFixed (AggregationCommon.h:91-158) -> uint8[n, key_bytes]; key_bytes None: 16 when the columns fit, else 32"""
    cols = [np.ascontiguousarray(c) for c in key_cols]
    total = sum(c.dtype.itemsize for c in cols)
    key_bytes = key_bytes or (16 if total <= 16 else 32)
    assert total <= key_bytes
    n = cols[0].shape[0]
    out = np.empty((n, key_bytes), dtype=np.uint8)
    sizes = np.array([c.dtype.itemsize for c in cols], dtype=np.uint32)
    ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    lib().cho_pack_fixed(len(cols), _p(sizes), ptrs, n, key_bytes, _p(out))
    return out


def hash_keys_fixed(packed_row: np.ndarray) -> int:
    """UInt128HashCRC32 / UInt256HashCRC32 of one packed key (uint8[16] or uint8[32])"""
    w = np.ascontiguousarray(packed_row).view(np.uint64)
    return int(lib().cho_hash_keys_fixed(_p(w), w.shape[0]))


class WideKeyMap:
    """HashMap<UInt128 / UInt256, id> restated (keys128 / keys256): emplace / find of packed keys -> ids by first appearance"""

    def __init__(self, key_bytes: int):
        self.key_bytes = key_bytes
        self._h = lib().cho_widemap_create(key_bytes)
        assert self._h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().cho_widemap_free(self._h)
                self._h = None
        except Exception:
            pass

    def batch(self, packed: np.ndarray, insert: bool = True) -> np.ndarray:
        packed = np.ascontiguousarray(packed)
        ids = np.empty(packed.shape[0], dtype=np.uint64)
        lib().cho_widemap_batch(self._h, _p(packed), packed.shape[0], int(insert), _p(ids))
        return ids

    def __len__(self):
        return int(lib().cho_widemap_size(self._h))

    def keys(self) -> np.ndarray:
        out = np.empty((len(self), self.key_bytes), dtype=np.uint8)
        lib().cho_widemap_keys(self._h, _p(out))
        return out


class KeysFixedAggregator:
    """Aggregator with the keys128 / keys256 method: HashMethodKeysFixed over packFixed keys; states per key as in Aggregator"""

    def __init__(self, key_dtypes, aggs):
        self.key_dtypes = [np.dtype(d) for d in key_dtypes]
        total = sum(d.itemsize for d in self.key_dtypes)
        self.map = WideKeyMap(16 if total <= 16 else 32)
        self.inner = Aggregator(np.uint64, aggs, two_level_threshold=0)

    def execute_on_block(self, key_cols, args):
        packed = pack_fixed([np.ascontiguousarray(c).astype(d, copy=False) for c, d in zip(key_cols, self.key_dtypes)], self.map.key_bytes)
        self.inner.execute_on_block(self.map.batch(packed, True), args)

    def convert_to_block(self):
        ids, res = self.inner.convert_to_block()
        keys = self.map.keys()[ids.astype(np.int64)]
        cols, off = [], 0
        for d in self.key_dtypes:
            cols.append(np.ascontiguousarray(keys[:, off:off + d.itemsize]).view(d).reshape(-1))
            off += d.itemsize
        return cols, res


def groupby_pipeline(keys: np.ndarray, aggs, args, threads: int = 1, block_rows: int = DEFAULT_BLOCK_SIZE, two_level_threshold: int = 100000):
    """N pipeline streams over Blocks -> merged Aggregator, (consume_s, merge_s).  The CPU baseline of config C3."""
    A = Aggregator.__new__(Aggregator)
    A.key_tag = TAG_OF[np.dtype(keys.dtype)]
    A.aggs = [(k, (TAG_OF[np.dtype(d)] if d is not None else I64)) for k, d in aggs]
    kinds = np.array([k for k, _ in A.aggs], dtype=np.int32)
    types = np.array([t for _, t in A.aggs], dtype=np.int32)
    keep = [np.ascontiguousarray(a) if a is not None else None for a in args]
    ptrs = (C.c_void_p * max(1, len(keep)))(*[(a.ctypes.data if a is not None else None) for a in keep])
    k = np.ascontiguousarray(keys)
    secs = np.zeros(2, dtype=np.float64)
    A._h = lib().cho_groupby_pipeline(A.key_tag, len(A.aggs), _p(kinds), _p(types), _p(k), ptrs, k.shape[0], block_rows, threads,
                                      two_level_threshold, _p(secs))
    assert A._h
    return A, (float(secs[0]), float(secs[1]))


def join_count_sum_pipeline(bk: np.ndarray, bv: np.ndarray, pk: np.ndarray, threads: int = 1, block_rows: int = DEFAULT_BLOCK_SIZE):
    """SELECT count(), sum(bv) FROM probe INNER JOIN build ON pk = bk -> (count, sum as u64 bits, build_s, probe_s).  CPU baseline of C4."""
    bk = np.ascontiguousarray(bk).view(np.uint64)
    pk = np.ascontiguousarray(pk).view(np.uint64)
    bv = np.ascontiguousarray(bv).view(np.int64)
    cnt, sm = C.c_uint64(0), C.c_uint64(0)
    secs = np.zeros(2, dtype=np.float64)
    rc = lib().cho_join_count_sum_pipeline(_p(bk), _p(bv), bk.shape[0], _p(pk), pk.shape[0], block_rows, threads, C.byref(cnt), C.byref(sm), _p(secs))
    assert rc == 0
    return int(cnt.value), int(sm.value), float(secs[0]), float(secs[1])


def hash_to_selector(keys: np.ndarray, num_shards: int) -> np.ndarray:
    keys = np.ascontiguousarray(keys)
    out = np.empty(keys.shape[0], dtype=np.uint64)
    lib().cho_hash_to_selector(tag_of(keys), _p(keys), keys.shape[0], num_shards, _p(out))
    return out


ASOF_LESS, ASOF_GREATER, ASOF_LESS_OR_EQUALS, ASOF_GREATER_OR_EQUALS = 1, 2, 3, 4


def asof_pairs(build_blocks, left_keys, left_asof, inequality: int, left_null_map=None, left_join: bool = False):
    """ASOF join restated from the reference's data structure: per key a SortedLookupVector of (asof value, row) (RowRefs.cpp:40-99 insert /
    sort: descending for > and >=, ascending for < and <=) and findAsof -> boundSearch (:100-166): the first entry, in that order, which the
    left value is allowed to meet -- `value >= v` (>=), `value > v` (>), `value <= v` (<=), `value < v` (<).  joinRightColumns emits that
    row (HashJoinMethodsImpl.h:462-478); INNER drops the left rows without one, LEFT keeps them with a default row.
    Rows with a NULL key / zero ON mask / NaN asof value are not inserted.  Equal (key, asof) right rows: the reference's pick is
    unspecified -- build sides of the tests have none.
    build_blocks: [(keys, asof, null_map or None, join_mask or None)] -> [(left_row, block, row)] in left-row order ((-1, -1) = default row)"""
    import bisect
    vec = {}
    for b, (keys, asof, nm, jm) in enumerate(build_blocks):
        for r in range(keys.shape[0]):
            if (nm is not None and nm[r]) or (jm is not None and not jm[r]) or asof[r] != asof[r]:
                continue
            vec.setdefault(int(keys[r]), []).append((asof[r].item(), b, r))
    for v in vec.values():
        v.sort(key=lambda e: e[0])              # ascending here; the descending order of > / >= is walked from the other end
    out = []
    for i in range(left_keys.shape[0]):
        hit = None
        t = left_asof[i]
        if not (left_null_map is not None and left_null_map[i]) and t == t:
            v = vec.get(int(left_keys[i]))
            if v:
                vals = [e[0] for e in v]
                t = t.item()
                if inequality == ASOF_GREATER_OR_EQUALS:
                    p = bisect.bisect_right(vals, t) - 1      # the greatest v <= t
                elif inequality == ASOF_GREATER:
                    p = bisect.bisect_left(vals, t) - 1       # the greatest v < t
                elif inequality == ASOF_LESS_OR_EQUALS:
                    p = bisect.bisect_left(vals, t)           # the smallest v >= t
                else:
                    p = bisect.bisect_right(vals, t)          # the smallest v > t
                if 0 <= p < len(v):
                    hit = v[p]
        if hit is not None:
            out.append((i, hit[1], hit[2]))
        elif left_join:
            out.append((i, -1, -1))
    return out


def right_once_pairs(build_blocks, probe_key_batches, anti: bool = False):
    """RIGHT ANY / RIGHT SEMI (and the flags of RIGHT ANTI) restated from joinRightColumns (HashJoinMethodsImpl.h:487-497, :515-519): the map
    is MapsAll (joinDispatch.h:37,53,61: every inserted right row is kept) with ONE flag per key; left rows are taken in order, over all
    probed blocks, and the first one to find a key sets its flag (setUsedOnce) and is joined with ALL right rows of that key
    (addFoundRowAll); every later left row with the same key adds nothing.  RIGHT ANTI emits nothing and only sets the flag.
    -> ([per batch: sorted [(left_row, block, row)]], sorted [(block, row)] of the right rows under a flag that was set)"""
    rows_of = {}
    for b, (keys, nm, jm) in enumerate(build_blocks):
        for r in range(keys.shape[0]):
            if (nm is not None and nm[r]) or (jm is not None and not jm[r]):
                continue
            rows_of.setdefault(int(keys[r]), []).append((b, r))
    used, out = set(), []
    for keys, nm in probe_key_batches:
        pairs = []
        for i in range(keys.shape[0]):
            if nm is not None and nm[i]:
                continue
            k = int(keys[i])
            if k in rows_of and k not in used:
                used.add(k)
                if not anti:
                    pairs += [(i, b, r) for b, r in rows_of[k]]
        out.append(sorted(pairs))
    return out, sorted(br for k in used for br in rows_of[k])


def non_joined_rows(build_blocks, probe_key_batches):
    """RIGHT / FULL join with strictness ALL: the build rows NotJoinedHash emits after the probe phase (HashJoin.cpp:1280-1420) are
    those whose JoinUsedFlags bit was never set (JoinUsedFlags.h; set in addFoundRowAll for every matching right row).  With ALL
    strictness every right row whose key equals some probed left key is emitted, so by definition a right row is non-joined iff its
    key is NULL, its ON mask is 0, or no probed (non-NULL) left key equals it.  Restated directly from that definition (numpy set
    membership), independently of any join implementation.
    build_blocks: [(keys, null_map or None, join_mask or None)]; probe_key_batches: [(keys, null_map or None)] -> sorted [(block, row)]"""
    seen = [k[(nm == 0) if nm is not None else slice(None)] for k, nm in probe_key_batches]
    left = np.unique(np.concatenate(seen)) if seen else np.zeros(0, dtype=np.uint64)
    out = []
    for b, (keys, nm, jm) in enumerate(build_blocks):
        inserted = np.ones(keys.shape[0], dtype=bool)
        if nm is not None:
            inserted &= nm == 0
        if jm is not None:
            inserted &= jm != 0
        matched = inserted & np.isin(keys, left)
        out += [(b, int(r)) for r in np.nonzero(~matched)[0]]
    return out
