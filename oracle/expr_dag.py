"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of the reference's expression evaluation for the DAG path
(SURVEY §8(f) rank 1).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

ExpressionActions::execute (src/Interpreters/ExpressionActions.cpp:595-747) runs the actions one by one, each function over
whole columns, materialising every intermediate: `evaluate` does exactly that, one numpy array per node.
  result types    src/DataTypes/NumberTraits.h:40-215 (Construct, ResultOfAdditionMultiplication, ResultOfSubtraction,
                  ResultOfFloatingPointDivision, ResultOfNegate, ResultOfBit, ResultOfIf)
  comparisons     src/Core/AccurateComparison.h:20-204 — mathematically exact for every operand pair, NaN compares false
                  (notEquals true).  Restated through x87 long double (64-bit mantissa: holds every Int64, UInt64 and Float64
                  exactly), an independent route from the product's integer/fraction split.
  arithmetic      FunctionBinaryArithmetic.h: static_cast<Result>(a) OP b, two's complement wrap (NO_SANITIZE_UNDEFINED)
  logical         FunctionsLogical.cpp:81,424: operands through static_cast<bool>
  dates           DateTimeTransforms.h ToYearImpl / ToMonthImpl / ToDayOfMonthImpl / ToYYYYMMImpl over DayNum — restated with
                  numpy's datetime64 calendar (independent of the product's civil-from-days arithmetic)
  intDiv, modulo  src/Functions/DivisionUtils.h:66-170 for integer operands, in the type C++'s usual arithmetic conversions choose
                  (restated on Python integers); the product compiles them only for constant divisors that cannot throw
Parity pinning: comparisons and result types are PINNED by the reference itself -- src/Core/AccurateComparison.h and
src/DataTypes/NumberTraits.h compiled in place into oracle/_ref/libchref_expr.so (ref_expr_wrapper.cpp; every type pair, adversarial
values; Int8 operands excepted in the type rules, see the wrapper) -- and by tests/golden/expr_cmp_kat.json (the reference's
00411_long_accurate_number_comparison_float answers); modulo values AND result types by 01700_mod_negative_type_promotion and 00516_modulo, intDiv by 00977_int_div
(tests/golden/expr_mod_kat.json); modulo + multiply + avg end to end by 01300_group_by_other_keys (tests/golden/sql_reference_rows.json).
Everything else here is PARITY UNPINNED by reference vectors: the calendar is checked against Python's datetime (and one 00479 row), plus / minus / multiply / divide / logical / if / bit / cast values only against their definitions
(static_cast<Result>(a) OP b) as restated here.
"""
from __future__ import annotations

import numpy as np

I64, U32, U64, F64, U8, I32, U16, I16, I8, F32 = range(10)
NP_OF = {I64: np.int64, U32: np.uint32, U64: np.uint64, F64: np.float64, U8: np.uint8, I32: np.int32, U16: np.uint16,
         I16: np.int16, I8: np.int8, F32: np.float32}
TAG_OF = {np.dtype(v): k for k, v in NP_OF.items()}
EX_INPUT, EX_CONST, EX_FUNC = 0, 1, 2
FN = {"equals": 0, "notEquals": 1, "less": 2, "greater": 3, "lessOrEquals": 4, "greaterOrEquals": 5, "plus": 10, "minus": 11,
      "multiply": 12, "divide": 13, "negate": 14, "intDiv": 15, "modulo": 16, "and": 20, "or": 21, "xor": 22, "not": 23, "if": 30, "bitAnd": 40,
      "bitOr": 41, "bitXor": 42, "toYear": 50, "toMonth": 51, "toDayOfMonth": 52, "toYYYYMM": 53, "toYYYYMMDD": 54, "toDayOfWeek": 55,
      "toQuarter": 56, "toStartOfMonth": 57}
FN_CAST = 64

assert np.finfo(np.longdouble).nmant >= 63, "this oracle needs the x87 80-bit long double"


def _size(t):
    return np.dtype(NP_OF[t]).itemsize


def _is_float(t):
    return t in (F64, F32)


def _is_signed(t):  # is_signed_v<T>: true for floats too
    return t in (I64, I32, I16, I8, F64, F32)


def _construct(sgn, flt, size):
    """NumberTraits::Construct; None = Error / a type this path does not carry"""
    if flt:
        return {8: F64, 4: F32}.get(size)
    return {(True, 1): I8, (True, 2): I16, (True, 4): I32, (True, 8): I64,
            (False, 1): U8, (False, 2): U16, (False, 4): U32, (False, 8): U64}.get((bool(sgn), size))


def _next_size(s):
    return s * 2 if s < 8 else s


def result_type(fn, a=None, b=None, c=None):
    if 0 <= fn <= 5:
        return U8
    if fn in (10, 12):
        return _construct(_is_signed(a) or _is_signed(b), _is_float(a) or _is_float(b), _next_size(max(_size(a), _size(b))))
    if fn == 11:
        return _construct(True, _is_float(a) or _is_float(b), _next_size(max(_size(a), _size(b))))
    if fn == 13:
        return F64
    if fn == 14:
        return _construct(True, _is_float(a), _size(a) if _is_signed(a) else _next_size(_size(a)))
    if fn == 15:  # ResultOfIntegerDivision (integers only on this path)
        if _is_float(a) or _is_float(b):
            return None
        return _construct(_is_signed(a) or _is_signed(b), False, _size(a))
    if fn == 16:  # ResultOfModulo
        if _is_float(a) or _is_float(b):
            return None
        return _construct(_is_signed(a), False, _next_size(_size(b)) if _is_signed(a) else _size(b))
    if fn in (20, 21, 22, 23):
        return U8
    if fn in (40, 41, 42):
        if _is_float(a) or _is_float(b):
            return None
        return _construct(_is_signed(a) or _is_signed(b), False, max(_size(a), _size(b)))
    if fn == 30:
        if _is_float(a):
            return None
        if b == c:
            return b
        has_float = _is_float(b) or _is_float(c)
        has_integer = (not _is_float(b)) or (not _is_float(c))
        has_signed = _is_signed(b) or _is_signed(c)
        has_unsigned = (not _is_signed(b)) or (not _is_signed(c))
        max_u = max(0 if _is_signed(b) else _size(b), 0 if _is_signed(c) else _size(c))
        max_s = max(_size(b) if _is_signed(b) else 0, _size(c) if _is_signed(c) else 0)
        max_i = max(0 if _is_float(b) else _size(b), 0 if _is_float(c) else _size(c))
        max_f = max(_size(b) if _is_float(b) else 0, _size(c) if _is_float(c) else 0)
        m = max(_size(b), _size(c))
        dbl = (has_float and has_integer and max_i >= max_f) or (has_signed and has_unsigned and max_u >= max_s)
        return _construct(has_signed, has_float, m * 2 if dbl else m)
    if fn == 50:
        return U16 if a == U16 else None
    if fn in (51, 52):
        return U8 if a == U16 else None
    if fn in (53, 54):
        return U32 if a == U16 else None
    if fn in (55, 56):
        return U8 if a == U16 else None
    if fn == 57:
        return U16 if a == U16 else None
    if FN_CAST <= fn < FN_CAST + 16:
        to = fn - FN_CAST
        if to not in NP_OF or (_is_float(a) and not _is_float(to)):
            return None
        return to
    return None


def _wide_u64(x):
    """value bits in 64-bit two's complement"""
    if x.dtype.kind == "i":
        return x.astype(np.int64).view(np.uint64)
    return x.astype(np.uint64)


def _exact(x):
    return x.astype(np.longdouble)


def _compare(fn, x, y):
    a, b = _exact(x), _exact(y)
    with np.errstate(invalid="ignore"):
        if fn == 0:
            r = a == b
        elif fn == 1:
            r = ~(a == b)
        elif fn == 2:
            r = a < b
        elif fn == 3:
            r = b < a
        elif fn == 4:
            r = ~np.isnan(a) & ~np.isnan(b) & ~(b < a)
        else:
            r = ~np.isnan(a) & ~np.isnan(b) & ~(a < b)
    return r.astype(np.uint8)


def _civil(days):
    d = days.astype("int64").astype("datetime64[D]")
    y = d.astype("datetime64[Y]").astype(np.int64) + 1970
    m = d.astype("datetime64[M]").astype(np.int64) % 12 + 1
    dom = (d - d.astype("datetime64[M]")).astype(np.int64) + 1
    return y, m, dom


def apply_function(fn, args, types):
    """one IFunction::executeImpl over whole columns"""
    rt = result_type(fn, *types)
    if rt is None:
        raise NotImplementedError((fn, types))
    out = NP_OF[rt]
    x = args[0]
    if 0 <= fn <= 5:
        return _compare(fn, x, args[1])
    if fn in (10, 11, 12):
        y = args[1]
        with np.errstate(over="ignore", invalid="ignore"):
            if _is_float(rt):
                a, b = x.astype(np.float64), y.astype(np.float64)
                return (a + b if fn == 10 else a - b if fn == 11 else a * b).astype(out)
            a, b = _wide_u64(x), _wide_u64(y)
            r = a + b if fn == 10 else a - b if fn == 11 else a * b
            return r.astype(np.dtype(out).str.replace("i", "u")).view(out) if np.dtype(out).kind == "i" else r.astype(out)
    if fn in (15, 16):
        return _int_div_mod(fn, x, args[1], types[0], types[1], rt)
    if fn == 13:
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            return x.astype(np.float64) / args[1].astype(np.float64)
    if fn == 14:
        if _is_float(rt):
            return (-x).astype(out)
        r = (np.uint64(0) - _wide_u64(x))
        return r.astype(np.dtype(out).str.replace("i", "u")).view(out)
    if fn in (20, 21, 22):
        a, b = x != 0, args[1] != 0
        return (a & b if fn == 20 else a | b if fn == 21 else a ^ b).astype(np.uint8)
    if fn == 23:
        return (~(x != 0)).astype(np.uint8)
    if fn in (40, 41, 42):
        a, b = _wide_u64(x), _wide_u64(args[1])
        r = a & b if fn == 40 else a | b if fn == 41 else a ^ b
        u = np.dtype(out).str.replace("i", "u")
        return r.astype(u).view(out)
    if fn == 30:
        return np.where(x != 0, _cast(args[1], rt), _cast(args[2], rt))
    if fn in (50, 51, 52, 53, 54, 56):
        y, m, d = _civil(x)
        return (y if fn == 50 else m if fn == 51 else d if fn == 52 else y * 100 + m if fn == 53 else y * 10000 + m * 100 + d if fn == 54
                else (m - 1) // 3 + 1).astype(out)
    if fn == 55:  # ToDayOfWeekImpl mode 0 (DateTimeTransforms.h): Monday = 1 ... Sunday = 7
        dd = x.astype("int64").astype("datetime64[D]")
        return ((dd.astype("datetime64[D]").view("int64") - np.datetime64("1969-12-29", "D").astype("int64")) % 7 + 1).astype(out)  # 1969-12-29 was a Monday
    if fn == 57:  # ToStartOfMonthImpl
        dd = x.astype("int64").astype("datetime64[D]")
        return dd.astype("datetime64[M]").astype("datetime64[D]").astype("int64").astype(out)
    return _cast(x, rt)


def _promote(t):
    """C++ integer promotion on LP64: everything narrower than int becomes int"""
    return I32 if _size(t) < 4 else t


def _usual_arithmetic_conversion(ta, tb):
    """the common type of `a OP b` for integer operands ([expr.arith.conv]; int = 32, long = 64 bits)"""
    ta, tb = _promote(ta), _promote(tb)
    if ta == tb:
        return ta
    if _is_signed(ta) == _is_signed(tb):
        return ta if _size(ta) >= _size(tb) else tb
    u, sg = (ta, tb) if not _is_signed(ta) else (tb, ta)
    return u if _size(u) >= _size(sg) else sg


def _to_type(v: int, t) -> int:
    """static_cast of a mathematical integer to type t (two's complement wrap)"""
    bits = 8 * _size(t)
    v &= (1 << bits) - 1
    if _is_signed(t) and v >> (bits - 1):
        v -= 1 << bits
    return v


def _int_div_mod(fn, x, y, tx, ty, rt):
    """DivideIntegralImpl / ModuloImpl for integer operands (src/Functions/DivisionUtils.h:66-170) on Python integers: exact C++
    semantics (truncation toward zero, remainder with the dividend's sign) in the type the usual arithmetic conversions choose."""
    out = np.empty(x.shape[0], dtype=NP_OF[rt])
    if fn == 15 and (_is_signed(tx) or _is_signed(ty)):
        sa = _construct(True, False, _size(tx))
        sb = _construct(True, False, _size(ty)) if _size(tx) <= _size(ty) else sa
        cast_a, cast_b = sa, sb
    else:
        cast_a, cast_b = tx, ty
    ct = _usual_arithmetic_conversion(cast_a, cast_b)
    res = []
    for a, b in zip(x.tolist(), y.tolist()):
        a = _to_type(_to_type(int(a), cast_a), ct)
        b = _to_type(_to_type(int(b), cast_b), ct)
        if b == 0:
            raise ZeroDivisionError("ILLEGAL_DIVISION")
        q = abs(a) // abs(b)
        if (a < 0) != (b < 0):
            q = -q
        r = q if fn == 15 else a - q * b
        res.append(_to_type(_to_type(r, ct), rt))
    out[:] = np.array(res, dtype=object).astype(NP_OF[rt]) if res else []
    return out


def _cast(x, to):
    """static_cast between the carried types (Float -> integer is not carried)"""
    out = np.dtype(NP_OF[to])
    if out.kind == "f":
        return x.astype(out)
    if x.dtype.kind == "f":
        raise NotImplementedError("Float -> integer cast")
    u = np.dtype(out.str.replace("i", "u"))
    return _wide_u64(x).astype(u).view(out)


def evaluate(nodes, cols):
    """nodes: (kind, code, type, args, bits) tuples in topological order; cols: list of ndarrays by INPUT position.
    Returns (values, types): one array / type tag per node."""
    n = None
    for c in cols:
        if c is not None:
            n = c.shape[0]
    vals, types = [], []
    for kind, code, typ, args, bits in nodes:
        if kind == EX_INPUT:
            v = np.ascontiguousarray(cols[code])
            assert TAG_OF[v.dtype] == typ
            vals.append(v)
            types.append(typ)
        elif kind == EX_CONST:
            one = np.frombuffer(int(bits).to_bytes(8, "little")[:_size(typ)], dtype=NP_OF[typ])
            vals.append(np.repeat(one, n))  # ColumnConst materialised
            types.append(typ)
        else:
            ar = 3 if code == 30 else 1 if (code in (14, 23) or 50 <= code <= 57 or code >= FN_CAST) else 2
            a = [vals[args[j]] for j in range(ar)]
            t = [types[args[j]] for j in range(ar)]
            vals.append(apply_function(code, a, t))
            types.append(TAG_OF[vals[-1].dtype])
    return vals, types


def filter_sum(nodes, cols, filter_node=-1, value_node=-1):
    """FilterTransform (rows where the filter column is non-zero) + sum / count without key over the surviving rows"""
    vals, types = evaluate(nodes, cols)
    n = vals[0].shape[0]
    keep = vals[filter_node] != 0 if filter_node >= 0 else np.ones(n, dtype=bool)
    cnt = int(keep.sum())
    if value_node < 0:
        return np.uint64(0), cnt
    v = vals[value_node][keep]
    if v.dtype.kind == "f":
        return np.float64(np.sum(v.astype(np.float64))), cnt
    s = np.add.reduce(_wide_u64(v), dtype=np.uint64) if v.size else np.uint64(0)
    return (np.array([s], dtype=np.uint64).view(np.int64)[0] if v.dtype.kind == "i" else np.uint64(s)), cnt


_ref_expr = None


def ref_expr():
    """The reference's own AccurateComparison.h + NumberTraits.h compiled in place (oracle/_ref/libchref_expr.so, oracle/Makefile);
    None when it was never built (no reference checkout and no prebuilt file)."""
    global _ref_expr
    if _ref_expr is None:
        import ctypes as C
        import os
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libchref_expr.so")
        if not os.path.exists(so):
            return None
        L = C.CDLL(so)
        L.ref_result_type.restype = C.c_int
        L.ref_result_type.argtypes = [C.c_int, C.c_int, C.c_int]
        L.ref_compare.restype = C.c_int
        L.ref_compare.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        _ref_expr = L
    return _ref_expr


def ref_compare(fn, x, y):
    """accurate::*Op of the compiled reference over two arrays"""
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    out = np.empty(x.shape[0], dtype=np.uint8)
    rc = ref_expr().ref_compare(fn, TAG_OF[x.dtype], TAG_OF[y.dtype], x.ctypes.data, y.ctypes.data, x.shape[0], out.ctypes.data)
    assert rc == 0
    return out
