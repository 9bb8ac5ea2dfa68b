"""ExpressionActions over numeric columns, compiled at run time into one HIP kernel.

Mirror of the reference's ActionsDAG / ExpressionActions (src/Interpreters/ActionsDAG.h, ExpressionActions.cpp:595-747):
nodes are INPUT columns, COLUMN constants and FUNCTION calls under the reference's function names; `execute` returns the
requested result columns, `filter_sum` is the fused `SELECT sum(v), count() WHERE f` step.  The work happens in
libchgpu.so (csrc/expr_jit.hip); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K
from .columns import NP_OF, TAG_OF, Column, Context, sum_result_dtype

# reference function name -> CHGPU_FN_* (include/chgpu.h)
FUNCTIONS = {
    "equals": 0, "notEquals": 1, "less": 2, "greater": 3, "lessOrEquals": 4, "greaterOrEquals": 5,
    "plus": 10, "minus": 11, "multiply": 12, "divide": 13, "negate": 14, "intDiv": 15, "modulo": 16,
    "and": 20, "or": 21, "xor": 22, "not": 23,
    "if": 30,
    "bitAnd": 40, "bitOr": 41, "bitXor": 42,
    "toYear": 50, "toMonth": 51, "toDayOfMonth": 52, "toYYYYMM": 53, "toYYYYMMDD": 54, "toDayOfWeek": 55, "toQuarter": 56,
    "toStartOfMonth": 57,
}
FN_CAST = 64
CASTS = {"toInt64": K.I64, "toUInt32": K.U32, "toUInt64": K.U64, "toFloat64": K.F64, "toUInt8": K.U8, "toInt32": K.I32,
         "toUInt16": K.U16, "toInt16": K.I16, "toInt8": K.I8, "toFloat32": K.F32}
EX_INPUT, EX_CONST, EX_FUNC = 0, 1, 2


class ExprNode(C.Structure):
    _fields_ = [("kind", C.c_int32), ("code", C.c_int32), ("type", C.c_int32), ("args", C.c_int32 * 3), ("bits", C.c_uint64)]


def function_code(name: str) -> int:
    if name in CASTS:
        return FN_CAST + CASTS[name]
    return FUNCTIONS[name]


class ActionsDAG:
    """Builder: every add_* returns the node's index (operands must exist before their users, as in ActionsDAG::addFunction)."""

    def __init__(self):
        self.nodes = []  # (kind, code, type, args, bits)

    def add_input(self, position: int, dtype) -> int:
        self.nodes.append((EX_INPUT, position, TAG_OF[np.dtype(dtype)], (-1, -1, -1), 0))
        return len(self.nodes) - 1

    def add_column(self, value, dtype) -> int:
        """a constant (ActionsDAG::addColumn of a ColumnConst)"""
        tag = TAG_OF[np.dtype(dtype)]
        bits = int.from_bytes(np.array([value], dtype=NP_OF[tag]).tobytes().ljust(8, b"\0"), "little")
        self.nodes.append((EX_CONST, 0, tag, (-1, -1, -1), bits))
        return len(self.nodes) - 1

    def add_function(self, name: str, *args: int) -> int:
        a = tuple(args) + (-1,) * (3 - len(args))
        self.nodes.append((EX_FUNC, function_code(name), 0, a, 0))
        return len(self.nodes) - 1

    def compile(self) -> "ExpressionActions":
        return ExpressionActions(self)


class ExpressionActions:
    def __init__(self, dag: ActionsDAG):
        arr = (ExprNode * len(dag.nodes))()
        for i, (kind, code, typ, args, bits) in enumerate(dag.nodes):
            arr[i].kind, arr[i].code, arr[i].type, arr[i].bits = kind, code, typ, bits
            for j in range(3):
                arr[i].args[j] = args[j]
        h = C.c_void_p()
        K.check(K.lib().chgpu_expr_compile(len(dag.nodes), arr, C.byref(h)))
        self._h = h
        self.n_nodes = len(dag.nodes)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                K.lib().chgpu_expr_free(self._h)
                self._h = None
        except Exception:
            pass

    def node_type(self, node: int) -> int:
        t = C.c_int(0)
        K.check(K.lib().chgpu_expr_node_type(self._h, node, C.byref(t)))
        return t.value

    def node_dtype(self, node: int):
        return np.dtype(NP_OF[self.node_type(node)])

    def precompile(self, out_nodes=(), filter_node: int = -1, value_node: int = -1) -> int:
        """hiprtc only (no device): bytes of the gfx950 code object"""
        outs = (C.c_uint32 * max(1, len(out_nodes)))(*out_nodes)
        nbytes = C.c_uint64(0)
        K.check(K.lib().chgpu_expr_precompile(self._h, len(out_nodes), outs, filter_node, value_node, C.byref(nbytes)))
        return nbytes.value

    def execute(self, ctx: Context, cols, out_nodes):
        """cols[position] for every INPUT node (None for unused positions) -> one new Column per out_nodes entry"""
        arr = (C.c_void_p * len(cols))(*[c._h if c is not None else None for c in cols])
        outs_n = (C.c_uint32 * len(out_nodes))(*out_nodes)
        outs = (C.c_void_p * len(out_nodes))()
        K.check(K.lib().chgpu_expr_execute(ctx._h, self._h, len(cols), arr, len(out_nodes), outs_n, outs))
        return [Column(ctx, C.c_void_p(h)) for h in outs]

    def filter_execute(self, ctx: Context, cols, filter_node: int, out_nodes):
        """WHERE filter_node + projection of out_nodes in one step -> ([Columns of the surviving rows, in order], rows)"""
        arr = (C.c_void_p * len(cols))(*[c._h if c is not None else None for c in cols])
        outs_n = (C.c_uint32 * len(out_nodes))(*out_nodes)
        outs = (C.c_void_p * len(out_nodes))()
        rows = C.c_uint64(0)
        K.check(K.lib().chgpu_expr_filter_execute(ctx._h, self._h, len(cols), arr, filter_node, len(out_nodes), outs_n, outs, C.byref(rows)))
        return [Column(ctx, C.c_void_p(h)) for h in outs], int(rows.value)

    def filter_minmax(self, ctx: Context, cols, filter_node: int, value_node: int):
        """(min(value_node), max(value_node), count()) over the rows where filter_node != 0 (-1: every row); integer values only"""
        arr = (C.c_void_p * len(cols))(*[c._h if c is not None else None for c in cols])
        vt = C.c_int(0)
        lo, hi = np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.uint64)
        cnt = C.c_uint64(0)
        K.check(K.lib().chgpu_expr_filter_minmax_node(ctx._h, self._h, len(cols), arr, filter_node, value_node, C.byref(vt),
                                                      lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), C.byref(cnt)))
        dt = np.dtype(NP_OF[vt.value])
        return lo.view(dt)[0], hi.view(dt)[0], int(cnt.value)

    def filter_sum(self, ctx: Context, cols, filter_node: int = -1, value_node: int = -1):
        """(sum(value_node), count()) over the rows where filter_node != 0, one pass"""
        arr = (C.c_void_p * len(cols))(*[c._h if c is not None else None for c in cols])
        rt = C.c_int(0)
        out = np.zeros(1, dtype=np.uint64)
        cnt = C.c_uint64(0)
        K.check(K.lib().chgpu_expr_filter_sum_node(ctx._h, self._h, len(cols), arr, filter_node, value_node, C.byref(rt),
                                                   out.ctypes.data_as(C.c_void_p), C.byref(cnt)))
        return out.view(sum_result_dtype(self.node_type(value_node)) if value_node >= 0 else np.uint64)[0], int(cnt.value)
