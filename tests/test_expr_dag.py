"""SURVEY §8(f) rank 1, general form: an expression DAG compiled at run time into one HIP kernel.

CPU (not gpu): the numpy oracle against the reference's accurate-comparison known answers
(tests/golden/expr_cmp_kat.json <- 00411_long_accurate_number_comparison_float) and an independent calendar; the product's
type inference against the oracle's for every function and operand-type combination; the run-time compiler producing a
gfx950 code object without a device.
GPU: the same known answers through the JIT kernel, a seeded differential fuzz of random DAGs over adversarial columns
(bit-exact, Float64 included), ragged / unaligned / empty inputs, the fused filter + sum, SSB Q1.1 against the hand-written
fused kernel.
"""
import datetime
import json
import os

import numpy as np
import pytest

from oracle import expr_dag as OE

HERE = os.path.dirname(os.path.abspath(__file__))
NP_NAME = {"UInt8": np.uint8, "Int8": np.int8, "UInt16": np.uint16, "Int16": np.int16, "UInt32": np.uint32, "Int32": np.int32,
           "UInt64": np.uint64, "Int64": np.int64}
CMP = ["equals", "notEquals", "less", "lessOrEquals", "greater", "greaterOrEquals"]  # the fixture's order: = != < <= > >=
ALL_TYPES = [np.int64, np.uint32, np.uint64, np.float64, np.uint8, np.int32, np.uint16, np.int16, np.int8, np.float32]


def _cmp_kat():
    with open(os.path.join(HERE, "golden", "expr_cmp_kat.json")) as f:
        return json.load(f)["cases"]


def _literal_dtype(v: int):
    """type of an integer literal: the narrowest, unsigned unless negative (src/Parsers + FieldToDataType.cpp)"""
    if v < 0:
        return np.int8 if v >= -2**7 else np.int16 if v >= -2**15 else np.int32 if v >= -2**31 else np.int64
    return np.uint8 if v < 2**8 else np.uint16 if v < 2**16 else np.uint32 if v < 2**32 else np.uint64


def _kat_by_type():
    """{numpy int dtype: (ints, floats, answers[12][n])}"""
    per = {}
    for c in _cmp_kat():
        v, f = int(c["int"]), float(c["float"])
        for tname, ans in zip(c["types"], c["answers"]):
            dt = np.dtype(_literal_dtype(v) if tname == "literal" else NP_NAME[tname])
            per.setdefault(dt, ([], [], []))
            per[dt][0].append(v)
            per[dt][1].append(f)
            per[dt][2].append(ans)
    return {dt: (np.array(i, dtype=dt), np.array(f, dtype=np.float64), np.array(a, dtype=np.uint8).T) for dt, (i, f, a) in per.items()}


def _cmp_nodes(int_dtype):
    """12 comparison nodes: i op f (6) then f op i (6)"""
    nodes = [(OE.EX_INPUT, 0, OE.TAG_OF[np.dtype(int_dtype)], (-1, -1, -1), 0), (OE.EX_INPUT, 1, OE.F64, (-1, -1, -1), 0)]
    for a, b in ((0, 1), (1, 0)):
        for name in CMP:
            nodes.append((OE.EX_FUNC, OE.FN[name], 0, (a, b, -1), 0))
    return nodes


def test_oracle_accurate_comparison_known_answers():
    n_checked = 0
    for dt, (ints, floats, answers) in _kat_by_type().items():
        vals, _ = OE.evaluate(_cmp_nodes(dt), [ints, floats])
        for k in range(12):
            assert vals[2 + k].tolist() == answers[k].tolist(), (dt, k)
            n_checked += ints.shape[0]
    assert n_checked > 5000


def test_oracle_calendar_against_python_datetime():
    days = np.array([0, 1, 58, 59, 60, 365, 366, 789, 8400, 8401, 8765, 8766, 11016, 11017, 19782, 47540, 47541, 65535], dtype=np.uint16)
    nodes = [(OE.EX_INPUT, 0, OE.U16, (-1, -1, -1), 0)] + [(OE.EX_FUNC, OE.FN[f], 0, (0, -1, -1), 0) for f in
             ("toYear", "toMonth", "toDayOfMonth", "toYYYYMM", "toYYYYMMDD", "toDayOfWeek", "toQuarter", "toStartOfMonth")]
    vals, types = OE.evaluate(nodes, [days])
    assert types[1:] == [OE.U16, OE.U8, OE.U8, OE.U32, OE.U32, OE.U8, OE.U8, OE.U16]
    for i, d in enumerate(days.tolist()):
        c = datetime.date(1970, 1, 1) + datetime.timedelta(days=d)
        assert int(vals[5][i]) == c.year * 10000 + c.month * 100 + c.day and int(vals[6][i]) == c.isoweekday()
        assert int(vals[7][i]) == (c.month - 1) // 3 + 1 and int(vals[8][i]) == (c.replace(day=1) - datetime.date(1970, 1, 1)).days
    for i, d in enumerate(days.tolist()):
        c = datetime.date(1970, 1, 1) + datetime.timedelta(days=d)
        assert (int(vals[1][i]), int(vals[2][i]), int(vals[3][i]), int(vals[4][i])) == (c.year, c.month, c.day, c.year * 100 + c.month)
    # the reference's own answer: 00479_date_and_datetime_to_number, toYYYYMM(toDate('2017-07-21')) = 201707
    with open(os.path.join(HERE, "golden", "expr_mod_kat.json")) as f:
        r479 = json.load(f)["00479_toYYYYMM_of_date_2017_07_21"]["rows"]
    d0 = np.array([(datetime.date(2017, 7, 21) - datetime.date(1970, 1, 1)).days], dtype=np.uint16)
    v0 = OE.evaluate(nodes, [d0])[0]
    assert int(v0[4][0]) == int(r479[0][0]) == 201707 and int(v0[5][0]) == int(r479[1][0]) == 20170721


def test_oracle_result_types_documented_examples():
    # NumberTraits.h:66-70 "UInt8 + Int32 = Int64"; :159-168 the if() table
    assert OE.result_type(OE.FN["plus"], OE.U8, OE.I32) == OE.I64
    assert OE.result_type(OE.FN["plus"], OE.U8, OE.U8) == OE.U16
    assert OE.result_type(OE.FN["minus"], OE.U8, OE.U8) == OE.I16
    assert OE.result_type(OE.FN["multiply"], OE.U64, OE.I8) == OE.I64
    assert OE.result_type(OE.FN["plus"], OE.F32, OE.F32) == OE.F64
    assert OE.result_type(OE.FN["divide"], OE.U8, OE.U8) == OE.F64
    assert OE.result_type(OE.FN["negate"], OE.U8) == OE.I16 and OE.result_type(OE.FN["negate"], OE.I64) == OE.I64
    assert OE.result_type(OE.FN["if"], OE.U8, OE.U8, OE.U16) == OE.U16       # UInt<x>, UInt<y> -> UInt<max>
    assert OE.result_type(OE.FN["if"], OE.U8, OE.U16, OE.I16) == OE.I32      # UInt<x>, Int<y> -> Int<max(x*2, y)>
    assert OE.result_type(OE.FN["if"], OE.U8, OE.F32, OE.I32) == OE.F64      # Float<x>, Int<y> -> Float<max(x, y*2)>
    assert OE.result_type(OE.FN["if"], OE.U8, OE.U64, OE.I8) is None         # UInt64, Int<x> -> Error
    assert OE.result_type(OE.FN["if"], OE.U8, OE.F64, OE.I64) is None        # Float<x>, [U]Int64 -> Error
    assert OE.result_type(OE.FN["bitAnd"], OE.U8, OE.I32) == OE.I32


def _all_function_cases():
    tags = list(range(10))
    for name, fn in OE.FN.items():
        ar = 3 if name == "if" else 1 if name in ("negate", "not") or name.startswith("to") else 2
        if ar == 1:
            for a in tags:
                yield fn, (a,)
        elif ar == 2:
            for a in tags:
                for b in tags:
                    yield fn, (a, b)
        else:
            for b in tags:
                for c in tags:
                    yield fn, (OE.U8, b, c)
    for to in tags:
        for a in tags:
            yield OE.FN_CAST + to, (a,)


def test_product_type_inference_equals_oracle_for_every_combination():
    """chgpu_expr_compile needs no device: its result types and NOT_IMPLEMENTED answers against the oracle's"""
    import clickhouse_amd as ch
    from clickhouse_amd.columns import NP_OF
    n = 0
    for fn, types in _all_function_cases():
        d = ch.ActionsDAG()
        ins = [d.add_input(j, NP_OF[t]) for j, t in enumerate(types)]
        if fn in (OE.FN["intDiv"], OE.FN["modulo"]) and types[1] not in (OE.F64, OE.F32):
            ins[1] = d.add_column(3, NP_OF[types[1]])  # compiled only for constant divisors that cannot throw
        d.nodes.append((OE.EX_FUNC, fn, 0, tuple(ins) + (-1,) * (3 - len(ins)), 0))
        want = OE.result_type(fn, *types)
        if want is None:
            with pytest.raises(ch.ChgpuError) as ei:
                d.compile()
            assert ei.value.code == ch._capi.ERR_NOT_IMPLEMENTED, (fn, types)
        else:
            assert d.compile().node_type(len(d.nodes) - 1) == want, (fn, types)
        n += 1
    assert n > 1500


def test_compile_errors():
    import clickhouse_amd as ch
    d = ch.ActionsDAG()
    d.nodes.append((OE.EX_FUNC, OE.FN["plus"], 0, (1, 2, -1), 0))  # operands that do not precede their user
    with pytest.raises(ch.ChgpuError) as ei:
        d.compile()
    assert ei.value.code == ch._capi.ERR_BAD_ARGUMENTS
    for build in (lambda d, a: d.add_function("intDiv", a, d.add_input(1, np.int64)),          # a column divisor can be zero
                  lambda d, a: d.add_function("modulo", a, d.add_column(0, np.uint8)),         # division by zero
                  lambda d, a: d.add_function("intDiv", a, d.add_column(-1, np.int64)),        # min / -1
                  lambda d, a: d.add_function("intDiv", a, d.add_column(2**64 - 1, np.uint64))):  # -1 once cast to Int64
        d = ch.ActionsDAG()
        build(d, d.add_input(0, np.int64))
        with pytest.raises(ch.ChgpuError) as ei:
            d.compile()
        assert ei.value.code == ch._capi.ERR_NOT_IMPLEMENTED
    d = ch.ActionsDAG()
    a = d.add_input(0, np.float64)
    d.add_function("toInt64", a)  # Float -> integer: not carried
    with pytest.raises(ch.ChgpuError) as ei:
        d.compile()
    assert ei.value.code == ch._capi.ERR_NOT_IMPLEMENTED


def _q11_dag(ch):
    d = ch.ActionsDAG()
    od, disc, qty, price = (d.add_input(j, np.uint32) for j in range(4))
    c = lambda v: d.add_column(v, np.uint32)
    f = d.add_function("and", d.add_function("greaterOrEquals", od, c(19930101)), d.add_function("lessOrEquals", od, c(19931231)))
    f = d.add_function("and", f, d.add_function("greaterOrEquals", disc, c(1)))
    f = d.add_function("and", f, d.add_function("lessOrEquals", disc, c(3)))
    f = d.add_function("and", f, d.add_function("less", qty, c(25)))
    v = d.add_function("multiply", price, disc)
    return d, f, v


def test_runtime_compiler_builds_gfx950_code_without_a_device():
    import clickhouse_amd as ch
    d, f, v = _q11_dag(ch)
    ex = d.compile()
    assert ex.node_dtype(v) == np.uint64 and ex.node_dtype(f) == np.uint8
    assert ex.precompile(filter_node=f, value_node=v) > 1000       # fused filter + sum kernel
    assert ex.precompile(out_nodes=[f, v]) > 1000                  # materialising kernel


# ------------------------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------------------------
SPECIAL_F64 = [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.5, np.nan, np.inf, -np.inf, 2.0**63, -(2.0**63), 2.0**64, 2.0**63 + 2048, 2.0**53, 2.0**53 + 2,
               9007199254740993.0, 4294967296.0, 4294967295.5, -2147483648.5, 1e-300, 1e300, 255.0, 256.0, 32767.0, -32769.0]


def _random_column(rng, dtype, n):
    dt = np.dtype(dtype)
    if dt.kind == "f":
        x = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 19, size=n)
        idx = rng.integers(0, n, size=max(1, n // 4))
        x[idx] = rng.choice(np.array(SPECIAL_F64), size=idx.shape[0])
        idx = rng.integers(0, n, size=max(1, n // 4))
        x[idx] = np.round(x[idx])  # integral values: the equal-to-an-integer branches
        return x.astype(dt)
    info = np.iinfo(dt)
    x = rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True)
    idx = rng.integers(0, n, size=max(1, n // 3))
    small = rng.integers(-3, 4, size=idx.shape[0])
    x[idx] = np.clip(small, info.min, info.max).astype(dt)
    idx = rng.integers(0, n, size=max(1, n // 8))
    x[idx] = rng.choice(np.array([info.min, info.max, info.max - 1, info.min + 1 if info.min < 0 else 1], dtype=dt), size=idx.shape[0])
    return x


def _random_dag(ch, rng, col_dtypes, n_funcs):
    from clickhouse_amd.columns import TAG_OF
    d = ch.ActionsDAG()
    types = []
    for j, dt in enumerate(col_dtypes):
        d.add_input(j, dt)
        types.append(TAG_OF[np.dtype(dt)])
    consts = {}
    for _ in range(3):
        dt = ALL_TYPES[rng.integers(0, len(ALL_TYPES))]
        val = rng.choice(np.array(SPECIAL_F64)) if np.dtype(dt).kind == "f" else int(rng.integers(-5, 300))
        if np.dtype(dt).kind == "u":
            val = abs(int(val))
        if np.dtype(dt).kind != "f":
            val = int(np.clip(val, np.iinfo(dt).min, np.iinfo(dt).max))
        consts[d.add_column(val, dt)] = val
        types.append(TAG_OF[np.dtype(dt)])
    names = list(OE.FN.keys()) + ["cast"]
    made = 0
    while made < n_funcs:
        name = names[rng.integers(0, len(names))]
        fn = OE.FN_CAST + int(rng.integers(0, 10)) if name == "cast" else OE.FN[name]
        ar = 3 if name == "if" else 1 if name in ("negate", "not", "cast") or name.startswith("to") else 2
        args = [int(rng.integers(0, len(types))) for _ in range(ar)]
        if name in ("intDiv", "modulo"):  # only constant divisors that cannot throw are compiled
            ok = [c for c, v in consts.items() if types[c] not in (OE.F64, OE.F32) and v not in (0, -1) and v != np.iinfo(OE.NP_OF[types[c]]).max]
            if not ok or types[args[0]] in (OE.F64, OE.F32):
                continue
            args[1] = ok[int(rng.integers(0, len(ok)))]
        at = [types[a] for a in args]
        if name == "if" and at[0] in (OE.F64, OE.F32):
            continue
        rt = OE.result_type(fn, *at)
        if rt is None:
            continue
        d.nodes.append((OE.EX_FUNC, fn, 0, tuple(args) + (-1,) * (3 - ar), 0))
        types.append(rt)
        made += 1
    return d, types


def _same(a, b):
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.dtype.kind == "f":
        return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))) and np.all(np.signbit(a[~np.isnan(a)]) == np.signbit(b[~np.isnan(b)])))
    return bool(np.array_equal(a, b))


@pytest.mark.gpu
def test_gpu_accurate_comparison_known_answers():
    import clickhouse_amd as ch
    ctx = ch.Context()
    for dt, (ints, floats, answers) in _kat_by_type().items():
        d = ch.ActionsDAG()
        d.nodes = _cmp_nodes(dt)
        ex = d.compile()
        cols = [ctx.upload(ints), ctx.upload(floats)]
        outs = ex.execute(ctx, cols, list(range(2, 8))) + ex.execute(ctx, cols, list(range(8, 14)))  # <= 8 outputs per call
        for k in range(12):
            assert outs[k].numpy().tolist() == answers[k].tolist(), (dt, k)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_random_dags_match_oracle_bit_exact(seed):
    import clickhouse_amd as ch
    seed += int(os.environ.get("CHGPU_FUZZ_SEED", "0")) * 1000
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    ctx = ch.Context()
    n = [0, 1, 63, 1023, 4096 * 4 + 3, 100_003, 1_000_001][seed % 7]
    n_cols = int(rng.integers(1, 5))
    dts = [ALL_TYPES[rng.integers(0, len(ALL_TYPES))] for _ in range(n_cols)]
    if seed % 3 == 0:
        dts[0] = np.uint16  # a Date column: the calendar functions get picked
    host = [_random_column(rng, dt, n + 1) for dt in dts]
    d, types = _random_dag(ch, rng, dts, n_funcs=int(rng.integers(4, 20)))
    ex = d.compile()
    unaligned = seed % 2 == 1  # views starting at row 1: the one-row-per-lane kernel
    cols_h = [h[1:] if unaligned else h[:n] for h in host]
    up = [ctx.upload(h) for h in host]
    cols_d = [c.cut(1, n) if unaligned else c.cut(0, n) for c in up]
    if n == 0:
        return  # an empty chunk never reaches the actions (ISimpleTransform skips it)
    vals, otypes = OE.evaluate(d.nodes, cols_h)
    assert otypes == types
    fnodes = list(range(len(dts) + 3, len(d.nodes)))
    for lo in range(0, len(fnodes), 8):
        part = fnodes[lo:lo + 8]
        outs = ex.execute(ctx, cols_d, part)
        for k, o in zip(part, outs):
            assert ex.node_type(k) == types[k]
            got = o.numpy()
            assert _same(got, vals[k]), (seed, k, d.nodes[k], got[:8], vals[k][:8])
    # WHERE + projection in one step: the surviving rows of up to 7 nodes, in order
    ints0 = [k for k in range(len(d.nodes)) if types[k] not in (OE.F64, OE.F32)]
    if ints0:
        fn0 = ints0[int(rng.integers(0, len(ints0)))]
        outs_k = [int(x) for x in rng.choice(len(d.nodes), size=min(len(d.nodes), 1 + int(rng.integers(0, 7))), replace=False)]
        got, nrows = ex.filter_execute(ctx, cols_d, fn0, outs_k)
        keep = vals[fn0] != 0
        assert nrows == int(keep.sum())
        for k, o in zip(outs_k, got):
            assert _same(o.numpy(), vals[k][keep]), (seed, "filter_execute", fn0, k)
    # fused WHERE + min / max / count over integer nodes
    if ints0:
        vn = ints0[int(rng.integers(0, len(ints0)))]
        fn1 = ints0[int(rng.integers(0, len(ints0)))]
        lo, hi, cnt = ex.filter_minmax(ctx, cols_d, fn1, vn)
        keep1 = vals[fn1] != 0
        sel = vals[vn][keep1]
        assert cnt == int(keep1.sum())
        if cnt:
            assert lo.dtype == sel.dtype and int(lo) == int(sel.min()) and int(hi) == int(sel.max()), (seed, "minmax", fn1, vn)
        else:
            assert int(lo) == 0 and int(hi) == 0
    # fused WHERE + sum + count over a random (filter, value) pair
    ints = [k for k in fnodes if types[k] not in (OE.F64, OE.F32)]
    if ints:
        fnode = ints[int(rng.integers(0, len(ints)))]
        vnode = fnodes[int(rng.integers(0, len(fnodes)))]
        s, c = ex.filter_sum(ctx, cols_d, fnode, vnode)
        es, ec = OE.filter_sum(d.nodes, cols_h, fnode, vnode)
        assert c == ec
        if types[vnode] in (OE.F64, OE.F32):
            if np.isfinite(es):
                absum = float(np.sum(np.abs(vals[vnode][vals[fnode] != 0].astype(np.float64))))
                assert abs(float(s) - float(es)) <= 1e-6 * max(absum, 1e-300)  # BASELINE: 1e-6 relative for sum(Float64)
        else:
            assert s.dtype == es.dtype and int(s) == int(es)


@pytest.mark.gpu
def test_gpu_ssb_q11_dag_equals_handwritten_fused_kernel_and_oracle(oracle_mod):
    import clickhouse_amd as ch
    from test_expr import _q11_columns, _q11_preds
    ctx = ch.Context()
    host = _q11_columns(3_000_017, seed=5)
    cols = [ctx.upload(h) for h in host]
    d, f, v = _q11_dag(ch)
    ex = d.compile()
    s, c = ex.filter_sum(ctx, cols, f, v)
    s2, c2 = ch.expr_filter_sum(cols, _q11_preds(ch), ch.VAL_MUL, 3, 1)
    es, ec = oracle_mod.expr_filter_sum_pipeline(list(host), _q11_preds(oracle_mod), oracle_mod.VAL_MUL, 3, 1)
    assert (int(s), c) == (int(s2), c2) == (int(es), ec) and s.dtype == np.uint64
    # count only, no WHERE
    s0, c0 = ex.filter_sum(ctx, cols, -1, -1)
    assert c0 == host[0].shape[0]
    # the same query with toYear over a Date column instead of the integer date range
    rng = np.random.Generator(np.random.PCG64(11))
    days = rng.integers(8000, 9500, size=host[0].shape[0]).astype(np.uint16)
    d2 = ch.ActionsDAG()
    dd, disc, qty, price = d2.add_input(0, np.uint16), d2.add_input(1, np.uint32), d2.add_input(2, np.uint32), d2.add_input(3, np.uint32)
    f2 = d2.add_function("equals", d2.add_function("toYear", dd), d2.add_column(1993, np.uint16))
    f2 = d2.add_function("and", f2, d2.add_function("less", qty, d2.add_column(25, np.uint8)))
    v2 = d2.add_function("multiply", price, disc)
    s3, c3 = d2.compile().filter_sum(ctx, [ctx.upload(days)] + cols[1:], f2, v2)
    m = (days >= 8401) & (days <= 8765) & (host[2] < 25)  # 1993-01-01 .. 1993-12-31
    assert c3 == int(m.sum()) and int(s3) == int((host[3][m].astype(np.uint64) * host[1][m]).sum())


@pytest.mark.gpu
def test_gpu_dag_size_mismatch_and_type_mismatch_are_errors():
    import clickhouse_amd as ch
    ctx = ch.Context()
    d = ch.ActionsDAG()
    a, b = d.add_input(0, np.int64), d.add_input(1, np.int64)
    p = d.add_function("plus", a, b)
    ex = d.compile()
    x, y = ctx.upload(np.arange(10, dtype=np.int64)), ctx.upload(np.arange(11, dtype=np.int64))
    with pytest.raises(ch.ChgpuError) as ei:
        ex.execute(ctx, [x, y], [p])
    assert ei.value.code == ch._capi.ERR_SIZES_MISMATCH
    with pytest.raises(ch.ChgpuError) as ei:
        ex.execute(ctx, [x, ctx.upload(np.arange(10, dtype=np.int32))], [p])
    assert ei.value.code == ch._capi.ERR_BAD_ARGUMENTS


@pytest.mark.gpu
def test_gpu_dag_linearity_at_2_pow_27_rows():
    """size-independent property at a size the oracle does not finish quickly: sum over ragged splits adds up, and
    count(p) + count(not p) = n"""
    import clickhouse_amd as ch
    import torch
    ctx = ch.Context()
    n = 1 << 27
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randint(-2**31, 2**31, (n,), dtype=torch.int64, device="cuda", generator=g)
    b = torch.randint(0, 1000, (n,), dtype=torch.int32, device="cuda", generator=g)
    ca, cb = ctx.wrap(a.data_ptr(), np.int64, n, a), ctx.wrap(b.data_ptr(), np.int32, n, b)
    d = ch.ActionsDAG()
    ia, ib = d.add_input(0, np.int64), d.add_input(1, np.int32)
    p = d.add_function("less", ib, d.add_column(100, np.uint8))
    q = d.add_function("not", p)
    v = d.add_function("multiply", ia, ib)
    ex = d.compile()
    s_all, c_all = ex.filter_sum(ctx, [ca, cb], p, v)
    _, c_not = ex.filter_sum(ctx, [ca, cb], q, v)
    assert c_all + c_not == n
    cuts = [0, 1, 12345, n // 3 + 7, n - 1, n]
    tot, cnt = 0, 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        s, c = ex.filter_sum(ctx, [ca.cut(lo, hi - lo), cb.cut(lo, hi - lo)], p, v)
        tot = (tot + int(s)) & (2**64 - 1)
        cnt += c
    assert cnt == c_all and tot == int(s_all) & (2**64 - 1)
    ref = int((a * b.to(torch.int64))[b < 100].sum().item()) & (2**64 - 1)
    assert ref == int(s_all) & (2**64 - 1)


@pytest.mark.gpu
def test_gpu_intdiv_modulo_by_constants_every_integer_type_pair():
    """DivideIntegralImpl / ModuloImpl (DivisionUtils.h:66-170) for every (dividend type, divisor type) and a few constant divisors,
    negative ones included: the kernel divides in the type C++'s usual arithmetic conversions give on the host"""
    import clickhouse_amd as ch
    ctx = ch.Context()
    rng = np.random.Generator(np.random.PCG64(15))
    ints = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64]
    n = 3001
    for ta in ints:
        x = _random_column(rng, ta, n)
        col = ctx.upload(x)
        for tb in ints:
            info = np.iinfo(tb)
            divs = [1, 2, 3, 7, 100, int(info.max) - 1] + ([-2, -3, int(info.min)] if info.min < 0 else [])
            d = ch.ActionsDAG()
            a = d.add_input(0, ta)
            outs = []
            for v in divs:
                c = d.add_column(v, tb)
                outs.append(d.add_function("intDiv", a, c))
                if v != int(info.min) or info.min == 0:
                    outs.append(d.add_function("modulo", a, c))
            ex = d.compile()
            if info.min < 0:
                # modulo by the most negative value of a signed divisor type: the reference's constant-divisor path throws ILLEGAL_DIVISION
                # "Division by the most negative number" (src/Functions/modulo.cpp:56-80) -> not compiled, the caller keeps its CPU function
                bad = ch.ActionsDAG()
                bad.add_function("modulo", bad.add_input(0, ta), bad.add_column(int(info.min), tb))
                with pytest.raises(ch.ChgpuError) as ei:
                    bad.compile()
                assert ei.value.code == ch._capi.ERR_NOT_IMPLEMENTED
            vals, types = OE.evaluate(d.nodes, [x])
            for lo in range(0, len(outs), 8):
                part = outs[lo:lo + 8]
                for k, o in zip(part, ex.execute(ctx, [col], part)):
                    assert ex.node_type(k) == types[k]
                    assert _same(o.numpy(), vals[k]), (ta, tb, d.nodes[k], d.nodes[d.nodes[k][3][1]])


@pytest.mark.gpu
def test_gpu_01300_group_by_modulo_expression_reference_rows(golden):
    """tests/queries/0_stateless/01300_group_by_other_keys: SELECT round(avg(log(2) * number), 6) FROM numbers(1e7) GROUP BY number % 5
    with BOTH expressions computed on the device by the run-time compiled DAG (key = modulo(number, 5) -> UInt8, the reference's key
    type; value = multiply(Float64 constant, number)), then the GROUP BY; expected rows are the reference's .reference lines."""
    import clickhouse_amd as ch
    ctx = ch.Context()
    want = sorted(float(r[0]) for r in golden["rows"]["01300_avg_group_by_mod5"]["rows"])
    d = ch.ActionsDAG()
    num = d.add_input(0, np.uint64)
    key = d.add_function("modulo", num, d.add_column(5, np.uint8))
    val = d.add_function("multiply", d.add_column(np.log(2.0), np.float64), num)
    ex = d.compile()
    assert ex.node_dtype(key) == np.uint8 and ex.node_dtype(val) == np.float64
    agg = ch.Aggregator(np.uint8, [(ch.AGG_AVG, np.float64)], ctx=ctx)
    rows, stripe = 10_000_000, 2_500_000
    for lo in range(0, rows, stripe):
        k, v = ex.execute(ctx, [ctx.upload(np.arange(lo, lo + stripe, dtype=np.uint64))], [key, val])
        agg.execute_on_block(k, [v])
    keys, (avg,) = agg.convert_to_block()
    assert sorted(keys.tolist()) == [0, 1, 2, 3, 4]
    got = sorted(float(x) for x in avg)
    for g, w in zip(got, want):
        assert abs(g - w) <= 1e-6 * abs(w)  # BASELINE: 1e-6 relative for avg(Float64)
    assert sum(round(g, 6) == w for g, w in zip(got, want)) >= 3


# the operands of the reference's modulo tests, restated: (dividend, its type, divisor, its type); integer literals take the narrowest
# type, unsigned unless negative (_literal_dtype)
_MOD_01700 = [(-199, np.int32, 200, np.uint8), (-199, np.int32, 200, np.uint16), (-199, np.int32, 200, np.uint32), (-199, np.int32, 200, np.uint64),
              (-199, np.int32, -200, np.int16), (199, np.uint8, -10, np.int8), (199, np.uint8, -200, np.int16)]
_MOD_00516 = [(1000, 32), (7, 3), (255, 510), (255, 512), (255, 1000000009), (0, 255), (2147483647, 255), (-1, -1), (-1, -2), (255, 99), (42, 13),
              (42, 22), (1234567, 123)]
_TYPE_NAME = {"Int8": np.int8, "Int16": np.int16, "Int32": np.int32, "Int64": np.int64, "UInt8": np.uint8, "UInt16": np.uint16, "UInt32": np.uint32,
              "UInt64": np.uint64}


def _mod_cases():
    with open(os.path.join(HERE, "golden", "expr_mod_kat.json")) as f:
        kat = json.load(f)
    cases = []
    for (a, ta, b, tb), row in zip(_MOD_01700, kat["01700_mod_negative_type_promotion"]["rows"]):
        cases.append((a, ta, b, tb, int(row[0]), _TYPE_NAME[row[1]]))
    for (a, b), row in zip(_MOD_00516, kat["00516_modulo"]["rows"]):
        cases.append((a, _literal_dtype(a), b, _literal_dtype(b), int(row[0]), None))
    assert len(cases) == 20
    return cases


def test_oracle_modulo_reference_rows():
    """01700_mod_negative_type_promotion (values AND result types, e.g. toInt32(-199) % toUInt32(200) = 97 :: Int64 -- the unsigned
    remainder of the usual arithmetic conversions) and 00516_modulo"""
    for a, ta, b, tb, want, want_type in _mod_cases():
        r = OE.apply_function(OE.FN["modulo"], [np.array([a], dtype=ta), np.array([b], dtype=tb)], [OE.TAG_OF[np.dtype(ta)], OE.TAG_OF[np.dtype(tb)]])
        assert int(r[0]) == want, (a, ta, b, tb)
        if want_type is not None:
            assert r.dtype == np.dtype(want_type), (a, ta, b, tb, r.dtype)


@pytest.mark.gpu
def test_gpu_modulo_reference_rows():
    import clickhouse_amd as ch
    ctx = ch.Context()
    done = 0
    for a, ta, b, tb, want, want_type in _mod_cases():
        d = ch.ActionsDAG()
        x = d.add_input(0, ta)
        try:
            m = d.add_function("modulo", x, d.add_column(b, tb))
            ex = d.compile()
        except ch.ChgpuError as e:  # a divisor of -1 could raise ILLEGAL_DIVISION for the minimal dividend: left to the CPU
            assert e.code == ch._capi.ERR_NOT_IMPLEMENTED and b == -1
            continue
        out = ex.execute(ctx, [ctx.upload(np.array([a, a, a], dtype=ta))], [m])[0].numpy()
        assert out.tolist() == [want] * 3 and (want_type is None or out.dtype == np.dtype(want_type)), (a, ta, b, tb, out)
        done += 1
    assert done == 19


def test_oracle_intdiv_reference_rows():
    """00977_int_div: intDiv(-1, number), intDiv(toInt32(number), -1), intDiv(toInt64(number), -1), intDiv(number, -number) over
    numbers(1, 10) -- column divisors and -1 (forms the device path leaves to the CPU; they pin the restated DivideIntegralImpl)"""
    with open(os.path.join(HERE, "golden", "expr_mod_kat.json")) as f:
        rows = json.load(f)["00977_int_div"]["rows"]
    number = np.arange(1, 11, dtype=np.uint64)
    neg = OE.apply_function(OE.FN["negate"], [number], [OE.U64])  # -number: Int64
    forms = [(np.full(10, -1, dtype=np.int8), number), (number.astype(np.int32), np.full(10, -1, dtype=np.int8)),
             (number.astype(np.int64), np.full(10, -1, dtype=np.int8)), (number, neg)]
    for (a, b), want in zip(forms, rows):
        r = OE.apply_function(OE.FN["intDiv"], [a, b], [OE.TAG_OF[a.dtype], OE.TAG_OF[b.dtype]])
        assert r.tolist() == [int(w[0]) for w in want], (a.dtype, b.dtype)


def test_result_types_and_comparisons_against_the_compiled_reference():
    """oracle/_ref/libchref_expr.so = the reference's own NumberTraits.h and AccurateComparison.h compiled in place: every result type
    of plus / minus / multiply / divide / negate / intDiv / modulo / bit* / if over all 100 operand-type pairs -- oracle AND product --
    and all six comparisons over adversarial values of every type pair"""
    R = OE.ref_expr()
    if R is None:
        pytest.skip("oracle/_ref/libchref_expr.so was never built (no reference checkout)")
    import clickhouse_amd as ch
    from clickhouse_amd.columns import NP_OF
    tags = list(range(10))
    checked = 0
    # the reference's Int8 is `signed _BitInt(8)`; its type traits rest on std::is_signed_v, which this image's libstdc++ answers false for
    # _BitInt (the reference builds against libc++) -- so the in-place build cannot speak for Int8 operands in the TYPE rules; the
    # comparisons are fed plain int8_t and cover Int8 fully
    typed = [t for t in tags if t != OE.I8]
    for name, ref_fn in (("plus", 10), ("minus", 11), ("multiply", 12), ("divide", 13), ("intDiv", 15), ("modulo", 16), ("bitAnd", 40), ("if", 30)):
        for a in typed:
            for b in typed:
                want = R.ref_result_type(ref_fn, a, b)
                args = (OE.U8, a, b) if name == "if" else (a, b)
                got = OE.result_type(OE.FN[name], *args)
                floats = a in (OE.F64, OE.F32) or b in (OE.F64, OE.F32)
                if name in ("intDiv", "modulo", "bitAnd") and floats:
                    assert got is None  # the float forms of these are not carried (they throw / need float -> int casts)
                    continue
                assert (got if got is not None else -1) == want, (name, a, b, got, want)
                # the product's inference through the C ABI
                d = ch.ActionsDAG()
                ins = [d.add_input(j, NP_OF[t]) for j, t in enumerate(args)]
                if name in ("intDiv", "modulo"):
                    ins[1] = d.add_column(3, NP_OF[b])
                d.nodes.append((OE.EX_FUNC, OE.FN[name], 0, tuple(ins) + (-1,) * (3 - len(ins)), 0))
                if want < 0:
                    with pytest.raises(ch.ChgpuError):
                        d.compile()
                else:
                    assert d.compile().node_type(len(d.nodes) - 1) == want, (name, a, b)
                checked += 1
    for a in typed:
        assert (OE.result_type(OE.FN["negate"], a) if OE.result_type(OE.FN["negate"], a) is not None else -1) == R.ref_result_type(14, a, 0)
    assert checked > 500
    rng = np.random.Generator(np.random.PCG64(99))
    n = 20_000
    cols = {t: _random_column(rng, ALL_TYPES[t], n) for t in tags}
    for t in tags:  # make cross-type equalities common
        if np.dtype(ALL_TYPES[t]).kind == "f":
            idx = rng.integers(0, n, size=n // 3)
            cols[t][idx] = cols[OE.I64][idx].astype(ALL_TYPES[t])
            idx = rng.integers(0, n, size=n // 3)
            cols[t][idx] = cols[OE.U64][idx].astype(ALL_TYPES[t])
    assert all(TAG_OF_NP(c.dtype) == t for t, c in cols.items())
    for a in tags:
        for b in tags:
            for fn in range(6):
                assert np.array_equal(OE._compare(fn, cols[a], cols[b]), OE.ref_compare(fn, cols[a], cols[b])), (fn, a, b)


def TAG_OF_NP(dt):
    return OE.TAG_OF[np.dtype(dt)]
