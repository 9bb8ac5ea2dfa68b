"""SSB Q3.1 with the schema's real key types, end to end on one GPU, from compressed column data — every §8(f) piece in one plan
(used by tests/test_gpu_ssb.py and tools/bench_ssb_q31.py):

    SELECT c_nation, s_nation, toYear(lo_orderdate) AS year, sum(lo_revenue) AS revenue
    FROM lineorder JOIN customer ON lo_custkey = c_custkey JOIN supplier ON lo_suppkey = s_suppkey
    WHERE c_region = 'ASIA' AND s_region = 'ASIA' AND year >= 1992 AND year <= 1997
    GROUP BY c_nation, s_nation, year ORDER BY year ASC, revenue DESC

  * lineorder columns arrive as compressed frames (CODEC LZ4; lo_orderdate is a Date with CODEC(Delta(2), LZ4)) and are decoded in
    HBM (chgpu_decompress_frames);
  * c_region / c_nation / s_region / s_nation are String columns: dictionary-encoded on the device, region predicates become id
    comparisons, nation ids of both tables are unified through ONE query-wide dictionary;
  * toYear + the year range + the projection run as one run-time compiled WHERE + projection kernel pair;
  * the filtered dimensions are the right sides of INNER ALL hash joins (unique keys), nation ids gathered through the row ids;
  * GROUP BY packs (c_nation id, s_nation id, year) into 6 key bytes; ORDER BY year ASC, revenue DESC is two stable radix sorts.
The CPU side restates the query with numpy + Python dicts over the uncompressed host arrays.
"""
import numpy as np

REGIONS = [b"AFRICA", b"AMERICA", b"ASIA", b"EUROPE", b"MIDDLE EAST"]
NATIONS = [f"NATION-{r.decode()[:3]}-{k}".encode() for r in REGIONS for k in range(5)]  # 25 nations, 5 per region
DAY0 = 8035  # 1992-01-01 as a Date (days since 1970-01-01)
N_DAYS = 2556  # .. 1998-12-30


def gen(rows, customers, suppliers, seed=17):
    rng = np.random.Generator(np.random.PCG64(seed))
    c_nat = rng.integers(0, 25, size=customers)
    s_nat = rng.integers(0, 25, size=suppliers)
    dims = dict(c_custkey=np.arange(1, customers + 1, dtype=np.uint32), c_nation=[NATIONS[i] for i in c_nat], c_region=[REGIONS[i // 5] for i in c_nat],
                s_suppkey=np.arange(1, suppliers + 1, dtype=np.uint32), s_nation=[NATIONS[i] for i in s_nat], s_region=[REGIONS[i // 5] for i in s_nat],
                _c_nat=c_nat, _s_nat=s_nat)
    for name in ("c_nation", "c_region", "s_nation", "s_region"):
        dims[name + "_str"] = column_string(dims[name])  # the Block's native ColumnString layout (built outside any timed region)
    lo = dict(lo_custkey=rng.integers(1, customers + 1, size=rows).astype(np.uint32),
              lo_suppkey=rng.integers(1, suppliers + 1, size=rows).astype(np.uint32),
              lo_orderdate=(DAY0 + np.sort(rng.integers(0, N_DAYS, size=rows))).astype(np.uint16),  # a fact table ordered by date
              lo_revenue=rng.integers(0, 1_000_000, size=rows).astype(np.uint32))
    return dims, lo


def column_string(values):
    """ColumnString layout of a host column: (offsets UInt64, chars UInt8 with a zero after every value)"""
    lens = np.fromiter((len(v) + 1 for v in values), dtype=np.uint64, count=len(values))
    return np.cumsum(lens, dtype=np.uint64), np.frombuffer(b"".join(v + b"\0" for v in values), dtype=np.uint8)


def compress_lineorder(OC, lo, block=65536):
    """what the storage layer would hand over: one compressed column file per column"""
    return {k: OC.write_frames(v.tobytes(), block, OC.DELTA_LZ4 if k == "lo_orderdate" else OC.METHOD_LZ4, 2) for k, v in lo.items()}


def q31_gpu(ch, ctx, dims, lo_files, dtypes, times=None):
    import time
    from clickhouse_amd import compression as CC
    t_prev = [time.perf_counter()]

    def lap(name):
        if times is not None:
            ctx.synchronize()
            now = time.perf_counter()
            times[name] = times.get(name, 0.0) + (now - t_prev[0]) * 1e3
            t_prev[0] = now
    # ---- dimensions: String columns -> ids on the device; region predicate on ids; nations through one dictionary ----
    nations = ch.LowCardinalityDictionary(ctx)

    def cs(pair):
        return ch.ColumnString(ctx.upload(pair[0]), ctx.upload(pair[1]))

    def dimension(key, nation, region):
        reg = cs(region).dictionary_encode()
        if b"ASIA" not in reg.dictionary:
            return None
        m = ch.cmp_const(reg.indexes, ch.EQ, reg.dictionary.index(b"ASIA"))
        nat = nations.map_block(cs(nation).dictionary_encode())
        k, n = ch.filter_columns([ctx.upload(key), nat], m)
        j = ch.HashJoin(ch.JOIN_INNER, ch.STRICT_ALL, key_dtype=np.uint32, ctx=ctx)
        j.add_block(k)
        j.finish_build()
        return j, n
    jc, c_nat = dimension(dims["c_custkey"], dims["c_nation_str"], dims["c_region_str"])
    js, s_nat = dimension(dims["s_suppkey"], dims["s_nation_str"], dims["s_region_str"])
    lap("dimensions: String dictionary encode, region filter, join build")
    # ---- fact columns: compressed frames -> HBM ----
    lo = {k: CC.read_column_file(ctx, buf, dtypes[k]) for k, buf in lo_files.items()}
    lap("fact columns: upload of compressed bytes + decode in HBM")
    # ---- WHERE toYear(lo_orderdate) BETWEEN 1992 AND 1997 + projection, one generated kernel pair ----
    d = ch.ActionsDAG()
    cust, supp, date, rev = d.add_input(0, np.uint32), d.add_input(1, np.uint32), d.add_input(2, np.uint16), d.add_input(3, np.uint32)
    year = d.add_function("toYear", date)
    f = d.add_function("and", d.add_function("greaterOrEquals", year, d.add_column(1992, np.uint16)),
                       d.add_function("lessOrEquals", year, d.add_column(1997, np.uint16)))
    ex = d.compile()
    (cust_c, supp_c, year_c, rev_c), kept = ex.filter_execute(ctx, [lo["lo_custkey"], lo["lo_suppkey"], lo["lo_orderdate"], lo["lo_revenue"]], f, [cust, supp, year, rev])
    lap("WHERE toYear(...) BETWEEN + projection (generated kernels)")
    # ---- joins: supplier first (smaller), then customer; nation ids gathered through the row ids ----
    r = js.probe_columns(supp_c)
    off = r["offsets"]
    cust_c, year_c, rev_c = (c.replicate(off) for c in (cust_c, year_c, rev_c))
    sn = s_nat.index(r["right_rowid"])
    r = jc.probe_columns(cust_c)
    off = r["offsets"]
    year_c, rev_c, sn = (c.replicate(off) for c in (year_c, rev_c, sn))
    cn = c_nat.index(r["right_rowid"])
    lap("two hash joins + replicate + nation gather")
    # ---- GROUP BY c_nation, s_nation, year: ids narrowed and packed into 6 key bytes ----
    nd = ch.ActionsDAG()
    nd.add_function("toUInt16", nd.add_input(0, np.uint32))
    narrow = nd.compile()
    key = ch.pack_fixed_keys([narrow.execute(ctx, [cn], [1])[0], narrow.execute(ctx, [sn], [1])[0], year_c])
    agg = ch.Aggregator(np.uint64, [(ch.AGG_SUM, np.uint32)], ctx=ctx)
    agg.execute_on_block(key, [rev_c])
    keys_c, (revenue,) = agg.finalize_columns()
    lap("GROUP BY packed keys")
    # ---- ORDER BY year ASC, revenue DESC ----
    cols = [ch.unpack_fixed_key(keys_c, 0, np.uint16), ch.unpack_fixed_key(keys_c, 2, np.uint16), ch.unpack_fixed_key(keys_c, 4, np.uint16), revenue]
    (cn_o, sn_o, y_o, r_o), _ = ch.sort_block(cols, [(2, False, 1), (3, True, 1)])
    cn_h, sn_h = nations.decode(cn_o.numpy()), nations.decode(sn_o.numpy())
    lap("ORDER BY + decode of the result keys")
    return [(a, b, int(y), int(v)) for a, b, y, v in zip(cn_h, sn_h, y_o.numpy(), r_o.numpy())], kept


def q31_cpu(dims, lo):
    """the query restated with numpy: filtered dimensions as lookup tables, one pass over the fact rows, dict GROUP BY, sorted()"""
    c_ok = np.zeros(dims["c_custkey"].shape[0] + 1, dtype=bool)
    c_ok[1:] = dims["_c_nat"] // 5 == 2
    s_ok = np.zeros(dims["s_suppkey"].shape[0] + 1, dtype=bool)
    s_ok[1:] = dims["_s_nat"] // 5 == 2
    c_nat = np.concatenate([[0], dims["_c_nat"]])
    s_nat = np.concatenate([[0], dims["_s_nat"]])
    days = lo["lo_orderdate"].astype("int64").astype("datetime64[D]")
    year = days.astype("datetime64[Y]").astype(np.int64) + 1970
    m = c_ok[lo["lo_custkey"]] & s_ok[lo["lo_suppkey"]] & (year >= 1992) & (year <= 1997)
    key = (c_nat[lo["lo_custkey"][m]] * 25 + s_nat[lo["lo_suppkey"][m]]) * 10000 + year[m]
    uk, inv = np.unique(key, return_inverse=True)
    sums = np.zeros(uk.shape[0], dtype=np.uint64)
    np.add.at(sums, inv, lo["lo_revenue"][m].astype(np.uint64))
    rows = [(NATIONS[int(k // 10000) // 25], NATIONS[int(k // 10000) % 25], int(k % 10000), int(s)) for k, s in zip(uk, sums)]
    return rows, int(m.sum())


def ordered(rows):
    """ORDER BY year ASC, revenue DESC (ties in revenue are not ordered by the query)"""
    return sorted(rows, key=lambda r: (r[2], -r[3]))
