#!/usr/bin/env python3
"""Generates tests/golden/*.json from the reference checkout (run in the build container only).

What is committed are DATA fixtures: the expected-output rows of the reference's own stateless tests
(tests/queries/0_stateless/*.reference) and known-answer vectors produced by running the reference's own
Hash.h (compiled in place into oracle/_ref by oracle/Makefile).  No reference source or SQL text is stored;
the queries are restated as Python input builders in tests/test_oracle_golden.py.

Usage: python tests/golden/make_golden.py [/root/reference]
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def rows_of(ref_root, name, first=None, last=None):
    path = os.path.join(ref_root, "tests/queries/0_stateless", name + ".reference")
    with open(path) as f:
        lines = [l.rstrip("\n") for l in f]
    lines = lines[first:last]
    return [l.split("\t") for l in lines]


def main():
    ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = {}

    # --- join semantics ---------------------------------------------------------------------
    out["00049_any_left_join"] = dict(source="tests/queries/0_stateless/00049_any_left_join.reference",
                                      rows=rows_of(ref_root, "00049_any_left_join"))
    out["00050_any_left_join"] = dict(source="tests/queries/0_stateless/00050_any_left_join.reference",
                                      rows=rows_of(ref_root, "00050_any_left_join"))
    out["00051_any_inner_join"] = dict(source="tests/queries/0_stateless/00051_any_inner_join.reference",
                                       rows=rows_of(ref_root, "00051_any_inner_join"))
    out["00052_all_left_join"] = dict(source="tests/queries/0_stateless/00052_all_left_join.reference",
                                      rows=rows_of(ref_root, "00052_all_left_join"))
    out["00053_all_inner_join"] = dict(source="tests/queries/0_stateless/00053_all_inner_join.reference",
                                       rows=rows_of(ref_root, "00053_all_inner_join"))
    out["00055_join_two_numbers"] = dict(source="tests/queries/0_stateless/00055_join_two_numbers.reference",
                                         rows=rows_of(ref_root, "00055_join_two_numbers"))
    # join + group by + SQL intHash64/intHash32 KAT (first line of the .reference is the other query's empty row)
    out["00120_join_and_group_by"] = dict(source="tests/queries/0_stateless/00120_join_and_group_by.reference",
                                          rows=rows_of(ref_root, "00120_join_and_group_by", 1, None))
    # --- group by / aggregate semantics -------------------------------------------------------
    out["00041_aggregation_remap"] = dict(source="tests/queries/0_stateless/00041_aggregation_remap.reference",
                                          rows=rows_of(ref_root, "00041_aggregation_remap"))
    out["00266_read_overflow_mode"] = dict(source="tests/queries/0_stateless/00266_read_overflow_mode.reference",
                                           rows=rows_of(ref_root, "00266_read_overflow_mode"))
    # avg(-8e18) over 65535*2 rows: the value lines (the .reference echoes the queries in between)
    avg_rows = [r for r in rows_of(ref_root, "02144_avg_ubsan") if len(r) == 1 and r[0] and r[0][0].isdigit()]
    out["02144_avg_ubsan"] = dict(source="tests/queries/0_stateless/02144_avg_ubsan.reference", rows=avg_rows)
    # sum(number) over numbers(1000000)
    out["01091_sum_numbers_1e6"] = dict(source="tests/queries/0_stateless/01091_num_threads.reference",
                                        rows=[rows_of(ref_root, "01091_num_threads")[2]])
    # round(avg(log(2)*number), 6) GROUP BY number % 5 over numbers(1e7): lines 7..11 of the .reference
    out["01300_avg_group_by_mod5"] = dict(source="tests/queries/0_stateless/01300_group_by_other_keys.reference",
                                          rows=rows_of(ref_root, "01300_group_by_other_keys", 6, 11))
    # min / max states per group (round 3): max(log(2) * number) GROUP BY number % 2, number % 3 -- and the integer forms of 01321
    out["01300_max_group_by_mod2_mod3"] = dict(source="tests/queries/0_stateless/01300_group_by_other_keys.reference",
                                               rows=rows_of(ref_root, "01300_group_by_other_keys", 0, 6))
    out["01321_min_max_group_by_mod2_mod3"] = dict(source="tests/queries/0_stateless/01321_aggregate_functions_of_group_by_keys.reference",
                                                   rows=rows_of(ref_root, "01321_aggregate_functions_of_group_by_keys", 0, 6))
    # any(number % 2), anyLast(number % 3) over the same groups (second query of the file): both arguments are constant inside a group, so
    # the rows pin any() -- first value = last value = the group's key
    out["01321_any_group_by_mod2_mod3"] = dict(source="tests/queries/0_stateless/01321_aggregate_functions_of_group_by_keys.reference",
                                               rows=rows_of(ref_root, "01321_aggregate_functions_of_group_by_keys", 6, 12))
    # ASOF joins (round 3): the three queries of 00927_asof_join_noninclusive (LEFT, INNER with `A.t >= B.t`, ASOF JOIN USING), the LEFT join of
    # 00927_asof_joins and the checksum of 00927_asof_join_long (1e7 build rows, 3e6 probe rows)
    out["00927_asof_noninclusive"] = dict(source="tests/queries/0_stateless/00927_asof_join_noninclusive.reference",
                                          rows=rows_of(ref_root, "00927_asof_join_noninclusive", 0, 29))
    out["00927_asof_joins_left"] = dict(source="tests/queries/0_stateless/00927_asof_joins.reference",
                                        rows=rows_of(ref_root, "00927_asof_joins", 0, 14))
    out["00927_asof_join_long"] = dict(source="tests/queries/0_stateless/00927_asof_join_long.reference",
                                       rows=rows_of(ref_root, "00927_asof_join_long", 0, 1))
    out["01321_max_product_group_by_mod7_mod5"] = dict(source="tests/queries/0_stateless/01321_aggregate_functions_of_group_by_keys.reference",
                                                       rows=rows_of(ref_root, "01321_aggregate_functions_of_group_by_keys", 12, 47))

    with open(os.path.join(HERE, "sql_reference_rows.json"), "w") as f:
        json.dump(out, f, indent=1)

    # --- hash KATs from the compiled reference Hash.h ---------------------------------------------
    import oracle
    oracle.build()
    R = oracle.ref_hash()
    assert R is not None, "oracle/_ref not built (reference checkout missing?)"
    fixed = [0, 1, 2, 42, 1000000, 0xFFFFFFFF, 0x0123456789ABCDEF, 0xFFFFFFFFFFFFFFFF]
    rng = np.random.Generator(np.random.PCG64(20250711))
    rnd = [int(x) for x in rng.integers(0, 2**64, size=256, dtype=np.uint64)]
    kat = []
    for k in fixed + rnd:
        crc = R.ref_intHashCRC32(k)
        kat.append(dict(key=str(k), intHash64=str(R.ref_intHash64(k)), intHashCRC32=str(crc),
                        two_level_bucket=(crc >> 24) & 0xFF, intHash32_salt0=str(R.ref_intHash32_salt0(k)),
                        intHash32_sql=str(R.ref_intHash32_sql(k)),
                        HashCRC32_UInt32=str(R.ref_HashCRC32_UInt32(k & 0xFFFFFFFF)),
                        crc_seed_12345=str(R.ref_intHashCRC32_seed(k, 12345))))
    with open(os.path.join(HERE, "hash_kat.json"), "w") as f:
        json.dump(dict(source="src/Common/HashTable/Hash.h compiled in place (oracle/ref_hash_wrapper.cpp)", kat=kat), f, indent=1)
    print("wrote", os.path.join(HERE, "sql_reference_rows.json"), "and hash_kat.json")
    make_cmp_kat(ref_root)
    make_codec_kat(ref_root)
    make_mod_kat(ref_root)
    make_sort_kat(ref_root)
    make_string_kat(ref_root)
    make_round2_kat(ref_root)


def make_round2_kat(ref_root):
    """Round-2 fixtures: CityHash128 vectors from the reference's own contrib/cityhash102 compiled in place (oracle/ref_city_wrapper.cpp),
    UInt128HashCRC32 / UInt256HashCRC32 vectors from its Hash.h (oracle/ref_hash_wrapper.cpp), and the serialized aggregate-state bytes the
    reference's tests expect: hex(avgState(number)) over numbers(10) (01926_bin_unbin), hex(countState) over 10 rows
    (00357_to_string_complex_types), hex(countState(if(even, number, null))) over numbers(5) (03210_...return_type_bug)."""
    import ctypes as C
    import oracle
    oracle.build()
    city = C.CDLL(os.path.join(REPO, "oracle", "_ref", "libchref_city.so"))
    rng = np.random.Generator(np.random.PCG64(102))
    vecs = []
    for n in list(range(0, 40)) + [63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4096 + 25]:
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        out = (C.c_uint64 * 2)()
        city.ref_CityHash128(data, n, out)
        vecs.append(dict(hex=data.hex(), low64=str(out[0]), high64=str(out[1])))
    R = oracle.ref_hash()
    wide = []
    for _ in range(32):
        w = [int(x) for x in rng.integers(0, 2**64, size=4, dtype=np.uint64)]
        if _ < 4:
            w = [[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [2**64 - 1] * 4][_]
        wide.append(dict(words=[str(x) for x in w], UInt128HashCRC32=str(R.ref_UInt128HashCRC32(w[0], w[1])),
                         UInt256HashCRC32=str(R.ref_UInt256HashCRC32(*w))))
    def last_line(name):
        with open(os.path.join(ref_root, "tests/queries/0_stateless", name + ".reference")) as f:
            return [l.rstrip("\n") for l in f]
    bin_unbin = last_line("01926_bin_unbin")
    states = dict(avgState_numbers10=dict(source="tests/queries/0_stateless/01926_bin_unbin.reference", hex=[l for l in bin_unbin if l == l.upper() and len(l) == 18][0]),
                  countState_10rows=dict(source="tests/queries/0_stateless/00357_to_string_complex_types.reference", hex=[l for l in last_line("00357_to_string_complex_types") if l == "0A"][0]),
                  countState_3rows=dict(source="tests/queries/0_stateless/03210_optimize_rewrite_aggregate_function_with_if_return_type_bug.reference",
                                        hex=last_line("03210_optimize_rewrite_aggregate_function_with_if_return_type_bug")[0]))
    with open(os.path.join(HERE, "round2_kat.json"), "w") as f:
        json.dump(dict(city_hash128_source="contrib/cityhash102 (CityHash_v1_0_2::CityHash128) compiled in place", city_hash128=vecs,
                       keys_fixed_source="src/Common/HashTable/Hash.h UInt128HashCRC32 / UInt256HashCRC32 compiled in place", keys_fixed=wide,
                       agg_states=states), f, indent=1)
    print("wrote round2_kat.json:", len(vecs), "city vectors,", len(wide), "wide-key vectors, states", {k: v["hex"] for k, v in states.items()})


def make_string_kat(ref_root):
    """String keys: expected rows of 00054_join_string (ALL LEFT JOIN USING a String key) and 00127_group_by_concat (GROUP BY a String
    and a number).  The inputs are restated in tests/test_lowcardinality.py."""
    out = {"00054_join_string": dict(source="tests/queries/0_stateless/00054_join_string.reference", rows=rows_of(ref_root, "00054_join_string")),
           "00127_group_by_concat": dict(source="tests/queries/0_stateless/00127_group_by_concat.reference", rows=rows_of(ref_root, "00127_group_by_concat")),
           # ALL FULL OUTER JOIN with non-joined right rows padded with defaults (Date 1970-01-01, 0)
           "00974_full_outer_join": dict(source="tests/queries/0_stateless/00974_full_outer_join.reference", rows=rows_of(ref_root, "00974_full_outer_join")),
           "00056_join_number_string": dict(source="tests/queries/0_stateless/00056_join_number_string.reference", rows=rows_of(ref_root, "00056_join_number_string"))}
    with open(os.path.join(HERE, "string_key_rows.json"), "w") as f:
        json.dump(out, f)
    print("wrote string_key_rows.json")


def make_sort_kat(ref_root):
    """NaN placement in ORDER BY: the expected output blocks of 03447_float_nan_order (the .reference echoes a '--- <name>' line before
    each block).  The sorted column -- number + number / number over numbers(3) / numbers(256) -- is restated in tests/test_sorting.py."""
    path = os.path.join(ref_root, "tests/queries/0_stateless", "03447_float_nan_order.reference")
    blocks, name = {}, None
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("--- "):
                name = line[4:]
                blocks[name] = []
            elif name is not None and line:
                blocks[name].append(line)
    with open(os.path.join(HERE, "sort_nan_order.json"), "w") as f:
        json.dump(dict(source="tests/queries/0_stateless/03447_float_nan_order.reference", blocks=blocks), f)
    print("wrote sort_nan_order.json:", len(blocks), "blocks")


def make_mod_kat(ref_root):
    """modulo known answers: the expected rows of 01700_mod_negative_type_promotion (value + result type name; the integer forms,
    i.e. its first seven lines) and of 00516_modulo (values).  The operands are restated as typed inputs in tests/test_expr_dag.py."""
    out = {"01700_mod_negative_type_promotion": dict(source="tests/queries/0_stateless/01700_mod_negative_type_promotion.reference",
                                                      rows=rows_of(ref_root, "01700_mod_negative_type_promotion", 0, 7)),
           "00516_modulo": dict(source="tests/queries/0_stateless/00516_modulo.reference", rows=rows_of(ref_root, "00516_modulo")),
           # 00977_int_div: four intDiv forms over numbers(1, 10), ten result lines each (lines 4-13, 24-33, 44-53, 64-73 of the .reference)
           "00977_int_div": dict(source="tests/queries/0_stateless/00977_int_div.reference",
                                 rows=[rows_of(ref_root, "00977_int_div", lo, lo + 10) for lo in (3, 23, 43, 63)]),
           # 00479_date_and_datetime_to_number: toYYYYMM / toYYYYMMDD of toDate('2017-07-21') -- the first two lines
           "00479_toYYYYMM_of_date_2017_07_21": dict(source="tests/queries/0_stateless/00479_date_and_datetime_to_number.reference",
                                                    rows=rows_of(ref_root, "00479_date_and_datetime_to_number", 0, 2))}
    with open(os.path.join(HERE, "expr_mod_kat.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote expr_mod_kat.json")


def make_cmp_kat(ref_root):
    """accurate comparison known answers: 00411_long_accurate_number_comparison_float.  Every query line compares one
    integer with one Float64 literal as `i op f` and `f op i` (op in =, !=, <, <=, >, >=), first with the bare integer
    literal and then through every toUInt8 ... toInt64 cast the value fits; the .reference holds the 12 answers per form.
    Stored: the two literals, the integer types in query order, the answers.  (No SQL text is stored.)"""
    import re
    base = os.path.join(ref_root, "tests/queries/0_stateless", "00411_long_accurate_number_comparison_float")
    with open(base + ".sql") as f:
        queries = [l for l in f if l.startswith("SELECT")]
    with open(base + ".reference") as f:
        answers = [l.rstrip("\n").split("\t") for l in f if l.strip()]
    assert len(queries) == len(answers)
    cases = []
    for q, a in zip(queries, answers):
        types = []
        for t in re.findall(r"to(U?Int\d+)\(", q):
            if t not in types:
                types.append(t)
        res = [int(x) for x in a[2:]]
        assert len(res) == 12 * (1 + len(types)), (a[0], a[1], len(res), types)
        cases.append(dict(int=a[0], float=a[1], types=["literal"] + types, answers=[res[k * 12:(k + 1) * 12] for k in range(1 + len(types))]))
    with open(os.path.join(HERE, "expr_cmp_kat.json"), "w") as f:
        json.dump(dict(source="tests/queries/0_stateless/00411_long_accurate_number_comparison_float.{sql,reference}",
                       order="i=f i!=f i<f i<=f i>f i>=f f=i f!=i f<i f<=i f>i f>=i", cases=cases), f)
    print("wrote expr_cmp_kat.json:", len(cases), "cases")


def make_codec_kat(ref_root):
    """DoubleDelta's compatibility vectors: the (sequence, frame bytes) pairs of src/Compression/tests/gtest_compressionCodec.cpp:1171-1209
    (the sequences are restated by tests/test_compression.py::dd_compat_sequence; only the expected BYTES are taken from the file) and the
    two worked examples of the codec's own documentation comment (CompressionCodecDoubleDelta.cpp:73-118)."""
    import re
    path = os.path.join(ref_root, "src/Compression/tests/gtest_compressionCodec.cpp")
    text = open(path).read()
    a = text.index("INSTANTIATE_TEST_SUITE_P(DoubleDelta,")
    b = text.index("template <typename ValueType>", a)
    block = text[a:b]
    vectors = []
    for m in re.finditer(r'DDCompatibilityTestSequence<(\w+)>\(\),\s*BIN_STR\("((?:[^"\\]|\\.)*)"\)', block):
        raw = m.group(2)
        data = bytes(int(h, 16) for h in re.findall(r"\\x([0-9a-fA-F]{2})", raw))
        assert len(re.findall(r"\\x[0-9a-fA-F]{2}", raw)) * 4 == len(raw), "only \\xNN escapes expected"
        vectors.append(dict(type=m.group(1), frame_hex=data.hex()))
    assert [v["type"] for v in vectors] == ["Int8", "UInt8", "Int16", "UInt16", "Int32", "UInt32", "Int64", "UInt64"]
    doc = [dict(type="UInt8", values=list(range(1, 11)), payload_hex="0a0000000101" + "00"),
           dict(type="Int16", values=[-10, 10, -20, 20, -40, 40], payload_hex="06000000f6ff1400b8e22eb1e458")]
    gorilla_doc = [dict(type="Float32", values=[0.1, 0.1, 0.11, 0.2, 0.1], payload_hex="05000000cdcccc3d6a5ad8b63ccd75b16c77000000")]
    with open(os.path.join(HERE, "codec_kat.json"), "w") as f:
        json.dump(dict(gorilla_doc_examples=gorilla_doc, gorilla_source="the worked example in src/Compression/CompressionCodecGorilla.cpp:58-104 (payload after "
                       "[width][bytes_to_skip])", source="src/Compression/tests/gtest_compressionCodec.cpp:1171-1209 (frames: method byte 0x94, compressed size, "
                              "decompressed size, codec payload) and the worked examples in CompressionCodecDoubleDelta.cpp:73-118 "
                              "(payload after [width][bytes_to_skip])",
                       double_delta_frames=vectors, double_delta_doc_examples=doc), f, indent=1)
    print("wrote codec_kat.json:", len(vectors), "frames")


def make_native_lc_blocks(ref_root):
    """tests/queries/0_stateless/02010_lc_native.python sends four hand-written Native blocks with one LowCardinality(String) column over
    the wire (:198-375: the bytes after the packet type and the external table name are a Native block at revision 54449 -- BlockInfo,
    dimensions, name, type, then the LowCardinality serialization); 02010_lc_native.reference holds what the server answers.  The byte
    fields below restate the script's literals (version, index type + flags, keys, indexes); the expected messages are read from the
    .reference file."""
    ref = open(os.path.join(ref_root, "tests/queries/0_stateless/02010_lc_native.reference")).read().splitlines()
    answers = [ln.split(":", 1)[1].strip() for ln in ref if ln.startswith("code 117:")]
    assert len(answers) == 3

    def varuint(x):
        out = bytearray()
        while x >= 0x80:
            out.append((x & 0x7F) | 0x80)
            x >>= 7
        out.append(x)
        return bytes(out)

    def string(b):
        return varuint(len(b)) + b

    def block(index_type_bytes, index_bytes):
        ba = bytearray()
        ba += varuint(1) + bytes([0]) + varuint(2) + bytes([0] * 4) + varuint(0)               # serializeBlockInfo :148-153: is_overflows 0, bucket_num bytes 0
        ba += varuint(1) + varuint(1)                                                           # one column, one row
        ba += string(b"x") + string(b"LowCardinality(String)")
        ba += bytes([1] + [0] * 7)                                                              # SharedDictionariesWithAdditionalKeys
        ba += bytes(index_type_bytes + [0] * 6)
        ba += bytes([1] + [0] * 7) + string(b"hello")                                           # one key
        ba += bytes([1] + [0] * 7) + bytes(index_bytes)                                         # one index
        return bytes(ba).hex()

    cases = [
        dict(name="valid", source_lines="198-233", block_hex=block([3, 2], [0] * 8), values=["hello"], error=None),
        dict(name="index_overflow", source_lines="236-270", block_hex=block([3, 2], [0] * 7 + [1]), values=None, error=answers[0]),
        dict(name="global_dictionary", source_lines="273-307", block_hex=block([3, 3], [0] * 8), values=None, error=answers[1]),
        dict(name="no_additional_keys", source_lines="310-344", block_hex=block([3, 0], [0] * 8), values=None, error=answers[2]),
    ]
    out = dict(source="tests/queries/0_stateless/02010_lc_native.python + .reference", server_revision=54449, cases=cases)
    with open(os.path.join(HERE, "native_lc_blocks.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote native_lc_blocks.json:", len(cases), "cases")


if __name__ == "__main__":
    main()
    make_native_lc_blocks(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
