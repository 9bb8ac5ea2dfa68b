"""The CPU oracle pinned against the reference's own fixtures: runs without a GPU."""
import numpy as np
import pytest

import scenarios as S


def test_hash_kat_against_compiled_reference_vectors(oracle_mod, golden):
    L = oracle_mod.lib()
    for e in golden["kat"]:
        k = int(e["key"])
        assert L.cho_intHash64(k) == int(e["intHash64"])
        assert L.cho_intHashCRC32(k) == int(e["intHashCRC32"])
        assert L.cho_intHashCRC32_soft(k, 0xFFFFFFFFFFFFFFFF) == int(e["intHashCRC32"])
        assert L.cho_two_level_bucket(int(e["intHashCRC32"])) == e["two_level_bucket"]
        assert L.cho_intHash32(k, 0) == int(e["intHash32_salt0"])
        assert L.cho_sql_intHash32(k) == int(e["intHash32_sql"])
        assert L.cho_intHashCRC32_seed(k, 12345) == int(e["crc_seed_12345"])
        k32 = np.array([k & 0xFFFFFFFF], dtype=np.uint32)
        assert int(oracle_mod.hash_crc32(k32)[0]) == int(e["HashCRC32_UInt32"])


def test_survey_kat_table(oracle_mod):
    # SURVEY.md §8c table (captured from the compiled reference Hash.h)
    L = oracle_mod.lib()
    table = {
        0: (0, 1943489909, 115, 0),
        1: (12994781566227106604, 988491858, 58, 3788511810),
        42: (9297814886316923340, 2928989816, 174, 3458605522),
        0xFFFFFFFF: (14731816277868330182, 0, 0, 3846827201),
        0xFFFFFFFFFFFFFFFF: (7256831767414464289, 3080238136, 183, 2504521962),
    }
    for k, (h64, crc, bucket, h32) in table.items():
        assert L.cho_intHash64(k) == h64
        assert L.cho_intHashCRC32(k) == crc
        assert L.cho_two_level_bucket(crc) == bucket
        assert L.cho_intHash32(k, 0) == h32


def test_hash_against_live_reference_build(oracle_mod):
    R = oracle_mod.ref_hash()
    if R is None:
        pytest.skip("oracle/_ref not built (no reference checkout on this machine)")
    rng = np.random.Generator(np.random.PCG64(7))
    keys = rng.integers(0, 2**64, size=200000, dtype=np.uint64)
    ref = np.empty_like(keys)
    R.ref_intHashCRC32_batch(keys.ctypes.data, keys.shape[0], ref.ctypes.data)
    assert np.array_equal(oracle_mod.hash_crc32(keys), ref)
    R.ref_intHash64_batch(keys.ctypes.data, keys.shape[0], ref.ctypes.data)
    mine = np.array([oracle_mod.lib().cho_intHash64(int(k)) for k in keys[:2000]], dtype=np.uint64)
    assert np.array_equal(mine, ref[:2000])


def test_crc32c_slice_tables_reproduce_crc(oracle_mod):
    t, c = oracle_mod.crc32c_tables()
    rng = np.random.Generator(np.random.PCG64(8))
    keys = rng.integers(0, 2**64, size=5000, dtype=np.uint64)
    acc = np.full(keys.shape[0], c, dtype=np.uint32)
    for j in range(8):
        acc ^= t[j][((keys >> np.uint64(8 * j)) & np.uint64(0xFF)).astype(np.int64)]
    assert np.array_equal(acc.astype(np.uint64), oracle_mod.hash_crc32(keys))


@pytest.mark.parametrize("name,fn", [
    ("00049_any_left_join", S.q00049), ("00050_any_left_join", S.q00050), ("00051_any_inner_join", S.q00051),
    ("00052_all_left_join", S.q00052), ("00053_all_inner_join", S.q00053), ("00055_join_two_numbers", S.q00055),
    ("00041_aggregation_remap", S.q00041), ("00266_read_overflow_mode", S.q00266), ("01091_sum_numbers_1e6", S.q01091),
])
def test_sql_reference_rows(oracle_mod, golden, name, fn):
    assert fn(oracle_mod) == golden["rows"][name]["rows"]


def test_00120_join_group_by_and_sql_hashes(oracle_mod, golden):
    L = oracle_mod.lib()
    got = S.q00120(oracle_mod, L.cho_sql_intHash64, L.cho_sql_intHash32)
    assert got == golden["rows"]["00120_join_and_group_by"]["rows"]


def test_02144_avg_wraps_like_reference(oracle_mod, golden):
    want = float(golden["rows"]["02144_avg_ubsan"]["rows"][0][0])
    for got in S.q02144(oracle_mod):
        assert f"{got:.2f}" == f"{want:.2f}"


def test_01300_avg_float64_group_by(oracle_mod, golden):
    want = sorted(float(r[0]) for r in golden["rows"]["01300_avg_group_by_mod5"]["rows"])
    got = S.q01300(oracle_mod)
    assert [round(g, 6) for g in got] == want


# ---- round 2: keys128 / keys256 restatement pinned by the compiled reference's vectors and by 00120 as a true two-key GROUP BY ------------
def test_keys_fixed_hashes_match_reference_vectors(oracle_mod):
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "round2_kat.json")))["keys_fixed"]
    for v in kat:
        w = np.array([int(x) for x in v["words"]], dtype=np.uint64)
        assert oracle_mod.hash_keys_fixed(w[:2].view(np.uint8)) == int(v["UInt128HashCRC32"])
        assert oracle_mod.hash_keys_fixed(w.view(np.uint8)) == int(v["UInt256HashCRC32"])


def test_keys_fixed_pack_and_group_by_two_keys(oracle_mod, golden):
    O = oracle_mod
    L = O.lib()
    a = np.array([1, 2, 3], dtype=np.uint64)
    b = np.array([7, 8, 9], dtype=np.uint32)
    c = np.array([0x1234, 0xFFFF, 0], dtype=np.uint16)
    p = O.pack_fixed([a, b, c])
    assert p.shape == (3, 16)
    assert bytes(p[1]) == (2).to_bytes(8, "little") + (8).to_bytes(4, "little") + (0xFFFF).to_bytes(2, "little") + bytes(2)   # packFixed: consecutive, zero padded
    # 00120_join_and_group_by: GROUP BY (intHash64(number), intHash32(number)) -> sum(number): 12 key bytes = keys128
    n = np.arange(10, dtype=np.uint64)
    v1 = np.array([L.cho_sql_intHash64(int(x)) for x in n], dtype=np.uint64)
    v2 = np.array([L.cho_sql_intHash32(int(x)) for x in n], dtype=np.uint32)
    A = O.KeysFixedAggregator([np.uint64, np.uint32], [(O.AGG_SUM, np.uint64)])
    A.execute_on_block([v1, v2], [n])
    (k1, k2), (s,) = A.convert_to_block()
    rows = sorted([[str(int(x)), str(int(y)), str(int(z))] for x, y, z in zip(k1, k2, s)], key=lambda r: (int(r[0]), int(r[1])))
    assert rows == golden["rows"]["00120_join_and_group_by"]["rows"]
    # the zero key, growth, find-without-insert
    m = O.WideKeyMap(32)
    rng = np.random.Generator(np.random.PCG64(3))
    keys = rng.integers(0, 3, size=(5000, 32), dtype=np.uint8)
    keys[0] = 0
    uniq, first = np.unique(keys, axis=0, return_index=True)
    ids = m.batch(keys, True)
    assert len(m) == uniq.shape[0] and ids[0] == 0 and np.array_equal(m.keys()[ids.astype(np.int64)], keys)
    probe = np.concatenate([keys[:10], np.full((1, 32), 9, dtype=np.uint8)])
    got = m.batch(probe, False)
    assert np.array_equal(got[:10], ids[:10]) and got[10] == 2**64 - 1 and len(m) == uniq.shape[0]


# ---- round 3: min / max states per group, pinned by the reference's rows of 01300 (Float64 max) and 01321 (integer min / max) ----------------
def test_01300_max_float64_group_by(oracle_mod, golden):
    want = sorted(float(r[0]) for r in golden["rows"]["01300_max_group_by_mod2_mod3"]["rows"])
    assert S.q01300_max(oracle_mod) == want


def test_01321_min_max_group_by(oracle_mod, golden):
    assert S.q01321_min_max(oracle_mod) == sorted(golden["rows"]["01321_min_max_group_by_mod2_mod3"]["rows"])
    assert S.q01321_max_product(oracle_mod) == sorted(int(r[0]) for r in golden["rows"]["01321_max_product_group_by_mod7_mod5"]["rows"])
    assert S.q01321_any(oracle_mod) == sorted(golden["rows"]["01321_any_group_by_mod2_mod3"]["rows"])


# ---- round 3: ASOF joins, pinned by the rows of the reference's 00927 tests --------------------------------------------------------------------
def test_00927_asof_joins(oracle_mod, golden):
    O = oracle_mod

    def jp(build, lk, lt, left):
        return O.asof_pairs(build, lk, lt, O.ASOF_GREATER_OR_EQUALS, None, left)
    assert S.asof_noninclusive(jp) == golden["rows"]["00927_asof_noninclusive"]["rows"]
    assert S.asof_joins_left(jp) == golden["rows"]["00927_asof_joins_left"]["rows"]
    # 00927_asof_join_long prints 3000000 for 1000 keys (3000 per key: every trade time 10 i meets the tvs time 3 * floor(10 i / 3)); the
    # pure-Python restatement runs 20 of the keys, the device runs all of them (tests/test_gpu_agg_join.py)
    assert int(golden["rows"]["00927_asof_join_long"]["rows"][0][0]) == 3_000_000
    assert S.asof_join_long(jp, keys=20) == [[str(3000 * 20)]]
