"""MergingAggregatedMemoryEfficientTransform (clickhouse_amd/merging.py): the block-order logic on the CPU engine (oracle), the
device path in tests/test_gpu_round2.py."""
import numpy as np
import pytest

import oracle as O
from clickhouse_amd.merging import LogicalError, MergingAggregatedMemoryEfficientTransform
from cpu_engine import CpuEngine

AGGS = [(O.AGG_SUM, np.int64), (O.AGG_COUNT, None)]


def partial_states(keys, values):
    """one source's partial result: (keys, [sum words, count words]) of an oracle aggregation"""
    a = O.Aggregator(np.uint64, AGGS)
    a.execute_on_block(keys, [values, None])
    k, (s, c) = a.convert_to_block()
    return k, [s.view(np.uint64), c.view(np.uint64)]


def two_level_blocks(k, words):
    """the same result as `split` blocks: [(bucket_num, keys, words)] in increasing bucket_num"""
    b = O.hash_to_selector(k, 256)
    return [(int(x), k[b == x], [w[b == x] for w in words]) for x in np.unique(b)]


def expected(all_keys, all_values):
    uk, inv = np.unique(all_keys, return_inverse=True)
    s = np.zeros(uk.shape[0], dtype=np.uint64)
    np.add.at(s, inv, all_values.astype(np.uint64))
    return uk, s, np.bincount(inv).astype(np.uint64)


def collect(blocks):
    k = np.concatenate([np.asarray(b.keys) for b in blocks]) if blocks else np.zeros(0, dtype=np.uint64)
    s = np.concatenate([np.asarray(b.columns[0]).view(np.uint64) for b in blocks]) if blocks else np.zeros(0, dtype=np.uint64)
    c = np.concatenate([np.asarray(b.columns[1]).view(np.uint64) for b in blocks]) if blocks else np.zeros(0, dtype=np.uint64)
    o = np.argsort(k)
    return k[o], s[o], c[o]


def test_split_sources_are_merged_bucket_by_bucket_as_soon_as_every_source_has_passed_a_bucket():
    rng = np.random.Generator(np.random.PCG64(5))
    srcs = [(rng.integers(0, 5000, size=40_000, dtype=np.uint64), rng.integers(-2**40, 2**40, size=40_000, dtype=np.int64)) for _ in range(3)]
    blocks = [two_level_blocks(*partial_states(k, v)) for k, v in srcs]
    t = MergingAggregatedMemoryEfficientTransform(np.uint64, AGGS, num_inputs=3, engine=CpuEngine())
    out = []
    # source 0 runs ahead: nothing may come out while the others have sent nothing
    for b, k, w in blocks[0][:100]:
        t.add_chunk(0, k, w, bucket_num=b)
    assert t.pull() == []
    # the three sources advance together; after each round the buckets below the slowest source's position are out
    pos = [100, 0, 0]
    while any(pos[i] < len(blocks[i]) for i in range(3)):
        for i in range(3):
            for b, k, w in blocks[i][pos[i]:pos[i] + 37]:
                if i == 1 and k.shape[0] > 1:           # a source may cut one bucket into several blocks
                    h = k.shape[0] // 2
                    t.add_chunk(i, k[:h], [x[:h] for x in w], bucket_num=b)
                    t.add_chunk(i, k[h:], [x[h:] for x in w], bucket_num=b)
                else:
                    t.add_chunk(i, k, w, bucket_num=b)
            pos[i] = min(len(blocks[i]), pos[i] + 37)
            if pos[i] == len(blocks[i]):
                t.finish_input(i)
        got = t.pull()
        limit = min([blocks[i][pos[i] - 1][0] for i in range(3) if pos[i] < len(blocks[i])] or [256])
        assert all(x.bucket_num < limit for x in got)
        out += got
    assert t.pull() == []
    nums = [x.bucket_num for x in out]
    assert nums == sorted(nums) and len(set(nums)) == len(nums) and not any(x.is_overflows for x in out)
    for x in out:                                       # every block holds exactly the keys of its bucket
        assert np.all(O.hash_to_selector(np.asarray(x.keys), 256) == x.bucket_num) and x.rows == np.asarray(x.keys).shape[0]
    want = expected(np.concatenate([k for k, _ in srcs]), np.concatenate([v for _, v in srcs]))
    for g, w in zip(collect(out), want):
        assert np.array_equal(g, w)


def test_an_unsplit_source_among_split_ones_is_split_and_the_overflow_block_comes_last():
    rng = np.random.Generator(np.random.PCG64(6))
    k0, v0 = rng.integers(0, 3000, size=20_000, dtype=np.uint64), rng.integers(0, 100, size=20_000, dtype=np.int64)
    k1, v1 = rng.integers(1000, 4000, size=20_000, dtype=np.uint64), rng.integers(0, 100, size=20_000, dtype=np.int64)
    ko, vo = np.array([7, 8, 9], dtype=np.uint64), np.array([1, 2, 3], dtype=np.int64)
    t = MergingAggregatedMemoryEfficientTransform(np.uint64, AGGS, num_inputs=2, engine=CpuEngine())
    pk, pw = partial_states(k1, v1)
    t.add_chunk(1, pk, pw)                               # bucket_num -1: unsplit
    t.add_chunk(1, *partial_states(ko, vo), is_overflows=True)
    t.finish_input(1)
    assert t.pull() == []                                # source 0 has sent nothing yet
    for b, k, w in two_level_blocks(*partial_states(k0, v0)):
        t.add_chunk(0, k, w, bucket_num=b)
    t.finish_input(0)
    out = t.pull()
    assert out[-1].is_overflows and out[-1].bucket_num == -1 and [x.bucket_num for x in out[:-1]] == sorted(x.bucket_num for x in out[:-1])
    for g, w in zip(collect(out[:-1]), expected(np.concatenate([k0, k1]), np.concatenate([v0, v1]))):
        assert np.array_equal(g, w)
    for g, w in zip(collect(out[-1:]), expected(ko, vo)):
        assert np.array_equal(g, w)


def test_unsplit_sources_alone_give_one_unsplit_block_and_order_violations_are_logical_errors():
    rng = np.random.Generator(np.random.PCG64(7))
    k0, v0 = rng.integers(0, 300, size=5000, dtype=np.uint64), rng.integers(0, 100, size=5000, dtype=np.int64)
    k1, v1 = rng.integers(0, 300, size=5000, dtype=np.uint64), rng.integers(0, 100, size=5000, dtype=np.int64)
    t = MergingAggregatedMemoryEfficientTransform(np.uint64, AGGS, num_inputs=2, engine=CpuEngine())
    t.add_chunk(0, *partial_states(k0, v0))
    t.add_chunk(1, *partial_states(k1, v1))
    t.finish_input(0)
    assert t.pull() == []
    t.finish_input(1)
    out = t.pull()
    assert len(out) == 1 and out[0].bucket_num == -1 and not out[0].is_overflows
    for g, w in zip(collect(out), expected(np.concatenate([k0, k1]), np.concatenate([v0, v1]))):
        assert np.array_equal(g, w)
    t = MergingAggregatedMemoryEfficientTransform(np.uint64, AGGS, num_inputs=1, engine=CpuEngine())
    blocks = two_level_blocks(*partial_states(k0, v0))
    t.add_chunk(0, blocks[5][1], blocks[5][2], bucket_num=blocks[5][0])
    with pytest.raises(LogicalError):
        t.add_chunk(0, blocks[2][1], blocks[2][2], bucket_num=blocks[2][0])   # a bucket below the one the source is at
    t.finish_input(0)
    with pytest.raises(LogicalError):
        t.add_chunk(0, blocks[9][1], blocks[9][2], bucket_num=blocks[9][0])   # after finishing
    with pytest.raises(LogicalError):
        t2 = MergingAggregatedMemoryEfficientTransform(np.uint64, AGGS, num_inputs=1, engine=CpuEngine())
        t2.add_chunk(0, blocks[3][1], blocks[3][2], bucket_num=(blocks[3][0] + 1) % 256)  # keys that do not belong to the declared bucket
        t2.finish_input(0)
        t2.pull()
