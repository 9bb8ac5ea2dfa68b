"""TEST INFRASTRUCTURE ONLY — CPU restatement of GROUP BY over a LowCardinality key column.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

The low_cardinality_key* variants (src/Interpreters/AggregatedDataVariants.h:119-127) aggregate by the dictionary VALUE a
row's index points at: HashMethodSingleLowCardinalityColumn (src/Common/ColumnsHashing.h:82-260) emplaces
dictionary[index[row]] (through its per-position cache), so rows of different Blocks whose dictionaries hold the same value
at different positions meet in one group, and the result key column holds each value once.  Restated with a plain Python
dict over the converted-to-full column (ColumnLowCardinality::convertToFullColumn, ColumnLowCardinality.h:53).
Parity pinning: the String-key paths are checked end to end against the reference's expected rows of 00054_join_string and
00127_group_by_concat (tests/golden/string_key_rows.json); the helpers in this file have no reference vectors of their own -- the group
set and sums are order-free facts of the inputs, and the GROUP BY semantics underneath are the ones pinned in
tests/golden/sql_reference_rows.json.
"""
from __future__ import annotations

import numpy as np


def convert_to_full(dictionary, indexes):
    return [dictionary[int(i)] for i in indexes]


def group_by_sum_count(blocks):
    """blocks: iterable of (dictionary, indexes ndarray, values ndarray[int64] or None) -> {key value: (sum mod 2^64 as int64, count)}"""
    out = {}
    for dictionary, indexes, values in blocks:
        n = len(indexes)
        vals = np.zeros(n, dtype=np.int64) if values is None else np.asarray(values)
        sums = np.zeros(len(dictionary), dtype=np.uint64)
        cnts = np.zeros(len(dictionary), dtype=np.uint64)
        np.add.at(sums, indexes.astype(np.int64), vals.astype(np.int64).view(np.uint64))
        np.add.at(cnts, indexes.astype(np.int64), 1)
        for pos in np.nonzero(cnts)[0]:
            k = dictionary[int(pos)]
            s, c = out.get(k, (np.uint64(0), 0))
            out[k] = (np.uint64((int(s) + int(sums[pos])) & (2**64 - 1)), c + int(cnts[pos]))
    return {k: (int(np.array([s], dtype=np.uint64).view(np.int64)[0]), c) for k, (s, c) in out.items()}


def dictionary_encode(values):
    """ColumnUnique::uniqueInsertRangeFrom (src/Columns/ColumnUnique.h:520-620) over a full String column: every row gets the
    position of its value in a dictionary that holds each distinct value once, in order of first appearance.
    -> (ids ndarray[uint32], dictionary list, first_rows ndarray[uint64])"""
    pos, d, first = {}, [], []
    ids = np.empty(len(values), dtype=np.uint32)
    for i, v in enumerate(values):
        k = pos.get(v)
        if k is None:
            k = len(d)
            pos[v] = k
            d.append(v)
            first.append(i)
        ids[i] = k
    return ids, d, np.array(first, dtype=np.uint64)


def string_filter(offsets: np.ndarray, chars: np.ndarray, filt: np.ndarray):
    """filterArraysImpl<UInt8> for a ColumnString (src/Columns/ColumnsCommon.cpp:191-286): result offsets and chars hold the values
    whose filter byte is non-zero, in order; SIZES_OF_COLUMNS_DOESNT_MATCH when the sizes differ (:199-200)."""
    if filt.shape[0] != offsets.shape[0]:
        raise ValueError("SIZES_OF_COLUMNS_DOESNT_MATCH")
    res_offsets, res_chars, pos, begin = [], bytearray(), 0, 0
    raw = chars.tobytes()
    for i in range(offsets.shape[0]):
        end = int(offsets[i])
        if filt[i]:
            res_chars += raw[begin:end]
            pos += end - begin
            res_offsets.append(pos)
        begin = end
    return np.array(res_offsets, dtype=np.uint64), np.frombuffer(bytes(res_chars), dtype=np.uint8)
