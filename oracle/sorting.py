"""TEST INFRASTRUCTURE ONLY — CPU restatement of IColumn::getPermutation for ColumnVector<T> with
PermutationSortStability::Stable (src/Columns/ColumnVector.cpp:245-330).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this.

The permutation is the one std::sort produces with less_stable / greater_stable (:120-166): rows ordered by
CompareHelper<T>::less / greater (FloatCompareHelper for floats: NaN against a number answers from nan_direction_hint, NaN
against NaN is neither less nor greater), ties — a == b, or both NaN — broken by the row number, ascending, in BOTH
directions.  Restated as Python's sorted() (stable) over row numbers with exactly that three-way comparator, which is an
independent route from the product's radix keys.  Multi-column ORDER BY = sortBlock's lexicographic comparator
(src/Interpreters/sortBlock.cpp:33-80), restated the same way.
Parity pinning: NaN placement and direction are PINNED by the 16 output blocks of 03447_float_nan_order (tests/golden/sort_nan_order.json:
ASC / DESC x NULLS FIRST / LAST over 3 and 256 rows); tie-breaking by row number has no reference vector -- a total order with index
tie-break has exactly one valid permutation, so the restated definition determines the expected output completely.
"""
from __future__ import annotations

import functools
import math

import numpy as np


def _compare(a, b, nan_direction_hint):
    """CompareHelper::compare: -1 / 0 / 1"""
    an = isinstance(a, float) and math.isnan(a)
    bn = isinstance(b, float) and math.isnan(b)
    if an and bn:
        return 0
    if an:
        return nan_direction_hint
    if bn:
        return -nan_direction_hint
    return (a > b) - (a < b)


def get_permutation(data: np.ndarray, descending: bool = False, nan_direction_hint: int = 1) -> np.ndarray:
    return sort_block([(data, descending, nan_direction_hint)])


def sort_block(description) -> np.ndarray:
    """description: [(column ndarray, descending, nan_direction_hint), ...] most significant first -> permutation (UInt64)"""
    cols = [(c.tolist(), -1 if desc else 1, hint) for c, desc, hint in description]
    n = len(cols[0][0])

    def cmp(i, j):
        for vals, direction, hint in cols:
            r = _compare(vals[i], vals[j], hint) * direction
            if r:
                return r
        return (i > j) - (i < j)

    return np.array(sorted(range(n), key=functools.cmp_to_key(cmp)), dtype=np.uint64)
