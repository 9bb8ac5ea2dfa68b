"""clickhouse_amd — MI355X-native block-processing hot path (filter -> aggregate -> hash join) behind the
reference's column / function / aggregate / join interfaces.  Requires libchgpu.so (HIP, gfx950): no CPU fallback."""
from . import _capi
from ._capi import (AGG_AVG, AGG_COUNT, AGG_SUM, AGG_MIN, AGG_MAX, AGG_ANY, ASOF_LESS, ASOF_GREATER, ASOF_LESS_OR_EQUALS, ASOF_GREATER_OR_EQUALS, EQ, F32, F64, GE, GT, I32, I64, JOIN_FULL, JOIN_INNER, JOIN_LEFT, JOIN_RIGHT, LE, LT, NE,
                    STRICT_ALL, STRICT_ANTI, STRICT_ANY, STRICT_SEMI, I8, I16, U8, U16, U32, U64, ChgpuError)
from ._capi import VAL_COL, VAL_MINUS, VAL_MUL, VAL_PLUS
from .columns import (Column, Context, set_default_option, and_, arith, concat, expr_filter_sum, cmp_const, count_bytes_in_filter, filter_columns, replicate_columns, filter_description_nullable, filter_sum,
                      filter_sum_async, hash_to_selector, pack_fixed_keys, partition_by_hash, sum_add_many,
                      sum_add_many_conditional, unpack_fixed_key, sort_permutation, sort_block, sort_permutation_limit, filter_to_indices)
from .aggregator import Aggregator, NullableKeyAggregator, group_by_min_max, serialize_states, deserialize_states
from .expression import ActionsDAG, ExpressionActions
from .lowcardinality import ColumnString, ColumnLowCardinality, LowCardinalityAggregator, LowCardinalityDictionary, PackedKeysAggregator
from .hashjoin import HashJoin, AsofJoin, join_probe_chain
from .merging import AggregatedBlock, MergingAggregatedMemoryEfficientTransform
from .keysfixed import KeyDict, KeysFixedAggregator, KeysFixedHashJoin, ColumnFixedString, FixedStringAggregator

__all__ = [n for n in dir() if not n.startswith("_")]
