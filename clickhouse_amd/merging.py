"""MergingAggregatedMemoryEfficientTransform over the C ABI (src/Processors/Transforms/MergingAggregatedMemoryEfficientTransform.h:17-57,
.cpp:33-330): the initiator's side of a distributed GROUP BY.  Several sources hand over blocks of PARTIAL states,
each either one "unsplit" block (bucket_num = -1) or "split" two-level blocks in increasing bucket_num (not every bucket need be
present), optionally one block of overflows; the transform groups the blocks of one bucket from all sources
(GroupingAggregatedTransform), merges them (MergingAggregatedBucketTransform -> Aggregator::mergeBlocks) and emits the merged blocks in
increasing bucket_num (SortingAggregatedTransform), holding only the buckets no source has passed yet.

The contract is the reference's; the work is done in bulk: the blocks of every bucket that has become complete are merged into ONE
device aggregator (buckets are disjoint key sets) and the result is cut into its buckets by the bucket hash
(chgpu_partition_by_hash: (crc32c(key) >> 24) & 0xFF, TwoLevelHashTable.h:53) -- one table build and one partition instead of up to 256
small ones.  An unsplit block that meets split ones is split the same way (Aggregator::convertBlockToTwoLevel, Aggregator.cpp:3300-3409).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _capi as K

NUM_BUCKETS = 256


@dataclass
class AggregatedBlock:
    """a block with BlockInfo{bucket_num, is_overflows} (src/Core/BlockInfo.h:21-29): keys + one column per aggregate (final) or the
    state words (not final: sum / count one word, avg numerator and denominator)"""
    bucket_num: int
    is_overflows: bool
    keys: object
    columns: list
    rows: int


class LogicalError(RuntimeError):
    """the reference's ErrorCodes::LOGICAL_ERROR for a source that breaks the block-order convention"""


class MergingAggregatedMemoryEfficientTransform:
    def __init__(self, key_dtype, aggs, num_inputs: int, final: bool = True, ctx=None, engine=None):
        """engine: the per-GPU operators (clickhouse_amd.distributed.LocalEngine over `ctx` by default; the CPU tests pass an engine
        built on the oracle, so the block-order logic runs without a GPU)"""
        if engine is None:
            from .columns import Context
            from .distributed import LocalEngine
            engine = LocalEngine(ctx if ctx is not None else Context(0))
        self.e = engine
        self.key_dtype = np.dtype(key_dtype)
        self.aggs = list(aggs)
        self.n_words = sum(2 if k == K.AGG_AVG else 1 for k, _ in self.aggs)
        self.final = final
        self.num_inputs = num_inputs
        self.last_bucket_number = [-1] * num_inputs   # GroupingAggregatedTransform::last_bucket_number
        self.finished = [False] * num_inputs
        self.chunks_map: dict[int, list] = {}          # bucket -> [(keys, words, rows)]
        self.single_level_chunks: list = []
        self.overflow_chunks: list = []
        self.has_two_level = False
        self.next_bucket_to_push = 0
        self.done = False

    # ---- input side -------------------------------------------------------------------------------------------------------------
    def add_chunk(self, input_num: int, keys, state_words, bucket_num: int = -1, is_overflows: bool = False):
        """GroupingAggregatedTransform::addChunk (.cpp:258-291): state_words = the word columns of chgpu_agg_export_states(_two_level)"""
        if self.finished[input_num]:
            raise LogicalError(f"input {input_num} sent a block after it had finished")
        kcol = self.e.to_column(keys, self.key_dtype)
        words = [self.e.to_column(w, None) for w in state_words]
        if len(words) != self.n_words:
            raise ValueError(f"{len(words)} state columns for {self.n_words} state words")
        rows = self.e.rows(kcol)
        if rows == 0:
            return
        if is_overflows:
            self.overflow_chunks.append((kcol, words, rows))
        elif bucket_num < 0:
            self.single_level_chunks.append((kcol, words, rows))
        else:
            if bucket_num >= NUM_BUCKETS:
                raise LogicalError(f"bucket_num {bucket_num} out of range")
            if bucket_num < self.last_bucket_number[input_num]:
                raise LogicalError(f"input {input_num}: bucket {bucket_num} after bucket {self.last_bucket_number[input_num]} "
                                   "(split blocks must arrive in the order of bucket_num)")
            if bucket_num < self.next_bucket_to_push:
                raise LogicalError(f"bucket {bucket_num} arrives after it was merged and pushed")
            self.chunks_map.setdefault(bucket_num, []).append((kcol, words, rows))
            self.has_two_level = True
            self.last_bucket_number[input_num] = bucket_num

    def add_serialized_chunk(self, input_num: int, keys, state_bytes, bucket_num: int = -1, is_overflows: bool = False):
        """the same from the wire form: one UInt8 column per aggregate function holding the rows' serialized states one after the other
        (SerializationAggregateFunction; chgpu_agg_deserialize_states)"""
        kcol = self.e.to_column(keys, self.key_dtype)
        rows = self.e.rows(kcol)
        words = []
        for (kind, _), data in zip(self.aggs, state_bytes):
            w0, w1 = self.e.deserialize_states(kind, self.e.to_column(data, np.uint8), rows)
            words.append(w0)
            if w1 is not None:
                words.append(w1)
        self.add_chunk(input_num, kcol, words, bucket_num, is_overflows)

    def finish_input(self, input_num: int):
        self.finished[input_num] = True

    # ---- output side ------------------------------------------------------------------------------------------------------------
    def _convert_single_level(self):
        """work() (.cpp:293-318): unsplit blocks become split ones as soon as any source is two-level"""
        for kcol, words, rows in self.single_level_chunks:
            parts, counts = self.e.partition_by_hash(kcol, [kcol] + words, NUM_BUCKETS)
            begin = 0
            for b in range(NUM_BUCKETS):
                c = int(counts[b])
                if c:
                    if b < self.next_bucket_to_push:
                        raise LogicalError(f"an unsplit block holds bucket {b}, which was already merged and pushed")
                    self.chunks_map.setdefault(b, []).append((self.e.cut(parts[0], begin, c), [self.e.cut(p, begin, c) for p in parts[1:]], c))
                begin += c
        self.single_level_chunks = []

    def _merge(self, chunks):
        """MergingAggregatedBucketTransform::transform -> Aggregator::mergeBlocks(blocks, final) over everything in `chunks`"""
        agg = self.e.Aggregator(self.key_dtype, self.aggs, size_hint=sum(r for _, _, r in chunks))
        for kcol, words, rows in chunks:
            self.e.agg_merge_states(agg, kcol, words)
        return self.e.agg_finalize(agg) if self.final else self.e.agg_export(agg)

    def pull(self):
        """-> the merged blocks that are complete now, in increasing bucket_num; after every input has finished: the rest, then the
        unsplit result (bucket_num -1, only if no source was two-level), then the overflows block (tryPushTwoLevelData /
        tryPushSingleLevelData / tryPushOverflowData, .cpp:33-92)"""
        out: list[AggregatedBlock] = []
        if self.done:
            return out
        all_finished = all(self.finished)
        if self.has_two_level and self.single_level_chunks:
            self._convert_single_level()
        if self.has_two_level:
            # a bucket is complete when no unfinished source can still send a block of it: sources may send several blocks of one bucket
            # (expect_several_chunks_for_single_bucket_per_source, .cpp:117-123), so the bucket a source is AT is not complete yet
            current = NUM_BUCKETS if all_finished else min(self.last_bucket_number[i] for i in range(self.num_inputs) if not self.finished[i])
            ready = sorted(b for b in self.chunks_map if b < current)
            if ready:
                chunks = [c for b in ready for c in self.chunks_map.pop(b)]
                keys, cols, n = self._merge(chunks)
                parts, counts = self.e.partition_by_hash(keys, [keys] + list(cols), NUM_BUCKETS)
                begin = 0
                for b in range(NUM_BUCKETS):
                    c = int(counts[b])
                    if c:
                        if b not in ready:
                            raise LogicalError(f"a block declared as bucket(s) {ready} holds keys of bucket {b}")
                        out.append(AggregatedBlock(b, False, self.e.cut(parts[0], begin, c), [self.e.cut(p, begin, c) for p in parts[1:]], c))
                    begin += c
            self.next_bucket_to_push = max(self.next_bucket_to_push, min(current, NUM_BUCKETS))
        if all_finished:
            if not self.has_two_level and self.single_level_chunks:
                keys, cols, n = self._merge(self.single_level_chunks)
                self.single_level_chunks = []
                out.append(AggregatedBlock(-1, False, keys, cols, n))
            if self.overflow_chunks:
                keys, cols, n = self._merge(self.overflow_chunks)
                self.overflow_chunks = []
                out.append(AggregatedBlock(-1, True, keys, cols, n))
            self.done = True
        return out
