#!/usr/bin/env python3
"""Turn the rocprofv3 kernel stats of tools/bench_kernels.py into a per-kernel roofline table (markdown).
usage: kernel_roofline.py <kernel_stats.csv> <rows>"""
import csv
import sys

path, rows = sys.argv[1], int(sys.argv[2])
R = rows
# algorithmic bytes per launch of the calls bench_kernels.py makes (see that script), keyed by a substring of the kernel name
ALG = [
    ("k_filter_sum<long, 2, true, false, IntRangePred>", 8 * R, "filter+sum, 1 column (8 B/row)"),
    ("k_filter_sum<long, 2, false, false, IntRangePred>", 16 * R, "filter+sum, predicate and value columns (16 B/row)"),
    ("k_expr_filter_sum<unsigned int>", 16 * R, "fused 2-column expression over 2R UInt32 rows"),
    ("k_cmp_mask<long", 9 * R, "compare -> UInt8 mask (8+1 B/row)"),
    ("k_count_nonzero", 1 * R, "countBytesInFilter (1 B/row; a 0.4 GB input at R = 4e8: ramp-up dominated)"),
    ("k_selector", 12 * R, "CRC32-C shard selector (8 B key in, 4 B out)"),
    ("k_weak_hash32", 16 * R, "getWeakHash32 (8 B key + 4 B hash in, 4 B out)"),
    ("k_part_hist_lds", 4 * R, "partition histogram (4 B selector)"),
    ("k_part_scatter_lds", 36 * R, "stable 8-way partition of 2 Int64 columns, LDS-staged (4 + 2*(8+8) B/row)"),
    ("k_index<unsigned long, unsigned long>", (8 + 8 + 8) * (R // 4), "random gather of R/4 rows (8 idx + 8 data + 8 out)"),
    ("k_filter_scatter<unsigned long, true>", (8 + 1 + 8 * 0.1) * R, "materialising filter of one Int64 column at 10 % (8 + 1 + 0.8 B/row), LDS-compacted stores (round 2)"),
    ("k_rp_tilesort<12288u, unsigned int", (12 + 12) * R, "GROUP BY tile-sorted partition (round 2): 4 B key + 8 B value in, 12 B record out"),
    ("k_agg_tiles_lds<unsigned int", 12 * R, "GROUP BY gather of the tiles' runs + LDS aggregation (round 2), 12 B/row, 1 M groups"),
    ("k_gb_hist_wide<unsigned int>", 4 * R, "GROUP BY partition histogram (4 B key)"),
    ("k_gb_scatter<12288u, unsigned int, true>", (12 + 12) * R, "GROUP BY partition scatter (12 B in, 12 B out)"),
    ("k_agg_part_lds<unsigned int, 8", 12 * R, "GROUP BY LDS aggregation of partitions (12 B/row), 1 M groups"),
    ("k_agg_part_lds<unsigned long, 8", 16 * R, "GROUP BY LDS-staged over the source columns (RANGE mode, 16 B/row), 1000 groups"),
    ("k_join_probe_filter<true>", 5 * R, "filter-only LEFT SEMI probe, dense prefilter (4 B key in, 1 B out)"),
    ("k_join_probe_filter_lds", 5 * R, "filter-only LEFT SEMI probe, key set staged in LDS slices (round 2; 4 B key in, 1 B out per pass)"),
    ("k_join_insert", 8 * 10_000_000, "join build: insert 1e7 keys (8 B/row)"),
    ("k_join_fill", 12 * 10_000_000, "join build: CSR fill"),
    ("k_join_probe_count", 8 * (R // 4), "join probe: lookup (8 B/row in)"),
    ("k_join_emit", 4 * (R // 4), "join probe: emit at a 0.5 % match rate (4 B count per left row; value/offset only for matches)"),
]
stats = {}
for r in csv.DictReader(open(path)):
    stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]), float(r["MaxNs"]))
print(f"| kernel | what (rows R = {R:.3g}) | max duration | algorithmic GB/s | ÷ 8000 |")
print("|---|---|---|---|---|")
for key, nbytes, what in ALG:
    m = [(n, v) for n, v in stats.items() if key in n]
    if not m:
        continue
    name, (calls, avg, mx) = m[0]
    gbs = nbytes / mx  # bytes / ns = GB/s ; max duration = the full-size call
    print(f"| `{key.split('(')[0]}` | {what} | {mx / 1e6:.3f} ms | {gbs:.0f} | {gbs / 8000:.2f} |")
