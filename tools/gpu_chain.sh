#!/bin/bash
# tools/gpu_chain.sh <tag> [lib]: tools/bench_chain.py under rocprofv3 --kernel-trace --stats; prints the chain kernels' average times
TAG=${1:-chain}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
[ -n "$2" ] && export CHGPU_LIB=$ROOT/$2
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o ch -- python3 $ROOT/tools/bench_chain.py > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
grep chain_ms $OUT/bench.log
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/prof/ch_kernel_stats.csv")):
    if r['Name'].startswith('k_chain') or r['Name'].startswith('k_join'):
        print(f"$TAG {r['Name'][:40]:40s} calls={r['Calls']:>3s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f}")
PY
