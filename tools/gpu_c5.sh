#!/bin/bash
# tools/gpu_c5.sh <tag> -- on the MI355X box: chain + SSB parity tests, then the SSB plan (bench.py --only-c5) under rocprofv3 --kernel-trace --stats
set -o pipefail
TAG=${1:-c5}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_chain.py tests/test_gpu_ssb.py -x -q -m gpu > $OUT/tests.log 2>&1
rc=$?
tail -4 $OUT/tests.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c5 -- python3 $ROOT/bench.py --only-c5 --no-cpu-baseline > $OUT/bench_c5.log 2>&1 || { tail -5 $OUT/bench_c5.log; exit 1; }
grep -o '"C5_one_gpu_share".*' $OUT/bench_c5.log | cut -c1-400
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/prof/c5_kernel_stats.csv")))
for r in rows[:24]:
    print(f"{r['Name'][:84]:84s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
