#!/bin/bash
# C3 A/B with per-kernel times (run on the GPU box via gpurun): for each variant, tools/bench_r02.py c3 under rocprofv3 --kernel-trace --stats.
# usage: tools/gpu_c3_ab.sh "label ENV=VAL ENV2=VAL" "label2 ..."      results -> gpurun_out/c3_ab.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/c3_ab.txt
for spec in "$@"; do
  set -- $spec
  label=$1; shift
  rm -rf $OUT/c3ab_$label
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3ab_$label -- python3 $ROOT/tools/bench_r02.py c3 1000000000 $label > $OUT/c3ab_$label.json 2> $OUT/c3ab_$label.err ) || { echo "$label FAILED" >> $OUT/c3_ab.txt; continue; }
  echo "== $label $*" >> $OUT/c3_ab.txt
  cut -c1-300 $OUT/c3ab_$label.json >> $OUT/c3_ab.txt
  python3 - $OUT/c3ab_$label >> $OUT/c3_ab.txt <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"))[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(t in n for t in ("k_rp_", "k_agg_part", "k_gb_", "k_agg_")) and float(r["AverageNs"]) > 20000:
        print("   %-110s calls %3s avg %.3f ms" % (n[:110], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
cat $OUT/c3_ab.txt
