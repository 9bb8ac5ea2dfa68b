#!/bin/bash
# profiles/collect.sh <round-tag> — run ON the MI355X box (via gpurun) from the repo root.  Produces under gpurun_out/:
#   prof_<tag>/       rocprofv3 --kernel-trace --stats of the default bench
#   pmc_fetch_<tag>/  rocprofv3 --pmc FETCH_SIZE   (own pass: FETCH_SIZE takes 3 of the 4 TCC slots)
#   pmc_write_<tag>/  rocprofv3 --pmc WRITE_SIZE   (own pass)
# Counters are never combined with trace domains other than --kernel-trace (node-stability rule of this pool).
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/prof_${TAG}_c5
cd /tmp && export TMPDIR=/tmp
# C5 (the SSB plan) reuses the join / GROUP BY kernels of C3 / C4: it is profiled in its own pass so the per-config kernel averages and
# PMC sums of C3 / C4 stay attributable
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-c5"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $ROOT/bench.py $ARGS > $OUT/prof_${TAG}_bench.json 2> $OUT/prof_${TAG}.err || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch_${TAG}.err || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$TAG -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_write_${TAG}.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_c5 -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --only-c5 > $OUT/prof_${TAG}_c5_bench.json 2> $OUT/prof_${TAG}_c5.err || exit 1
cp $(ls -S $OUT/prof_${TAG}_c5/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_c5_kernel_stats.csv
python3 $ROOT/profiles/summarize_pmc.py $TAG
# the per-kernel summary of the --stats pass, for profiles/${TAG}_bench_kernel_stats.csv
cp $(ls -S $OUT/prof_$TAG/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
