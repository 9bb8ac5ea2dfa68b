#!/bin/bash
# The round's evidence in ONE gpurun call (run from the repo root on the MI355X box):
#   gpurun --timeout 1200 -- 'bash tools/gpu_round_end.sh r03'
# full -m gpu suite -> default bench line -> profiles/collect.sh (kernel stats + PMC passes) -> the C5 plan's PMC passes.  Steps are joined
# with && so nothing runs on the card after a failed or killed step.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python -m pytest tests -q -m gpu -x > $OUT/${TAG}_gputest.log 2>&1 && tail -2 $OUT/${TAG}_gputest.log &&
timeout -k 10 300 python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err && tail -c 600 $OUT/${TAG}_bench_n1.json &&
bash profiles/collect.sh $TAG &&
bash tools/gpu_pmc_c5.sh $TAG
