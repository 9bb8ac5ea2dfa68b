#!/bin/bash
# round-2 A/B of the GROUP BY / join strategies (run on the GPU box via gpurun); results -> gpurun_out/r02_ab.jsonl
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
: > $OUT/r02_ab.jsonl
run() { # label, what, env...
  local label=$1 what=$2; shift 2
  env "$@" timeout -k 10 240 python3 tools/bench_r02.py $what 1000000000 "$label" >> $OUT/r02_ab.jsonl 2>> $OUT/r02_ab.err || echo "{\"label\": \"$label\", \"failed\": true}" >> $OUT/r02_ab.jsonl
  tail -1 $OUT/r02_ab.jsonl | cut -c1-420
}
run default both X=1
run c3_old_scatter c3 CHGPU_TUNE_GB_OLD_SCATTER=1
run c4_noregions c4 CHGPU_TUNE_JOIN_NO_REGIONS=1
