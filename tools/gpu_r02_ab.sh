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
run c4_default c4 X=1
run c4_nofuse c4 CHGPU_TUNE_JOIN_NO_FUSED_PAYLOAD=1
run c4_region2048 c4 CHGPU_TUNE_JOIN_REGION_KIB=2048
run c4_region512 c4 CHGPU_TUNE_JOIN_REGION_KIB=512
