# the parity tests of the hot-path operators under each plan-level A/B switch (the older plans stay reachable and serve other shapes).
# Run on the GPU box:  gpurun --timeout 1200 -- 'bash tools/gpu_ab_parity.sh'
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
T="tests/test_gpu_round2.py tests/test_gpu_agg_join.py tests/test_gpu_filter.py tests/test_gpu_ssb.py"
for v in CHGPU_TUNE_GB_NO_TILED CHGPU_TUNE_GB_NO_AOS CHGPU_TUNE_JOIN_NO_RADIX CHGPU_TUNE_JOIN_NO_LDS_PROBE CHGPU_TUNE_JOIN_NO_SLICE_BUILD CHGPU_TUNE_JOIN_NO_LDS_FILTER CHGPU_TUNE_JOIN_NO_LDS_FILTER_MULTI CHGPU_TUNE_JOIN_NO_DENSE_PREFILTER CHGPU_TUNE_FILTER_NO_STAGED CHGPU_TUNE_FILTER_NO_MULTI; do
  echo "== $v=1"
  env $v=1 timeout -k 10 400 python3 -m pytest $T -x -q -m gpu 2>&1 | tail -2 || exit 1
done
echo "all switches green"
