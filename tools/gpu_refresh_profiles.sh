# Refresh the secondary evidence under gpurun_out/ (copy into profiles/ afterwards): per-config numbers and the per-kernel roofline.
# Run on the GPU box:  gpurun -- 'bash tools/gpu_refresh_profiles.sh r01'
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
timeout -k 10 900 python3 $ROOT/tools/bench_configs.py > $OUT/${TAG}_configs.json 2> $OUT/${TAG}_configs.err || { tail -5 $OUT/${TAG}_configs.err; exit 1; }
echo "configs done"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_allk
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_allk -- python3 $ROOT/tools/bench_kernels.py 400000000 > $OUT/${TAG}_allk.log 2>&1 || { tail -5 $OUT/${TAG}_allk.log; exit 1; }
cp $(find /tmp/p_allk -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_all_kernels_stats.csv
python3 $ROOT/tools/kernel_roofline.py $OUT/${TAG}_all_kernels_stats.csv 400000000 > $OUT/${TAG}_kernel_roofline.md
cat $OUT/${TAG}_kernel_roofline.md
