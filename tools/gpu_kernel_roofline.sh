# per-kernel roofline table of the operator sweep (tools/bench_kernels.py) under rocprofv3.  Run on the GPU box:
#   gpurun -- 'bash tools/gpu_kernel_roofline.sh r02'   then copy gpurun_out/<tag>_kernel_roofline.md and <tag>_all_kernels_stats.csv into profiles/
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_kern
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kern -- python3 $ROOT/tools/bench_kernels.py 400000000 > $OUT/${TAG}_kernels.log 2> $OUT/${TAG}_kernels.err || { tail -5 $OUT/${TAG}_kernels.err; exit 1; }
cp $(find /tmp/p_kern -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_all_kernels_stats.csv
python3 $ROOT/tools/kernel_roofline.py $OUT/${TAG}_all_kernels_stats.csv 400000000 > $OUT/${TAG}_kernel_roofline.md
cat $OUT/${TAG}_kernel_roofline.md
