# rocprofv3 kernel stats of the §8(f) paths (run-time compiled DAG kernels, LowCardinality remap, radix sort) + a fuzz soak of the
# DAG compiler.  Run on the GPU box:  gpurun -- 'bash tools/gpu_profile_widening.sh r01'   (copy the CSVs into profiles/ afterwards)
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for s in 4; do
  CHGPU_FUZZ_SEED=$s timeout -k 10 300 python3 -m pytest tests/test_expr_dag.py -m gpu -x -q -k random_dags > $OUT/${TAG}_dag_soak_$s.log 2>&1 || { tail -30 $OUT/${TAG}_dag_soak_$s.log; exit 1; }
  tail -1 $OUT/${TAG}_dag_soak_$s.log
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_dag /tmp/p_sort
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dag -- python3 $ROOT/tools/bench_expr_dag.py 1000000000 > $OUT/${TAG}_expr_dag_profiled.json 2> $OUT/${TAG}_expr_dag_profiled.err || { tail -5 $OUT/${TAG}_expr_dag_profiled.err; exit 1; }
cp $(find /tmp/p_dag -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_expr_dag_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_sort -- python3 $ROOT/tools/bench_sort.py 100000000 > $OUT/${TAG}_sort_profiled.json 2> $OUT/${TAG}_sort_profiled.err || { tail -5 $OUT/${TAG}_sort_profiled.err; exit 1; }
cp $(find /tmp/p_sort -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_sort_kernel_stats.csv
rm -rf /tmp/p_lz4 /tmp/p_q31
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_lz4 -- python3 $ROOT/tools/bench_decompress.py 50000000 > $OUT/${TAG}_decompress_profiled.json 2> $OUT/${TAG}_decompress_profiled.err || { tail -5 $OUT/${TAG}_decompress_profiled.err; exit 1; }
cp $(find /tmp/p_lz4 -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_decompress_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_q31 -- python3 $ROOT/tools/bench_ssb_q31.py > $OUT/${TAG}_ssb_q31_profiled.json 2> $OUT/${TAG}_ssb_q31_profiled.err || { tail -5 $OUT/${TAG}_ssb_q31_profiled.err; exit 1; }
cp $(find /tmp/p_q31 -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_ssb_q31_kernel_stats.csv
head -12 $OUT/${TAG}_expr_dag_kernel_stats.csv | cut -c1-160
head -12 $OUT/${TAG}_sort_kernel_stats.csv | cut -c1-160
