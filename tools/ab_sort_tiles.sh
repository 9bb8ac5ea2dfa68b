# A/B of the stable partition's tile size (rows per thread: PL_RPT_V in csrc/partition_kernels.hip) on the radix sort: builds one
# library per variant next to the product library (hipcc is on the GPU box too) and runs tools/bench_sort.py through CHGPU_LIB.
# Run on the GPU box:  gpurun -- 'bash tools/ab_sort_tiles.sh'      (16 rows/thread was chosen with it: 13.1 ms vs 14.7 / 14.1 ms)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
SRC=$ROOT/clickhouse_amd/csrc
FLAGS="-munsafe-fp-atomics -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off"
make -C $SRC > /dev/null || exit 1
for v in 32 16 8; do
  /opt/rocm/bin/hipcc $FLAGS -DPL_RPT_V=$v -c $SRC/partition_kernels.hip -o /tmp/pk_$v.o || exit 1
  OBJS=$(ls $SRC/build/*.o | grep -v partition_kernels)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/pk_$v.o -ldl -o /tmp/libchgpu_rpt$v.so || exit 1
  echo "== $v rows per thread"
  CHGPU_LIB=/tmp/libchgpu_rpt$v.so timeout -k 10 200 python3 $ROOT/tools/bench_sort.py 100000000 2>/dev/null | python3 -c "
import json,sys
for r in json.load(sys.stdin)['results']: print(' ', r['case'], round(r['ms'],2), 'ms')"
done
