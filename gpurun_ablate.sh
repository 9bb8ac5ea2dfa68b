cd /tmp && export TMPDIR=/tmp
for n in 1 2; do
  export CHGPU_LIB=$GRAFT_REPO_ROOT/clickhouse_amd/libchgpu_ab$n.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pab_$n -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py groupby 1000000000 > /tmp/ab_$n.log 2>&1 || { echo "variant $n failed"; tail -5 /tmp/ab_$n.log; }
  python3 -c "
import csv,glob
f=glob.glob('/tmp/pab_$n/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_agg_part_lds' in r['Name']: print('ablate $n', r['Name'][:30], round(float(r['MaxNs'])/1e6,2), 'ms max', round(float(r['MinNs'])/1e6,2), 'min')
"
done
