timeout -k 10 400 python -m pytest tests/test_gpu_agg_join.py tests/test_gpu_full_size.py -m gpu -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_gb -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py groupby 1000000000 2>&1 | grep "^groupby"
python3 -c "
import csv,glob
f=glob.glob('/tmp/p_gb/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_gb' in r['Name'] or 'k_agg' in r['Name']: print('   ', r['Name'][:44], r['Calls'], 'calls', round(float(r['MaxNs'])/1e6,2), 'ms max', round(float(r['MinNs'])/1e6,2), 'min')
"
