# per-kernel times of the C4 join on one GPU (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_jn -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py join 100000000 2>&1 | grep "^join"
python3 -c "
import csv,glob
f=glob.glob('/tmp/p_jn/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['TotalDurationNs'])>2e5: print('   ', r['Name'][:60], r['Calls'], 'calls', round(float(r['AverageNs'])/1e6,3), 'ms avg', round(float(r['MinNs'])/1e6,3), 'min')
"
