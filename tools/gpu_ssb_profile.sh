# C5 (SSB Q4.1-style plan, one GPU's 750 M-row share) with a per-kernel breakdown (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ssb -- python3 $GRAFT_REPO_ROOT/tools/bench_ssb.py 750000000 2000000 2>&1 | tail -3
python3 -c "
import csv,glob
f=glob.glob('/tmp/p_ssb/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if r['Name'].startswith(('k_','void k_'))]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:16]: print('   ', r['Name'][:56], r['Calls'], 'calls', round(float(r['TotalDurationNs'])/1e6/3,3), 'ms per query', round(float(r['AverageNs'])/1e6,3), 'avg')
"
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/p_ssb/**/*kernel_trace.csv', recursive=True)
if f:
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
    rows = [r for r in rows if 'k_join_probe' in r['Kernel_Name'] or 'k_join_insert' in r['Kernel_Name'] or 'k_filter_scatter' in r['Kernel_Name']]
    for r in [r for r in rows if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 300000][-14:]:
        print(r['Kernel_Name'][:24], 'grid', r.get('Grid_Size_X', '?'), round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 3), 'ms')
PY
