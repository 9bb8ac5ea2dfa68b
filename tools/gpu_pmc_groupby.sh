# instruction-mix counters of the GROUP BY kernels (run on the GPU box via gpurun); --pmc only with --kernel-trace
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py groupby 1000000000 > /dev/null 2> /tmp/pmc_$tag.err || { echo "set $set failed"; tail -3 /tmp/pmc_$tag.err; continue; }
  python3 - <<PY
import csv,glob
v={}
for p in glob.glob('/tmp/pmc_$tag/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        n=r['Kernel_Name']
        if 'k_gb_scatter' in n or 'k_agg_part' in n or 'k_gb_hist' in n:
            v.setdefault((n[:34], r['Counter_Name']),[]).append(float(r['Counter_Value']))
for (k,c),x in sorted(v.items()): print(f"{k:36s} {c:22s} max {max(x):.4g}  per 64-row group of 1e9 rows {max(x)/15625000:.1f}")
PY
done
