#!/bin/bash
# PMC passes over the GROUP BY partition kernels for two scatter variants (run on the GPU box via gpurun); --pmc only with --kernel-trace
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_scatter
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for variant in "nocarry CHGPU_TUNE_GB_NOCARRY=1" "carry1 CHGPU_TUNE_GB_CARRY=1" "carry2 CHGPU_TUNE_GB_CARRY=2"; do
  set -- $variant; vname=$1; venv=$2
  i=0
  for cs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU"; do
    i=$((i+1))
    export $venv
    timeout -k 10 200 rocprofv3 --pmc $cs --kernel-trace --output-format csv -d $OUT/${vname}_$i -- python3 $ROOT/tools/bench_r02.py c3 1000000000 $vname > /dev/null 2> $OUT/${vname}_$i.err || { echo "pass $vname $cs failed"; tail -3 $OUT/${vname}_$i.err; }
    unset ${venv%%=*}
  done
done
python3 - <<PY
import csv, glob, os, json
out = {}
for d in sorted(glob.glob('$OUT/*_[0-9]')):
    v = os.path.basename(d).rsplit('_', 1)[0]
    for p in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            n = r['Kernel_Name']
            if not any(t in n for t in ('k_gb_scatter', 'k_rp_scatter', 'k_agg_part', 'k_rp_hist')):
                continue
            short = n.split('(')[0][5:60]
            out.setdefault(v, {}).setdefault(short, {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
res = {v: {k: {c: sum(x) / len(x) for c, x in cs.items()} for k, cs in ks.items()} for v, ks in out.items()}
json.dump(res, open('$OUT/summary.json', 'w'), indent=1)
for v, ks in res.items():
    for k, cs in ks.items():
        print(v, k)
        for c, x in sorted(cs.items()):
            extra = f"  = {x*1024/1e9:.2f} GB (x2 for reads: {x*2048/1e9:.2f})" if c in ('FETCH_SIZE', 'WRITE_SIZE') else f"  per 64 rows {x/15625000:.2f}"
            print(f"    {c:24s} {x:.4g}{extra}")
PY
