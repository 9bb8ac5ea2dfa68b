#!/bin/bash
# PMC passes over named kernels of tools/bench_r02.py (run on the GPU box via gpurun); --pmc only with --kernel-trace.
# usage: tools/gpu_pmc_kernels.sh <c3|c4> <kernel-name-substring> [more substrings ...]   -> gpurun_out/pmc_kernels/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_kernels
rm -rf $OUT; mkdir -p $OUT
MODE=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for cs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $cs --kernel-trace --output-format csv -d $OUT/p_$i -- python3 $ROOT/tools/bench_r02.py $MODE 1000000000 pmc > /dev/null 2> $OUT/p_$i.err || { echo "pass $cs failed"; tail -3 $OUT/p_$i.err; }
done
python3 - "$@" > $OUT/summary.txt <<PY
import csv, glob, sys
subs = sys.argv[1:]
out = {}
for p in glob.glob('$OUT/p_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        n = r['Kernel_Name']
        if not any(t in n for t in subs):
            continue
        out.setdefault(n.split('(')[0][:70], {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, cs in out.items():
    print(k)
    for c, x in sorted(cs.items()):
        v = sum(x) / len(x)
        print(f"    {c:24s} {v:.4g}" + (f"  = {v*1024/1e9:.2f} GB (x2: {v*2048/1e9:.2f})" if c in ('FETCH_SIZE', 'WRITE_SIZE') else ""))
PY
cat $OUT/summary.txt
