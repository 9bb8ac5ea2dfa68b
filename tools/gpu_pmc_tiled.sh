#!/bin/bash
# PMC passes over the tile-sorted GROUP BY kernels (run on the GPU box via gpurun); --pmc only with --kernel-trace.
# usage: tools/gpu_pmc_tiled.sh [ENV=VAL ...]    -> gpurun_out/pmc_tiled/summary.json
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_tiled
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
i=0
for cs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $cs --kernel-trace --output-format csv -d $OUT/p_$i -- python3 $ROOT/tools/bench_r02.py c3 1000000000 pmc > /dev/null 2> $OUT/p_$i.err || { echo "pass $cs failed"; tail -3 $OUT/p_$i.err; }
done
python3 - <<PY
import csv, glob, os, json
out = {}
for d in sorted(glob.glob('$OUT/p_[0-9]')):
    for p in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            n = r['Kernel_Name']
            if not any(t in n for t in ('k_rp_tilesort', 'k_agg_tiles', 'k_rp_scatter', 'k_agg_part', 'k_rp_hist')):
                continue
            short = n.split('(')[0][5:60]
            out.setdefault(short, {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
res = {k: {c: sum(x) / len(x) for c, x in cs.items()} for k, cs in out.items()}
json.dump(res, open('$OUT/summary.json', 'w'), indent=1)
for k, cs in res.items():
    print(k)
    for c, x in sorted(cs.items()):
        extra = f"  = {x*1024/1e9:.2f} GB (x2 for reads: {x*2048/1e9:.2f})" if c in ('FETCH_SIZE', 'WRITE_SIZE') else f"  per 64 rows {x/15625000:.2f}"
        print(f"    {c:28s} {x:.4g}{extra}")
PY
