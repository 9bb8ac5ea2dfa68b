for cfg in "4 4 2" "8 4 2" "2 4 2" "4 8 2" "4 4 3" "4 4 4" "8 8 4"; do set -- $cfg
  echo "== unroll=$1 wg_map=$2 wg_sum=$3"
  CHGPU_TUNE_JIT_UNROLL=$1 CHGPU_TUNE_JIT_WG_MAP=$2 CHGPU_TUNE_JIT_WG_SUM=$3 timeout -k 10 120 python tools/bench_expr_dag.py 1000000000 2>/dev/null | python -c "
import json,sys
for r in json.load(sys.stdin)['results']:
    if 'JIT' in r['case'] and 'multiply' not in r['case']: print('  ', r['case'][:60], round(r['ms'],3))"
done
