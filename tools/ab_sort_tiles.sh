for v in "" _rpt16 _rpt8; do
  echo "== libchgpu$v.so"
  CHGPU_LIB=$PWD/clickhouse_amd/libchgpu$v.so timeout -k 10 200 python tools/bench_sort.py 100000000 2>/dev/null | python -c "
import json,sys
for r in json.load(sys.stdin)['results']: print(r['case'], round(r['ms'],2))"
done
