# kernel-by-kernel trace of one hint-less 1e9-row / 1 M-group GROUP BY (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/nohint.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import clickhouse_amd as ch
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
ctx = ch.Context(0, st.cuda_stream)
rows, groups = 1_000_000_000, int(os.environ.get("GROUPS", "1000000"))
g = torch.Generator(device=dev).manual_seed(2)
k = torch.randint(0, groups, (rows,), dtype=torch.int32, device=dev, generator=g)
v = torch.randint(-2**31, 2**31, (rows,), dtype=torch.int64, device=dev, generator=g)
kc = ctx.wrap(k.data_ptr(), np.uint32, rows, keepalive=k); vc = ctx.wrap(v.data_ptr(), np.int64, rows, keepalive=v)
for it in range(3):
    a = ch.Aggregator(np.uint32, [(ch.AGG_SUM, np.int64), (ch.AGG_COUNT, None)], size_hint=int(os.environ.get("HINT", "0")), ctx=ctx)
    a.execute_on_block(kc, [vc, None]); n = len(a); a.close()
ctx.synchronize(); print("groups", n)
PY
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_nh -- python3 /tmp/nohint.py > /tmp/nh.log 2>&1 || tail -5 /tmp/nh.log
python3 - <<'PY'
import csv, glob, os
f = glob.glob('/tmp/p_nh/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if r['Kernel_Name'].startswith(('k_', 'void k_'))]
# last iteration = after the last k_agg_part_lds-with-small... just print the final 24 kernels
t_end_prev = None
for r in rows[-int(os.environ.get("LAST", "26")):]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - t_end_prev) / 1e3 if t_end_prev else 0
    print(f"{r['Kernel_Name'][:48]:48s} {(e-s)/1e6:8.3f} ms   gap before {gap:8.1f} us")
    t_end_prev = e
PY
