# HBM traffic of the SSB plan (C5) on its own: FETCH_SIZE and WRITE_SIZE passes of `bench.py --only-c5` (separate runs, --kernel-trace only),
# summed over the plan's kernels and divided by the number of plan runs.  Run on the GPU box:  gpurun -- 'bash tools/gpu_pmc_c5.sh r03'
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c5_fetch /tmp/c5_write
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --only-c5"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/c5_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_c5_pmc_bench.json 2> $OUT/${TAG}_c5_pmc.err || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/c5_write -- python3 $ROOT/bench.py $ARGS > /dev/null 2>> $OUT/${TAG}_c5_pmc.err || exit 1
python3 - $TAG <<'PY'
import csv, glob, json, os, sys
tag = sys.argv[1]
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
def per_kernel(d, counter):
    vals = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") == counter:
                vals.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return vals
fetch, write = per_kernel("/tmp/c5_fetch", "FETCH_SIZE"), per_kernel("/tmp/c5_write", "WRITE_SIZE")
runs = 8  # bench.py: three warm-up passes + five timed ones of the plan
skip = ("k_filter_sum<long",)  # the headline steps of the same command, not part of the plan
res = {"tag": tag, "plan_runs": runs, "kernels": {}}
tot_r = tot_w = 0.0
for name in sorted(set(fetch) | set(write)):
    short = name[5:] if name.startswith("void ") else name
    if not short.startswith("k_") or any(s in short for s in skip):
        continue
    r = sum(fetch.get(name, [])) * 1024 * 2   # KiB, and the gfx950 correction for wide streaming reads (MI355X_MICROARCH.md, HBM section)
    w = sum(write.get(name, [])) * 1024
    tot_r += r
    tot_w += w
    res["kernels"][short.split("(")[0][:100]] = {"launches": max(len(fetch.get(name, [])), len(write.get(name, []))), "read_bytes_per_run": r / runs, "write_bytes_per_run": w / runs}
res["C5_hbm_read_bytes_per_run"] = tot_r / runs
res["C5_hbm_write_bytes_per_run"] = tot_w / runs
res["C5_hbm_bytes_per_run"] = (tot_r + tot_w) / runs
res["algorithmic_bytes"] = 24.0 * 750_000_000
res["note"] = "FETCH_SIZE counts L2 -> fabric requests, Infinity Cache hits included; the x2 correction is calibrated for 16-byte streaming loads"
json.dump(res, open(os.path.join(out, f"{tag}_traffic_ssb.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
top = sorted(res["kernels"].items(), key=lambda kv: -(kv[1]["read_bytes_per_run"] + kv[1]["write_bytes_per_run"]))[:8]
for k, v in top:
    print(f"{k[:70]:72s} r {v['read_bytes_per_run'] / 1e9:6.2f} GB  w {v['write_bytes_per_run'] / 1e9:6.2f} GB")
PY
