#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of profiles/collect.sh into gpurun_out/<tag>_traffic.json.

HBM bytes per launch of the dominant kernel, corrected as MI355X_MICROARCH.md §HBM prescribes for gfx950:
  FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports exactly HALF the bytes of a wide (16 B/lane) coalesced
  streaming read, so reads = FETCH_SIZE * 1024 * 2; WRITE_SIZE is exact for 16-B streaming stores.
"""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
out = os.path.join(root, "gpurun_out")


def per_kernel(dirname, counter):
    vals = {}
    paths = glob.glob(os.path.join(out, dirname, "**", "*counter_collection.csv"), recursive=True)
    # only the newest pass: gpurun_out/ accumulates the files of earlier calls next to it
    for path in sorted(paths, key=os.path.getmtime)[-1:]:
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                vals.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return vals


fetch = per_kernel(f"pmc_fetch_{tag}", "FETCH_SIZE")
write = per_kernel(f"pmc_write_{tag}", "WRITE_SIZE")
res = {"tag": tag, "kernels": {}}
for name in sorted(set(fetch) | set(write)):
    if not name.startswith("void k_") and not name.startswith("k_"):
        continue
    f = fetch.get(name, [])
    w = write.get(name, [])
    f_avg = sum(f) / len(f) if f else 0.0
    w_avg = sum(w) / len(w) if w else 0.0
    res["kernels"][name] = {
        "launches": max(len(f), len(w)),
        "FETCH_SIZE_KiB_avg": f_avg,
        "WRITE_SIZE_KiB_avg": w_avg,
        "hbm_read_bytes_per_launch_corrected": f_avg * 1024 * 2,
        "hbm_write_bytes_per_launch": w_avg * 1024,
    }
main = [k for k in res["kernels"] if "k_filter_sum<" in k]
if main:
    k = res["kernels"][main[0]]
    res["k_filter_sum_hbm_bytes_per_launch"] = k["hbm_read_bytes_per_launch_corrected"] + k["hbm_write_bytes_per_launch"]
# the workload the counters were collected on (bench.py only quotes `traffic` for the same row count)
bj = {}
try:
    bj = json.loads(open(os.path.join(out, f"prof_{tag}_bench.json")).read().strip().splitlines()[-1])
    res["rows"] = bj["config"]["rows_per_gpu"]
except Exception:
    res["rows"] = None


def group_bytes(prefixes):
    """HBM bytes moved by every launch of the kernels of one operator, read and write, and the per-kernel split.  A prefix "k_x_" matches
    kernel names that start with it; "k_rp_<Marker" matches the shared partition kernels instantiated with that functor type."""
    tot_r = tot_w = 0.0
    split = {}
    for name, k in res["kernels"].items():
        short = name[5:] if name.startswith("void ") else name
        hit = any(short.startswith(p) for p in prefixes if "<" not in p) or any(all(t in short for t in p.split("<")) for p in prefixes if "<" in p)
        if not hit:
            continue
        r = k["hbm_read_bytes_per_launch_corrected"] * k["launches"]
        w = k["hbm_write_bytes_per_launch"] * k["launches"]
        tot_r += r
        tot_w += w
        label = short.split("(")[0][:110]
        split[label] = {"launches": k["launches"], "read_bytes": r, "write_bytes": w}
    return tot_r, tot_w, split


# configs C3 / C4 of the same command: bytes of their kernel families divided by the number of operator calls the bench made
cfg = bj.get("configs") or {}
if "C3" in cfg and cfg["C3"].get("calls"):
    r, w, split = group_bytes(["k_gb_", "k_agg_", "k_tile_", "k_rp_<GbpPartFn"])
    calls = cfg["C3"]["calls"]
    res["C3_hbm_bytes_per_call"] = (r + w) / calls
    res["C3"] = {"calls": calls, "read_bytes_per_call": r / calls, "write_bytes_per_call": w / calls, "algorithmic_bytes": cfg["C3"]["roofline"]["algorithmic_bytes"],
                 "kernels_total_over_all_calls": split}
if "C4_one_gpu" in cfg and cfg["C4_one_gpu"].get("probe_calls"):
    c4 = cfg["C4_one_gpu"]
    r, w, split = group_bytes(["k_join_", "k_jp_", "k_rp_<JoinRegionFn", "k_rp_<JoinBucket2Fn", "k_rp_<JoinSliceFn", "k_rp_<JoinRadixFn"])
    # build and probe run a different number of times: weigh each kernel by the calls of its phase
    per_call = 0.0
    for name, v in split.items():
        is_build = any(t in name for t in ("k_join_insert", "k_join_fill", "k_join_root", "k_join_finalize", "k_join_stage", "k_join_build_slices", "k_join_iota", "JoinSliceFn"))
        per_call += (v["read_bytes"] + v["write_bytes"]) / (c4["build_calls"] if is_build else c4["probe_calls"])
    res["C4_hbm_bytes_per_call"] = per_call
    res["C4"] = {"build_calls": c4["build_calls"], "probe_calls": c4["probe_calls"], "algorithmic_bytes": c4["roofline"]["algorithmic_bytes"],
                 "kernels_total_over_all_calls": split}
res["source"] = f"profiles/{tag}_traffic.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) of `python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline`, profiles/collect.sh {tag}"
with open(os.path.join(out, f"{tag}_traffic.json"), "w") as fo:
    json.dump(res, fo, indent=1)
# the per-operator views on their own (same numbers)
for name, key in (("groupby", "C3"), ("join", "C4")):
    if key in res:
        with open(os.path.join(out, f"{tag}_traffic_{name}.json"), "w") as fo:
            json.dump({"source": res["source"], "rows": res.get("rows"), f"{key}_hbm_bytes_per_call": res.get(f"{key}_hbm_bytes_per_call"), key: res[key]}, fo, indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1)[:6000])
