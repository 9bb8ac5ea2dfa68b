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
    for path in glob.glob(os.path.join(out, dirname, "**", "*counter_collection.csv"), recursive=True):
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
try:
    bj = json.loads(open(os.path.join(out, f"prof_{tag}_bench.json")).read().strip().splitlines()[-1])
    res["rows"] = bj["config"]["rows_per_gpu"]
except Exception:
    res["rows"] = None
with open(os.path.join(out, f"{tag}_traffic.json"), "w") as fo:
    json.dump(res, fo, indent=1)
print(json.dumps(res, indent=1)[:3000])
