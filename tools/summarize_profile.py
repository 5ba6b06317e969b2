#!/usr/bin/env python3
"""Copies the judged summaries of a gpurun_out/<tag>/ profile run (tools/gpu_profile.sh) into profiles/<tag>/."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = os.path.join("gpurun_out", tag), os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
shutil.copyfile(os.path.join(src, "rocprof_trace", "bench_kernel_stats.csv"), os.path.join(dst, "bench_n30_kernel_stats.csv"))
shutil.copyfile(os.path.join(src, "bench_n30.json"), os.path.join(dst, "bench_n30.json"))
shutil.copyfile(os.path.join(src, "probe_1q_n30.jsonl"), os.path.join(dst, "probe_1q_n30.jsonl"))
out = {"note": "rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE; command: python3 bench.py --steps 1 --warmup 0 "
               "--no-cpu-baseline (n=30, depth 1000, fuse 3). Counter unit KiB. gfx950 correction per "
               "/opt/skills/guides/MI355X_MICROARCH.md HBM section: FETCH_SIZE reports 1/2 of wide coalesced reads -> doubled.",
       "kernels": {}}
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(os.path.join(src, f"rocprof_{name}", "bench_counter_collection.csv"))))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "qsim::" not in k:
            continue
        short = k.split("(")[0].replace("void ", "")
        d = out["kernels"].setdefault(short, {})
        d[ctr + "_KiB_mean"] = sum(v) / len(v)
        d["dispatches_" + name] = len(v)
for d in out["kernels"].values():
    f = d.get("FETCH_SIZE_KiB_mean", 0) * 1024 * 2
    w = d.get("WRITE_SIZE_KiB_mean", 0) * 1024
    d["hbm_read_bytes_per_launch_corrected"] = f
    d["hbm_write_bytes_per_launch"] = w
    d["hbm_traffic_bytes_per_launch"] = f + w
json.dump(out, open(os.path.join(dst, "bench_n30_pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: v["hbm_traffic_bytes_per_launch"] for k, v in out["kernels"].items()}, indent=1))
