#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (rocprofv3 --kernel-trace --stats + separate --pmc
passes, scripts/profile_gpu.sh / profile_pmc2.sh) into profiles/<name>_kernel_stats.csv and
profiles/<name>_counters.json (per-dispatch averages per kernel)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{name}_kernel_stats.csv"))
out = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(src, "pmc*", "*", "*_counter_collection.csv"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "sai2b" not in k:
            continue
        for c, v in cs.items():
            out[k][c] = {"avg_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
json.dump(out, open(os.path.join(dst, f"{name}_counters.json"), "w"), indent=1, sort_keys=True)
print("wrote", name, "kernels:", list(out))
