#!/usr/bin/env python3
"""Summarise a profiles/run_prof.sh output directory (gpurun_out/prof_<tag>) into a small text file."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
out = []
j = json.loads(open(d + "/trace.json").read().strip().splitlines()[-1])
out.append("bench line under rocprofv3 --kernel-trace: value=%.4g reads/s  roofline=%s" % (j["value"], json.dumps(j["roofline"])))
f = glob.glob(d + "/trace/*/*_kernel_stats.csv")[0]
out.append("\n== rocprofv3 --kernel-trace --stats (top kernels) ==")
out.append("%-90s %8s %14s %14s %7s" % ("Name", "Calls", "TotalNs", "AverageNs", "Pct"))
for r in list(csv.DictReader(open(f)))[:8]:
    out.append("%-90s %8s %14s %14.0f %7s" % (r["Name"][:90], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
out.append("\n== PMC counters, average per dispatch (separate --pmc passes) ==")
for tag in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_tcc"):
    fs = glob.glob(d + "/" + tag + "/*/*_counter_collection.csv")
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "classify" in k or "vote_k" in k:
            k = k.split("(")[0][-60:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        out.append("%s %s %s" % (tag, k, {c: x / cnt[(k, c)] for c, x in v.items()}))
print("\n".join(out))
