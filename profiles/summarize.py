#!/usr/bin/env python3
"""Summarise a profiles/run_prof.sh output directory (gpurun_out/prof_<tag>) into a small text file (stdout) and a
machine-readable counters.json next to it (read by profiles/make_traffic.py)."""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
out = []
sha = open(d + "/kernel_source_sha256.txt").read().strip() if os.path.exists(d + "/kernel_source_sha256.txt") else None
args = open(d + "/bench_args.txt").read().strip() if os.path.exists(d + "/bench_args.txt") else ""
out.append("bench args: %s   kernel sources sha256: %s" % (args or "(default)", sha))
j = json.loads(open(d + "/trace.json").read().strip().splitlines()[-1])
out.append("bench line under rocprofv3 --kernel-trace: value=%.4g reads/s  roofline=%s" % (j["value"], json.dumps(j["roofline"])))
# (a bench run that starts child processes -- the hit-dense workload runs the product's builder -- leaves one file per process: the one
# that holds the classify kernels is the bench's own)
cands = glob.glob(d + "/trace/*/*_kernel_stats.csv")
f = next((c for c in cands if "classify" in open(c).read()), cands[0])
out.append("\n== rocprofv3 --kernel-trace --stats (top kernels) ==")
out.append("%-110s %8s %14s %14s %7s" % ("Name", "Calls", "TotalNs", "AverageNs", "Pct"))
stats = {}
# the search kernels, then the largest of the rest (database generation and image build of the bench run)
rows = list(csv.DictReader(open(f)))
mine = [r for r in rows if "classify" in r["Name"] or "vote_k" in r["Name"]]
for r in mine + [r for r in rows if r not in mine][:6]:
    out.append("%-110s %8s %14s %14.0f %7s" % (r["Name"][:110], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
    stats[r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
out.append("\n== PMC counters, average per dispatch (separate --pmc passes) ==")
counters = {}
for tag in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_tcc"):
    fs = glob.glob(d + "/" + tag + "/*/*_counter_collection.csv")
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in (row for one in fs for row in csv.DictReader(open(one))):
        k = r["Kernel_Name"]
        if "classify" in k or "vote_k" in k:
            k = k.replace("void ", "", 1).replace("(anonymous namespace)::", "").split("(")[0].strip()
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        avg = {c: x / cnt[(k, c)] for c, x in v.items()}
        out.append("%s %s %s" % (tag, k, avg))
        counters.setdefault(k, {}).update(avg)
json.dump({"kernel_source_sha256": sha, "bench_args": args, "bench_line": j, "kernel_stats": stats, "counters": counters},
          open(d + "/counters.json", "w"), indent=1)
print("\n".join(out))
