#!/bin/bash
# k = 64, 568 M nodes: sizing of the sub-slices (UTREE_LUMP_SLACK: overflow share allowed, times the design load's) and of the buckets (UTREE_BUCKET_TARGET) -- same box
R=${GRAFT_REPO_ROOT:-/root/repo}
pick='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; m=r["model"]; print("%-28s %.4g reads/s  kernel %.3f ms  ovf/read %.3f  buckets/read %.2f %s" % (sys.argv[1], j["value"], r["avg_launch_ms"], m["overflow_buckets_per_read"], m.get("buckets_per_read", 0), j["config"]["workload"].split("image")[1].split(",")[0]))'
for rep in 1 2; do
for v in "UTREE_LUMP_SLACK=1.5" "UTREE_LUMP_SLACK=1.0" "UTREE_LUMP_SLACK=2.5" "UTREE_LUMP_SLACK=4" "UTREE_LUMP_SLACK=8" "UTREE_BUCKET_TARGET=1.0" "UTREE_BUCKET_TARGET=2.0" "UTREE_SUB_SLICES=0"; do
    env $v python3 $R/bench.py --no-cpu-baseline --no-e2e --kmer 64 --nodes 568000000 --steps 5 $EXTRA 2>/dev/null | python3 -c "$pick" "$v"
done
done
