#!/bin/bash
# footprint / speed of the device image on config 2 (same box): table width (fine_bits) and bucket load target
R=${GRAFT_REPO_ROOT:-/root/repo}
pick='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; m=r["model"]; import re; g=re.search(r"image ([0-9.]+) GiB", j["config"]["workload"]).group(1); print("%-22s image %6s GiB  %.4g reads/s  kernel %.3f ms  buckets/read %.2f  overflow buckets/read %.3f" % (sys.argv[1], g, j["value"], r["avg_launch_ms"], m["distinct_buckets_per_read"], m["overflow_buckets_per_read"]))'
for F in 8 7 6 5 4 3; do python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 5 --fine-bits $F 2>/dev/null | python3 -c "$pick" "fine_bits=$F"; done
for T in 3 4 6 8; do UTREE_BUCKET_TARGET=$T python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 5 2>/dev/null | python3 -c "$pick" "bucket_target=$T"; done
