#!/bin/bash
# tools/bq.sh [bench args]: one bench.py run without the CPU / file legs, the essentials on one line
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/bench.py --no-cpu-baseline --no-e2e "$@" 2>/tmp/bq_err.txt | python3 -c '
import json,sys
t=sys.stdin.read().strip().splitlines()
if not t: print("NO OUTPUT"); print(open("/tmp/bq_err.txt").read()[-3000:]); sys.exit(1)
j=json.loads(t[-1]); r=j["roofline"]; m=r["model"]
print("%s | %.4g reads/s %.3f ms/step kernel %.3f ms | %s | buckets/read %.2f ovf/read %.3f model %.0f B frac %.3f | %s" % (" ".join(sys.argv[1:]), j["value"], j["ms_per_step"], r["avg_launch_ms"], r["kernel"], m["distinct_buckets_per_read"], m["overflow_buckets_per_read"], r["algorithmic_bytes_per_read"], r["frac"] or 0, j["config"]["workload"].split("image")[1].split(",")[0]))
' "$@"
