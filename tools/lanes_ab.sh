#!/bin/bash
# same-box A/B: lane-per-read pass on / off (UTREE_LANE_PASS), config 2 kernel rate
R=${GRAFT_REPO_ROOT:-/root/repo}
pick='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print("%-10s %.4g reads/s  %.3f ms/step  kernel %.3f ms  %s" % (sys.argv[1], j["value"], j["ms_per_step"], r["avg_launch_ms"], r["kernel"]))'
for v in 0 1 0 1; do
  UTREE_LANE_PASS=$v python3 $R/bench.py --no-cpu-baseline --no-e2e "$@" 2>gpurun_out/lanes_ab_$v.err | python3 -c "$pick" lanes=$v
done
