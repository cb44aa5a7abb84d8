#!/bin/bash
# lanes kernel: timing (HIP events around lanes + exception pass), phase timers variant, parity subset
R=${GRAFT_REPO_ROOT:-/root/repo}
pick='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print("%-10s %.4g reads/s  %.3f ms/step  kernel %.3f ms  %s" % (sys.argv[1], j["value"], j["ms_per_step"], r["avg_launch_ms"], r["kernel"]))'
for v in 1 0 1; do
  UTREE_LANE_PASS=$v python3 $R/bench.py --no-cpu-baseline --no-e2e "$@" 2>gpurun_out/lanes_q_$v.err | python3 -c "$pick" lanes=$v
done
[ -f $R/utree_amd/libexp_lt.so ] && UTREE_AMD_SO=$R/utree_amd/libexp_lt.so python3 $R/bench.py --no-cpu-baseline --no-e2e "$@" 2>&1 | grep -A6 "lanes phase"
