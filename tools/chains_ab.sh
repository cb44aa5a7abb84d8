#!/bin/bash
# heavy overflow runs as chains (image version 14) against records behind a position directory (UTREE_OVF_CHAINS=0): parity tests, then the
# hit-dense workload both ways on the same box, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_lanes.py tests/test_gpu_configs.py -x -q -k "chains or hit_dense" > $O/chains_tests.log 2>&1 || { tail -30 $O/chains_tests.log; exit 1; }
tail -2 $O/chains_tests.log
B="python3 $R/bench.py --workload hit_dense --no-cpu-baseline --no-e2e"
pick='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print("%-14s %.4g reads/s  %.3f ms/step  kernel %.3f ms  %s" % (sys.argv[1], j["value"], j["ms_per_step"], r["avg_launch_ms"], r["kernel"]))'
for i in 1 2; do
  UTREE_OVF_CHAINS=1 UTREE_TIMING=1 $B 2>$O/hd_chains_$i.err | tee $O/hd_chains_fwd_$i.json | python3 -c "$pick" chains-fwd
  $B 2>/dev/null | tee $O/hd_dir_fwd_$i.json | python3 -c "$pick" dir-fwd
done
UTREE_OVF_CHAINS=1 $B --rc 1 2>/dev/null | tee $O/hd_chains_rc.json | python3 -c "$pick" chains-rc
$B --rc 1 2>/dev/null | tee $O/hd_dir_rc.json | python3 -c "$pick" dir-rc
grep "image:" $O/hd_chains_1.err | tail -5
