#!/bin/bash
# phase timers of the both-strands lane kernels (tools/build_variant.sh tbs -DUTREE_LANES_TIMERS -DUTREE_LANES_TIMERS_BS=1) on config 3 (pieces of long reads) and config 2 + RC
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r04b
export UTREE_AMD_SO=$R/utree_amd/libexp_tbs.so
{
echo "== config 3 (72 M nodes, lognormal 1-100 kb, both strands, 400 k reads per launch) =="
python3 bench.py --no-cpu-baseline --no-e2e --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 400000 --len-dist lognormal --model-reads 2000 --steps 4 --warmup 2 2>&1 >/dev/null | grep -A7 "lanes phase timers"
echo "== config 2, both strands, 16 M reads per launch =="
python3 bench.py --no-cpu-baseline --no-e2e --rc 1 --steps 4 --warmup 2 2>&1 >/dev/null | grep -A7 "lanes phase timers"
} | tee gpurun_out/r04b/lanes_phase_timers_both_strands.txt
