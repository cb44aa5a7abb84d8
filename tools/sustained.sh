#!/bin/bash
# tools/sustained.sh: does the kernel hold its rate? the default step count (10), then 100 and 1000 steps over the same resident batches,
# with the clocks and the power the card reports sampled beside the longest run
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for s in 10 100 1000; do
    if [ $s = 1000 ]; then (for i in 1 2 3 4 5 6 7 8; do sleep 0.4; /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|memory)" | tr -s ' ' | tr '\n' ';'; echo; done) > gpurun_out/sustained_smi.txt 2>&1 & fi
    python3 bench.py --no-cpu-baseline --no-e2e --steps $s --warmup 2 --distinct-batches 10 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('steps %5d  %.4g reads/s  %.3f ms/step  kernel %.3f ms' % (j['steps'], j['value'], j['ms_per_step'], r['avg_launch_ms']))"
    wait
done
cat gpurun_out/sustained_smi.txt
