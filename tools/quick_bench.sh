#!/bin/bash
# kernel-rate checks on the GPU box (resident batches, no CPU / file legs): config 2, RC, k = 64, long reads (config 3 shape)
R=${GRAFT_REPO_ROOT:-/root/repo}
B="python3 $R/bench.py --no-cpu-baseline --no-e2e"
pick='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print("%-10s %.4g reads/s  %.3f ms/step  kernel %.3f ms  %s" % (sys.argv[1], j["value"], j["ms_per_step"], r["avg_launch_ms"], r["kernel"]))'
$B 2>/dev/null | python3 -c "$pick" config2
$B --rc 1 --steps 5 2>/dev/null | python3 -c "$pick" config2+RC
$B --kmer 64 --nodes 568000000 --steps 5 2>/dev/null | python3 -c "$pick" config5-k64
$B --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 100000 --steps 5 --model-reads 1000 2>/dev/null | python3 -c "$pick" config3
