#!/bin/bash
# same-box A/B of the current tree against the round-3 tree (image version 10) exported to old_r03/ (its own bench.py and library; recreate with:
#  mkdir old_r03 && git archive dfafbd8 bench.py utree_amd include profiles/traffic.json oracle/orc.py oracle/__init__.py | tar -x -C old_r03 && make -C old_r03/utree_amd/csrc;
#  it is git-ignored)
# usage: tools/ab_r03.sh [quick|rc]
R=${GRAFT_REPO_ROOT:-/root/repo}
pick='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; m=r["model"]; print("%-4s %-60s %.4g reads/s  %.3f ms/step  kernel %.3f ms  ovf/read %.3f  %s" % (sys.argv[1], sys.argv[2], j["value"], j["ms_per_step"], r["avg_launch_ms"], m["overflow_buckets_per_read"], j["config"]["workload"].split("image")[1].split(",")[0]))'
run() { # tag dir args
    python3 $2/bench.py --no-cpu-baseline --no-e2e $3 2>/dev/null | python3 -c "$pick" $1 "$3"
}
if [ "$1" = quick ]; then SETS=("--steps 5" "--rc 1 --steps 5");
elif [ "$1" = rc ]; then SETS=("--rc 1 --steps 5" "--nodes 72000000 --read-len 10000 --rc 1 --batch-reads 400000 --steps 4 --len-dist lognormal --model-reads 1000");
else SETS=("--steps 5" "--rc 1 --steps 5" "--kmer 64 --nodes 568000000 --steps 5" "--kmer 64 --nodes 568000000 --steps 5 --rc 1" "--nodes 72000000 --read-len 10000 --rc 1 --batch-reads 400000 --steps 4 --len-dist lognormal --model-reads 1000" "--read-len 250 --batch-reads 4000000 --steps 5"); fi
for ARGS in "${SETS[@]}"; do
    run r04 $R "$ARGS"; run r03 $R/old_r03 "$ARGS"; run r04 $R "$ARGS"; run r03 $R/old_r03 "$ARGS"
done
