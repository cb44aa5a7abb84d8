#!/bin/bash
# same-box A/B of the current tree against the round-2 tree (image version 8) exported to old_v8/ (git archive 5394ba5; its own bench.py and library)
# (recreate it with: mkdir old_v8 && git archive 5394ba5 bench.py utree_amd oracle/orc.py oracle/__init__.py oracle/utree_oracle.c oracle/utree_oracle.h \
#  oracle/utree_build_oracle.c oracle/Makefile include profiles/traffic.json | tar -x -C old_v8 && make -C old_v8/utree_amd/csrc; it is not kept in the tree)
# usage: tools/ab_v8.sh [quick]
R=${GRAFT_REPO_ROOT:-/root/repo}
pick='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; m=r["model"]; print("%-4s %-48s %.4g reads/s  %.3f ms/step  kernel %.3f ms  ovf/read %.3f  %s" % (sys.argv[1], sys.argv[2], j["value"], j["ms_per_step"], r["avg_launch_ms"], m["overflow_buckets_per_read"], j["config"]["workload"].split("image")[1].split(",")[0]))'
run() { # tag dir args
    python3 $2/bench.py --no-cpu-baseline --no-e2e $3 2>/dev/null | python3 -c "$pick" $1 "$3"
}
if [ "$1" = quick ]; then SETS=("" "--kmer 64 --nodes 568000000" "--read-len 250"); else SETS=("" "--rc 1" "--kmer 64 --nodes 568000000" "--nodes 72000000 --read-len 10000 --rc 1 --batch-reads 100000" "--read-len 250"); fi
for ARGS in "${SETS[@]}"; do
    run v9 $R "$ARGS"; run v8 $R/old_v8 "$ARGS"; run v9 $R "$ARGS"; run v8 $R/old_v8 "$ARGS"
done
