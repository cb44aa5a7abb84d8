#!/bin/bash
# same box, alternating: the round-3 tree (old_r03/), this tree's library, and variant libraries (utree_amd/libexp_<name>.so)
# usage: tools/ab3.sh "<bench args>" name ...     (name: r03 | main | <variant>)
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=$1; shift
pick='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; m=r["model"]; print("%-8s %-40s %.4g reads/s  %.3f ms/step  kernel %.3f ms  ovf/read %.3f  %s" % (sys.argv[1], sys.argv[2], j["value"], j["ms_per_step"], r["avg_launch_ms"], m["overflow_buckets_per_read"], j["config"]["workload"].split("image")[1].split(",")[0]))'
for rep in 1 2; do
for n in "$@"; do
    unset UTREE_AMD_SO
    D=$R
    if [ "$n" = r03 ]; then D=$R/old_r03; elif [ "$n" != main ]; then export UTREE_AMD_SO=$R/utree_amd/libexp_$n.so; fi
    python3 $D/bench.py --no-cpu-baseline --no-e2e $ARGS 2>/dev/null | python3 -c "$pick" $n "$ARGS"
done
done
