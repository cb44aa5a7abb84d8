#!/bin/bash
# same-box A/B of library variants (utree_amd/libexp_<name>.so built by tools/build_variant.sh); "main" = libutree_amd.so
# usage: tools/ab.sh [bench args --] name1 name2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=""
if [[ " $* " == *" -- "* ]]; then while [ "$1" != "--" ]; do ARGS="$ARGS $1"; shift; done; shift; fi
for n in "$@"; do
    if [ "$n" = main ]; then unset UTREE_AMD_SO; else export UTREE_AMD_SO=$R/utree_amd/libexp_$n.so; fi
    python3 $R/bench.py --no-cpu-baseline --no-e2e $ARGS 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print("%-14s %.4g reads/s  %.3f ms/step  kernel %.3f ms" % (sys.argv[1], j["value"], j["ms_per_step"], r["avg_launch_ms"]))' $n
done
