#!/bin/bash
# same-box comparison of lanes-kernel library variants: tools/lanes_var.sh [bench args --] name1 name2 ... ("main" = libutree_amd.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=""
if [[ " $* " == *" -- "* ]]; then while [ "$1" != "--" ]; do ARGS="$ARGS $1"; shift; done; shift; fi
pick='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print("%-10s %.4g reads/s  %.3f ms/step  kernel %.3f ms  %s" % (sys.argv[1], j["value"], j["ms_per_step"], r["avg_launch_ms"], r["kernel"]))'
for n in "$@"; do
    if [ "$n" = main ]; then unset UTREE_AMD_SO; else export UTREE_AMD_SO=$R/utree_amd/libexp_$n.so; fi
    python3 $R/bench.py --no-cpu-baseline --no-e2e $ARGS 2>$R/gpurun_out/lanes_var_$n.err | python3 -c "$pick" $n
done
