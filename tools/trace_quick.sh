#!/bin/bash
# kernel-trace summary of one bench.py run: tools/trace_quick.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e $* > $OUT/b.json 2> $OUT/e.txt || exit 1
python3 - "$OUT" <<'PY'
import sys, glob, csv
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((float(r["TotalDurationNs"]), r["Name"], int(r["Calls"]), float(r["AverageNs"])))
for tot, name, calls, avg in sorted(rows, reverse=True)[:14]:
    print("%-110s calls %4d avg %9.1f us total %8.2f ms" % (name[:110], calls, avg / 1e3, tot / 1e6))
PY
find $OUT -name "*.csv" -size +1M -delete
