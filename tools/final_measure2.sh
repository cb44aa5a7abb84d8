#!/bin/bash
# second half of the round's measurement batch: hit-dense workload (plain + under the kernel trace), the final default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 tests/scale/hit_dense.py --rc 0 > gpurun_out/hit_dense_fwd.json 2> gpurun_out/hit_dense_fwd.err; echo "hit-dense fwd rc=$?"
python3 tests/scale/hit_dense.py --rc 1 > gpurun_out/hit_dense_rc.json 2> gpurun_out/hit_dense_rc.err; echo "hit-dense rc rc=$?"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hit_dense -- python3 $R/tests/scale/hit_dense.py --rc 1 --sample 20000 > $R/gpurun_out/hit_dense_rc_traced.json 2> $R/gpurun_out/hit_dense_rc_traced.err); echo "traced rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("/root/repo/gpurun_out/prof_hit_dense/*/*_kernel_stats.csv")
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "classify" in r["Name"] or "vote_k" in r["Name"] or "route_k" in r["Name"]]
    with open("/root/repo/gpurun_out/hit_dense_kernels.txt", "w") as o:
        for r in rows:
            o.write("%-100s calls %5s  avg %10.0f ns  total %12s ns\n" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]), r["TotalDurationNs"]))
PY
find gpurun_out/prof_hit_dense -name "*.csv" -size +2M -delete
python3 bench.py > gpurun_out/bench_r02_n1.json 2> gpurun_out/bench_r02_n1.err; echo "bench rc=$?"
