# round-3 evidence: the default bench line with every leg, rocprofv3 passes for the BASELINE configurations (both bucket sizes for configs 2 and 5),
# footprint sweeps, the hit-dense workload under the kernel trace (the profiles of the first two sessions are of 4 M-read launches: said explicitly since the default moved to 16 M)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python3 bench.py > gpurun_out/r03/bench_r03_n1.json 2> gpurun_out/r03/bench_r03_n1.err
profiles/run_prof.sh r03_config2 --batch-reads 4000000 > gpurun_out/r03/p1.log 2>&1
profiles/run_prof.sh r03_config2_rc --rc 1 --batch-reads 4000000 > gpurun_out/r03/p2.log 2>&1
profiles/run_prof.sh r03_config5_k64 --kmer 64 --nodes 568000000 --batch-reads 4000000 > gpurun_out/r03/p3.log 2>&1
profiles/run_prof.sh r03_config3_long_rc --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 100000 > gpurun_out/r03/p4.log 2>&1
profiles/run_prof.sh r03_config3_lognormal_rc --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 100000 --len-dist lognormal > gpurun_out/r03/p5.log 2>&1
UTREE_BUCKET_BYTES=128 profiles/run_prof.sh r03_config2_bucket128 --batch-reads 4000000 > gpurun_out/r03/p6.log 2>&1
UTREE_BUCKET_BYTES=128 profiles/run_prof.sh r03_config5_k64_bucket128 --kmer 64 --nodes 568000000 --batch-reads 4000000 > gpurun_out/r03/p7.log 2>&1
profiles/run_prof.sh r03_hit_dense --workload hit_dense > gpurun_out/r03/p8.log 2>&1
# the long launches (bench.py's default since the round's third session: 16 M reads per launch)
profiles/run_prof.sh r03_config2_16M > gpurun_out/r03/p9.log 2>&1
profiles/run_prof.sh r03_config2_rc_16M --rc 1 > gpurun_out/r03/p10.log 2>&1
profiles/run_prof.sh r03_config5_k64_16M --kmer 64 --nodes 568000000 > gpurun_out/r03/p11.log 2>&1
profiles/run_prof.sh r03_config3_lognormal_rc_400k --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 400000 --len-dist lognormal --model-reads 2000 > gpurun_out/r03/p12.log 2>&1
echo "== 64-byte buckets (default), UTREE_BUCKET_TARGET = nodes per bucket ==" > gpurun_out/r03/footprint_sweep.txt
for t in 2 3 4 5 6; do UTREE_BUCKET_TARGET=$t tools/bq.sh >> gpurun_out/r03/footprint_sweep.txt 2>&1; done
echo "== 128-byte buckets (UTREE_BUCKET_BYTES=128) ==" >> gpurun_out/r03/footprint_sweep.txt
for t in 7 9 11 14; do UTREE_BUCKET_BYTES=128 UTREE_BUCKET_TARGET=$t tools/bq.sh >> gpurun_out/r03/footprint_sweep.txt 2>&1; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/hd_trace -- python3 $GRAFT_REPO_ROOT/tests/scale/hit_dense.py --rc 0 --sample 50000 > $GRAFT_REPO_ROOT/gpurun_out/r03/hit_dense_fwd_traced.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > gpurun_out/r03/hit_dense_fwd_kernel_trace.txt
import csv, glob
for f in glob.glob("gpurun_out/r03/hd_trace/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if any(k in r["Name"] for k in ("classify", "vote_k", "route", "fmt_", "frame_k", "nl_"))]
    for r in rows:
        print("%-110s calls %5s avg %10.1f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/r03/hd_trace
python3 tests/scale/hit_dense.py --rc 0 --sample 200000 > gpurun_out/r03/hit_dense_fwd.json 2>/dev/null
python3 tests/scale/hit_dense.py --rc 1 --sample 200000 > gpurun_out/r03/hit_dense_rc.json 2>/dev/null
UTREE_BUCKET_BYTES=128 python3 tests/scale/hit_dense.py --rc 0 --sample 1000 > gpurun_out/r03/hit_dense_fwd_bucket128.json 2>/dev/null
du -sh gpurun_out; echo finished
