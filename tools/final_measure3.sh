#!/bin/bash
# round 2, after the lane-per-read pass: profiles of the kernel shapes, RC, the default bench line, scale checks
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
profiles/run_prof.sh r02c_config2 > gpurun_out/prof_c2.log 2>&1; echo "config2 profile rc=$?"
profiles/run_prof.sh r02c_config2_rc --rc 1 --steps 5 > gpurun_out/prof_c2rc.log 2>&1; echo "config2 rc profile rc=$?"
profiles/run_prof.sh r02c_config5_k64 --kmer 64 --nodes 568000000 > gpurun_out/prof_c5.log 2>&1; echo "config5 profile rc=$?"
profiles/run_prof.sh r02c_config3_long_rc --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 100000 --model-reads 2000 > gpurun_out/prof_c3.log 2>&1; echo "config3 profile rc=$?"
python3 bench.py > gpurun_out/bench_r02c_n1.json 2> gpurun_out/bench_r02c_n1.err; echo "bench rc=$?"
