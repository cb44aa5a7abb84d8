#!/bin/bash
# the round's measurement batch on one GPU box: profiles of the three kernel shapes, footprint sweep, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
profiles/run_prof.sh r02_config2 > gpurun_out/prof_c2.log 2>&1; echo "config2 profile rc=$?"
profiles/run_prof.sh r02_config5_k64 --kmer 64 --nodes 568000000 > gpurun_out/prof_c5.log 2>&1; echo "config5 profile rc=$?"
profiles/run_prof.sh r02_config3_long_rc --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 100000 --model-reads 2000 > gpurun_out/prof_c3.log 2>&1; echo "config3 profile rc=$?"
tools/footprint_sweep.sh > gpurun_out/footprint_sweep2.txt 2>&1; echo "sweep rc=$?"
python3 bench.py > gpurun_out/bench_r02_n1.json 2> gpurun_out/bench_r02_n1.err; echo "bench rc=$?"
