#!/bin/bash
# the round's measurement batch on one GPU box: profiles of the kernel shapes with the lane-per-read pass in place, hit-dense workload and chain check on the new
# image version, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
profiles/run_prof.sh r02d_config2 > gpurun_out/prof_c2.log 2>&1; echo "config2 profile rc=$?"
profiles/run_prof.sh r02d_config2_rc --rc 1 --steps 5 > gpurun_out/prof_c2rc.log 2>&1; echo "config2 rc profile rc=$?"
profiles/run_prof.sh r02d_config5_k64 --kmer 64 --nodes 568000000 > gpurun_out/prof_c5.log 2>&1; echo "config5 profile rc=$?"
profiles/run_prof.sh r02d_config3_long_rc --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 100000 --model-reads 2000 > gpurun_out/prof_c3.log 2>&1; echo "config3 profile rc=$?"
python3 tests/scale/hit_dense.py --rc 0 > gpurun_out/hit_dense_fwd_v8.json 2> gpurun_out/hit_dense_fwd_v8.err; echo "hit dense fwd rc=$?"
python3 tests/scale/hit_dense.py --rc 1 > gpurun_out/hit_dense_rc_v8.json 2> gpurun_out/hit_dense_rc_v8.err; echo "hit dense rc rc=$?"
python3 tests/scale/chain_check.py > gpurun_out/chain_check_v8.json 2> gpurun_out/chain_check_v8.err; echo "chain check rc=$?"
python3 bench.py > gpurun_out/bench_r02d_n1.json 2> gpurun_out/bench_r02d_n1.err; echo "bench rc=$?"
