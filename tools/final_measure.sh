#!/bin/bash
# the round's measurement batch on one GPU box: profiles of the kernel shapes, CLI-vs-genuine-reference checks at scale through the
# lane-per-read pass (k = 32 +- RC, k = 64, 250 bp reads), hit-dense workload, chain check, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
T=${1:-r02e}
profiles/run_prof.sh ${T}_config2 > gpurun_out/prof_c2.log 2>&1; echo "config2 profile rc=$?"
profiles/run_prof.sh ${T}_config2_rc --rc 1 --steps 5 > gpurun_out/prof_c2rc.log 2>&1; echo "config2 rc profile rc=$?"
profiles/run_prof.sh ${T}_config5_k64 --kmer 64 --nodes 568000000 > gpurun_out/prof_c5.log 2>&1; echo "config5 profile rc=$?"
profiles/run_prof.sh ${T}_config3_long_rc --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 100000 --model-reads 2000 > gpurun_out/prof_c3.log 2>&1; echo "config3 profile rc=$?"
python3 tests/scale/e2e_scale.py --threads 16 > gpurun_out/e2e_scale_lanes_k32.json 2> gpurun_out/e2e_scale_lanes_k32.err; echo "e2e scale k32 rc=$?"
python3 tests/scale/e2e_scale.py --threads 16 --nodes 400000000 --reads 2000000 --rc 1 > gpurun_out/e2e_scale_lanes_k32_rc.json 2> gpurun_out/e2e_scale_lanes_k32_rc.err; echo "e2e scale k32 rc rc=$?"
python3 tests/scale/e2e_scale.py --threads 16 --kmer 64 --nodes 568000000 --reads 2000000 > gpurun_out/e2e_scale_lanes_k64.json 2> gpurun_out/e2e_scale_lanes_k64.err; echo "e2e scale k64 rc=$?"
python3 tests/scale/e2e_scale.py --threads 16 --nodes 400000000 --reads 1000000 --read-len 250 > gpurun_out/e2e_scale_lanes_250bp.json 2> gpurun_out/e2e_scale_lanes_250bp.err; echo "e2e scale 250bp rc=$?"
python3 tests/scale/e2e_scale.py --threads 16 --nodes 72000000 --reads 50000 --read-len 10000 --rc 1 > gpurun_out/e2e_scale_lanes_long_rc.json 2> gpurun_out/e2e_scale_lanes_long_rc.err; echo "e2e scale long rc rc=$?"
python3 tests/scale/e2e_scale.py --threads 16 --kmer 64 --nodes 72000000 --reads 30000 --read-len 5000 > gpurun_out/e2e_scale_lanes_long_k64.json 2> gpurun_out/e2e_scale_lanes_long_k64.err; echo "e2e scale long k64 rc=$?"
python3 bench.py > gpurun_out/bench_${T}_n1.json 2> gpurun_out/bench_${T}_n1.err; echo "bench rc=$?"
