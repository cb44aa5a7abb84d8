set -x
cd $GRAFT_REPO_ROOT
tools/lanes_var.sh main d3 nv main > gpurun_out/var2.txt 2>&1
UTREE_AMD_SO=$PWD/utree_amd/libexp_t.so python3 bench.py --no-cpu-baseline --no-e2e > /dev/null 2> gpurun_out/timers_k32.txt
UTREE_AMD_SO=$PWD/utree_amd/libexp_t.so python3 bench.py --no-cpu-baseline --no-e2e --kmer 64 --nodes 568000000 > /dev/null 2> gpurun_out/timers_k64.txt
for t in 5 7 10 12; do UTREE_BUCKET_TARGET=$t tools/bq.sh >> gpurun_out/fill2.txt 2>&1; done
for t in 3 5 6; do UTREE_BUCKET_TARGET=$t tools/bq.sh --kmer 64 --nodes 568000000 >> gpurun_out/fill2.txt 2>&1; done
tools/lanes_prof.sh v9k32 > gpurun_out/prof_v9k32.txt 2>&1
tools/lanes_prof.sh v9k64 --kmer 64 --nodes 568000000 > gpurun_out/prof_v9k64.txt 2>&1
