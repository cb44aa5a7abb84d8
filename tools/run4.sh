set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03/t_full_64.log 2>&1; tail -4 gpurun_out/r03/t_full_64.log
tools/ab_v8.sh > gpurun_out/r03/ab3.txt 2>&1
for t in 2 2.5 3 4; do UTREE_BUCKET_TARGET=$t tools/bq.sh >> gpurun_out/r03/fill64.txt 2>&1; done
for t in 1 1.25 1.5 2; do UTREE_BUCKET_TARGET=$t tools/bq.sh --kmer 64 --nodes 568000000 >> gpurun_out/r03/fill64.txt 2>&1; done
