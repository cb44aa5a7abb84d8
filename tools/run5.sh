set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03/t_full_64b.log 2>&1; tail -4 gpurun_out/r03/t_full_64b.log
UTREE_BUCKET_BYTES=128 timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03/t_full_128.log 2>&1; tail -4 gpurun_out/r03/t_full_128.log
tools/ab_v8.sh > gpurun_out/r03/ab_v8_vs_r03_64B.txt 2>&1
