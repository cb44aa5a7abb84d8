cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04b
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04b/gputests.log 2>&1; rc=$?; tail -3 gpurun_out/r04b/gputests.log; [ $rc -eq 0 ] || exit $rc
bash tools/quick_bench.sh 2>&1 | tee gpurun_out/r04b/quick_bench_with_chain_code.txt
