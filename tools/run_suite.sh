cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04/gputests_final.log 2>&1; rc=$?; tail -3 gpurun_out/r04/gputests_final.log; grep -n "utree_amd\] warning" gpurun_out/r04/gputests_final.log | head; exit $rc
