cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04b
timeout -k 10 600 python3 -m pytest tests/test_gpu_search_dev.py -x -q > gpurun_out/r04b/search_dev_tests.log 2>&1; tail -3 gpurun_out/r04b/search_dev_tests.log
