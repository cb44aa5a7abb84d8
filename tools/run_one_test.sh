cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04b
timeout -k 10 600 python3 -m pytest tests/test_gpu_search_dev.py -x -q -k "eight_device or two_device" > gpurun_out/r04b/one_test.log 2>&1; tail -25 gpurun_out/r04b/one_test.log
