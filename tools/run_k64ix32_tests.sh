cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04b
timeout -k 10 800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_search_dev.py tests/test_gpu_compress.py tests/test_gpu_lanes.py -x -q -k "k64ix32 or related_genomes_k64" > gpurun_out/r04b/k64ix32_tests.log 2>&1; tail -15 gpurun_out/r04b/k64ix32_tests.log
