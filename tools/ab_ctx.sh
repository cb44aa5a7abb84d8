#!/bin/bash
# same box, alternating: the shipped library against the variant in which a searched run's context is worked out once per run (utree_amd/libexp_ctx.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r04b
{
UTREE_AMD_SO=$R/utree_amd/libexp_ctx.so timeout -k 10 600 python3 -m pytest tests/test_gpu_lanes.py tests/test_gpu_configs.py -x -q -k "hit_dense or related or chains or overflow" 2>&1 | tail -2
bash tools/ab3.sh "--workload hit_dense" main ctx
bash tools/ab3.sh "--workload hit_dense --rc 1" main ctx
bash tools/ab3.sh "" main ctx
bash tools/ab3.sh "--kmer 64 --nodes 568000000 --steps 5" main ctx
} 2>&1 | tee gpurun_out/r04b/ab_context_once_per_run.txt
