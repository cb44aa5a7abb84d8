#!/bin/bash
# same box, alternating: the library of the previous commit (utree_amd/libexp_head.so) against this tree's, on the four kernel-rate configurations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r04b
{
bash tools/ab3.sh "" head main
bash tools/ab3.sh "--rc 1 --steps 5" head main
bash tools/ab3.sh "--kmer 64 --nodes 568000000 --steps 5" head main
} 2>&1 | tee gpurun_out/r04b/ab_head_vs_chain_code.txt
