#!/bin/bash
# rocprofv3 kernel traces of the three widened rows' command lines (run on the GPU box through gpurun):
#   xtree-search (rank-specific search), utree-buildGG (database BUILD), xtree-compress (.ubt -> .ctr)
# usage: profiles/run_prof_tools.sh <tag>
set -o pipefail
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/proftools_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
# inputs: a 400 M-node CTR + 4 M reads; 600 x 1 Mbp references
KEEP_FILES=1 python3 tests/scale/e2e_scale.py --rank --nodes 400000000 --reads 4000000 --skip-reference > $OUT/gen_search.json 2> $OUT/gen_search.err || exit 1
KEEP_FILES=1 SKIP_REFERENCE=1 python3 tests/scale/build_bench.py 600 1000000 1 > $OUT/gen_build.json 2> $OUT/gen_build.err || exit 2
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rank -- $R/utree_amd/xtree-search /dev/shm/utree_e2e/synth.ctr /dev/shm/utree_e2e/reads.fa /dev/shm/utree_e2e/o.txt 16 > $OUT/rank.out 2> $OUT/rank.err || exit 3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/build -- $R/utree_amd/utree-buildGG /dev/shm/utree_bld/refs.fa /dev/shm/utree_bld/refs.map /dev/shm/utree_bld/p.ubt 0 1 > $OUT/build.out 2> $OUT/build.err || exit 4
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/compress -- $R/utree_amd/xtree-compress /dev/shm/utree_bld/p.ubt /dev/shm/utree_bld/p.ctr > $OUT/compress.out 2> $OUT/compress.err || exit 5
rm -rf /dev/shm/utree_e2e /dev/shm/utree_bld
find $OUT -name "*.csv" -size +8M -delete
echo done
