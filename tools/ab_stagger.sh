#!/bin/bash
# Experiment: does a launch of the lane pass that starts its three workgroups per CU a third of a grab apart
# lose less time to the lockstep of its first rounds?  The shipped sources stay as they are (every kept profile is tied to their hash):
# the patch is applied to a scratch copy of the tree on the GPU box, built there, and the two libraries run alternately on the same box.
# usage (GPU box): bash tools/ab_stagger.sh [sleeps per third, default 7 (s_sleep 127 = 8128 cycles each)] > gpurun_out/ab_stagger.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-7}
V=/tmp/utree_stagger
rm -rf $V && mkdir -p $V && (cd $R && tar cf - --exclude=gpurun_out --exclude=.git .) | (cd $V && tar xf -) || exit 1
python3 - "$V/utree_amd/csrc/lanes_core.hpp" <<'P' || exit 2
import sys
p = sys.argv[1]; s = open(p).read()
at = "    LT_DECL\n    for (;;) {\n"
assert s.count(at) == 1
s = s.replace(at, """    if constexpr (MODE == 0) {                                          // (experiment: the CU's second / third workgroup starts later)
        const uint32_t third = (blockIdx.x / ((gridDim.x + 2u) / 3u)) % 3u;
        if (gridDim.x >= 96u) for (uint32_t q = 0; q < third * UTREE_LANES_STAGGER; ++q) __builtin_amdgcn_s_sleep(127);
    }
""" + at)
open(p, "w").write(s)
P
(cd $V/utree_amd/csrc && make -j16 LANESFLAGS=-DUTREE_LANES_STAGGER=$N ARCH=gfx950 > $V/build.log 2>&1) || { tail -20 $V/build.log; exit 3; }
echo "# A = shipped library, B = workgroups of a CU started $N x s_sleep(127) apart; config 2, alternating runs on one box"
for rep in 1 2 3; do
  for b in 4000000 16000000; do
    echo -n "A "; GRAFT_REPO_ROOT=$R bash $R/tools/batch_sweep.sh $b | tail -1 || exit 4
    echo -n "B "; GRAFT_REPO_ROOT=$V bash $R/tools/batch_sweep.sh $b | tail -1 || exit 5
  done
done
echo "# parity of B: the variant's 240 000-read full-size test"
(cd $V && python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config2_k32" 2>&1 | tail -2)
