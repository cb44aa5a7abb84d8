#!/bin/bash
# file -> file A/B on the GPU box: device text pipeline vs host framing/formatting, same files (tests/scale/e2e_scale.py keeps them)
# usage: tools/e2e_ab.sh <nodes> <reads>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
D=/dev/shm/utree_e2e
export UTREE_TIMING=1 KEEP_FILES=1
python3 $R/tests/scale/e2e_scale.py --nodes $1 --reads $2 --skip-reference --threads 16 > $R/gpurun_out/e2e_dev.json 2> $R/gpurun_out/e2e_dev.err || exit 1
for mode in dev host dev host; do
    if [ $mode = host ]; then export UTREE_HOST_TEXT=1; else unset UTREE_HOST_TEXT; fi
    $R/utree_amd/xtree-searchGG $D/synth.ctr $D/reads.fa $D/ours_$mode.txt 16 > /dev/null 2> $R/gpurun_out/e2e_cli_$mode.err || exit 2
    grep -E "pipeline|stages|search " $R/gpurun_out/e2e_cli_$mode.err
done
cmp $D/ours_dev.txt $D/ours_host.txt && echo "outputs identical: $(wc -c < $D/ours_dev.txt) bytes, $(wc -l < $D/ours_dev.txt) lines"
rm -rf $D
