#!/bin/bash
# the command line on the bench's database as a `.ctr` file: where its wall time goes (UTREE_TIMING phase lines); usage: tools/cli_load_check.sh [reads]
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$(mktemp -d /dev/shm/utree_cli_XXXX)
python3 $R/bench.py --make-files $D --e2e-reads ${1:-4000000} > /dev/null 2>&1
ls -la $D
for rep in 1 2; do
  sleep ${SLEEP:-0}; T0=$(date +%s.%N)
  UTREE_TIMING=1 $R/utree_amd/xtree-searchGG $D/db.ctr $D/reads.fa $D/out.txt 16 2>&1 | grep -v "batch report\|Searched"
  python3 -c "import time,sys; print('wall %.2f s' % (time.time() - float(sys.argv[1])))" $T0
done
rm -rf $D
