#!/bin/bash
# Same-box comparison of launch sizes: bench.py --batch-reads B on config 2 without the CPU / file legs, every run builds its image anew.
# usage (GPU box): bash tools/batch_sweep.sh "4000000 16000000 4000000 16000000" > gpurun_out/batch_sweep.txt
# EXTRA="--kmer 64 --nodes 568000000" (or "--rc 1", ...) adds bench arguments; TOTAL = reads per run
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TOTAL=${TOTAL:-160000000}
echo "# reads per launch, steps, reads/s, ms per step, classify_lanes_k avg launch ms (HIP events), us per M reads in the kernel"
for b in $1; do
  s=$((TOTAL / b))
  python3 $R/bench.py $EXTRA --batch-reads $b --steps $s --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print($b, d['steps'], '%.4g' % d['value'], '%.4f' % d['ms_per_step'], '%.4f' % r['avg_launch_ms'], '%.1f' % (1e3 * r['avg_launch_ms'] / ($b / 1e6)))" || exit 1
done
