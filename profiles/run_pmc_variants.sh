#!/bin/bash
# SQ instruction counters of classify_short_k for several builds of the library (same box, one gpurun call).
# usage: profiles/run_pmc_variants.sh <name> [<name> ...]     (utree_amd/libexp_<name>.so, selected through UTREE_AMD_SO)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
    OUT=$R/gpurun_out/pmcv_$n
    mkdir -p $OUT
    export UTREE_AMD_SO=$R/utree_amd/libexp_$n.so
    timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY \
        --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt || { echo "$n failed"; exit 2; }
    python3 - "$OUT" "$n" <<'PY'
import sys, glob, csv, collections
d, n = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/pmc/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    if "classify_short_k" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print(n, {k: round(v / cnt[k] / 4e6, 2) for k, v in sorted(agg.items())}, flush=True)
PY
    find $OUT -name "*.csv" -size +1M -delete
done
