#!/bin/bash
# more SQ counters of the dominant classify kernel (latency / stall view), per read; usage: tools/pmc_more.sh <tag> [variant]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; V=$2
[ -n "$V" ] && [ "$V" != main ] && export UTREE_AMD_SO=$R/utree_amd/libexp_$V.so
OUT=$R/gpurun_out/pmcm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e"
rocprofv3 --pmc SQ_INSTS SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $R/bench.py $A > $OUT/b1.json 2> $OUT/e1.txt || echo "p1 failed"
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_LDS --output-format csv -d $OUT/p2 -- python3 $R/bench.py $A > $OUT/b2.json 2> $OUT/e2.txt || echo "p2 failed"
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python3 $R/bench.py $A > $OUT/b3.json 2> $OUT/e3.txt || echo "p3 failed"
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections, json
d = sys.argv[1]
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(d + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "classify_short_k" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
n = 4e6
per = {k: v / cnt[k] for k, v in agg.items()}
for k in sorted(per): print("%-26s %12.4g per launch  %10.2f per read" % (k, per[k], per[k] / n))
PY
find $OUT -name "*.csv" -size +1M -delete
