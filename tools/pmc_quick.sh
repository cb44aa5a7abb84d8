#!/bin/bash
# SQ instruction counters of the dominant classify kernel for the current library (two --pmc passes), per read
# usage: tools/pmc_quick.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -- python3 $R/bench.py $A > $OUT/b1.json 2> $OUT/e1.txt || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -- python3 $R/bench.py $A > $OUT/b2.json 2> $OUT/e2.txt || exit 2
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections, json
d = sys.argv[1]
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(d + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "classify_short_k" in r["Kernel_Name"] or "classify_long_k" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
j = json.loads(open(d + "/b1.json").read().strip().splitlines()[-1])
n = j["roofline"]["reads_per_launch"]
per = {k: v / cnt[k] for k, v in agg.items()}
cyc = per.get("GRBM_GUI_ACTIVE", 0) / 8
print("per read:", {k: round(v / n, 2) for k, v in sorted(per.items()) if k.startswith("SQ_INSTS") or k.startswith("SQ_ACTIVE")})
if cyc:
    print("cycles/launch %.3g  VALU busy (ACTIVE_INST_VALU*4/1024/cycles) %.3f  insts*4 %.3f  SALU (per CU) %.3f  any %.3f" % (
        cyc, per["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, per["SQ_INSTS_VALU"] * 4 / 1024 / cyc, per["SQ_INSTS_SALU"] / 256 / cyc,
        per["SQ_ACTIVE_INST_ANY"] * 4 / 1024 / cyc))
    print("wave cycles / (waves x cycles): %.3f   wait_inst_any / wave_cycles: %.3f" % (per["SQ_WAVE_CYCLES"] / (8192 * cyc) if "SQ_WAVE_CYCLES" in per else 0, per["SQ_WAIT_INST_ANY"] / per["SQ_WAVE_CYCLES"]))
print("kernel avg ms (HIP events, under profiler):", j["roofline"]["avg_launch_ms"])
PY
find $OUT -name "*.csv" -size +1M -delete
