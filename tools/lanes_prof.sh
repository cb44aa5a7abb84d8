#!/bin/bash
# kernel durations (kernel trace) and SQ counters per read of classify_lanes_k and of the exception pass behind it
# usage: tools/lanes_prof.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/lanesprof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/bench.py $A > $OUT/bt.json 2> $OUT/et.txt || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -- python3 $R/bench.py $A > $OUT/b1.json 2> $OUT/e1.txt || exit 2
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/p2 -- python3 $R/bench.py $A > $OUT/b2.json 2> $OUT/e2.txt || exit 3
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p3 -- python3 $R/bench.py $A > $OUT/b3.json 2> $OUT/e3.txt || echo "lds pass failed"
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections, json
d = sys.argv[1]
for f in glob.glob(d + "/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-90s calls %s avg %.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
j = json.loads(open(d + "/b1.json").read().strip().splitlines()[-1])
n = j["roofline"]["reads_per_launch"]
for kn in ("classify_lanes_k", "classify_short_k"):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/p*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kn in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    if not agg: continue
    per = {k: v / cnt[k] for k, v in agg.items()}
    print(kn, "per read of the batch:", {k: round(v / n, 2) for k, v in sorted(per.items())})
    cyc = per.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc:
        print("  cycles/launch %.3g  VALU issue %.3f  SALU %.3f" % (cyc, per["SQ_INSTS_VALU"] * 4 / 1024 / cyc, per["SQ_INSTS_SALU"] / 256 / cyc))
PY
find $OUT -name "*.csv" -size +1M -delete
