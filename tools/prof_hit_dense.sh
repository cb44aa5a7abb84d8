# rocprofv3 passes for the hit-dense workload: the default image (records behind a position directory) and the opt-in chains (UTREE_OVF_CHAINS=1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
profiles/run_prof.sh r04_hit_dense --workload hit_dense > gpurun_out/r04/p5.log 2>&1; echo "dir rc=$?"
UTREE_OVF_CHAINS=1 profiles/run_prof.sh r04_hit_dense_chains --workload hit_dense > gpurun_out/r04/p6.log 2>&1; echo "chains rc=$?"
python3 profiles/make_traffic.py gpurun_out/prof_r04_hit_dense profiles/r04/prof_r04_hit_dense.txt > /dev/null
cp profiles/traffic.json gpurun_out/r04/traffic_with_hit_dense.json
for t in hit_dense hit_dense_chains; do python3 - $t <<'PY'
import json, sys
c = json.load(open("gpurun_out/prof_r04_%s/counters.json" % sys.argv[1]))
sig = c["bench_line"]["roofline"]["kernel"]
k = [x for x in c["counters"] if sig in x][0]; v = c["counters"][k]
st = [s for n, s in c["kernel_stats"].items() if sig in n][0]
n = 4_000_000
print(sys.argv[1], "avg ms %.3f" % (st["AverageNs"] / 1e6 if "AverageNs" in st else -1), "L2 misses/read %.1f" % (v["TCC_MISS_sum"] / n), "VALU/read %.0f" % (v["SQ_INSTS_VALU"] / n), "HBM GB %.2f" % ((v["FETCH_SIZE"] * 2 + v["WRITE_SIZE"]) * 1024 / 1e9))
PY
done
