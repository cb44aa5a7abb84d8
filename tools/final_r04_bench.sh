# closing run, part 2: the bench lines that cite profiles/traffic.json (part 1: tools/final_r04_profiles.sh)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python3 bench.py > gpurun_out/r04/bench_r04_n1.json 2> gpurun_out/r04/bench_r04_n1.err
python3 bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r04/bench_r04_n1_driver_cmd.json 2>/dev/null
python3 bench.py --rc 1 --no-cpu-baseline --no-e2e > gpurun_out/r04/bench_r04_n1_rc.json 2>/dev/null
python3 bench.py --kmer 64 --nodes 568000000 --no-cpu-baseline --no-e2e > gpurun_out/r04/bench_r04_n1_k64.json 2>/dev/null
python3 bench.py --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 400000 --len-dist lognormal --model-reads 2000 > gpurun_out/r04/bench_r04_n1_config3.json 2>/dev/null
python3 bench.py --batch-reads 4000000 --no-e2e --no-cpu-baseline > gpurun_out/r04/bench_r04_n1_4M_batches.json 2>/dev/null
python3 bench.py --workload hit_dense --no-cpu-baseline --no-e2e > gpurun_out/r04/hit_dense_fwd.json 2>/dev/null
python3 bench.py --workload hit_dense --rc 1 --no-cpu-baseline --no-e2e > gpurun_out/r04/hit_dense_rc.json 2>/dev/null
bash tools/nccl_one_rank.sh gpurun_out/r04 > /dev/null 2>&1
UTREE_BENCH_CLI_LEG=1 UTREE_RCCL_FORCE=1 python3 bench.py --no-cpu-baseline --no-e2e --steps 3 > gpurun_out/r04/bench_cli_leg.json 2>/dev/null
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04/bench_r04_n1*.json") + glob.glob("gpurun_out/r04/hit_dense_*.json") + ["gpurun_out/r04/bench_cli_leg.json", "gpurun_out/r04/bench_nccl_1rank.json"]):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); r = j["roofline"]
        print("%-50s value %.4g ms/step %.3f kernel %.3f frac %.3f used %s %s" % (f.split("/")[-1], j["value"], j["ms_per_step"], r["avg_launch_ms"], r["frac"], r["profile"]["used"], r["kernel"]))
        if "e2e" in j:
            e = j["e2e"]; print("    e2e %.4g parts %s db_load %s cpu %.4g parity %s %s" % (e["value"], e.get("output_in_parts", {}).get("reads_per_second"), e["db_load"]["runs_seconds"], j["cpu_baseline"]["value"], j["cpu_baseline"]["parity_ok"], e.get("parity_ok")))
        if "cli" in j:
            c = j["cli"]; c.pop("stderr_tail", None); print("    cli", {k: c.get(k) for k in ("value", "wall_seconds", "broadcast_seconds", "fanout", "output_in_parts")})
        if "bcast_image_s" in j: print("    bcast", j["bcast_s"], j["bcast_image_s"])
    except Exception as ex:
        print(f, "ERR", ex)
PY
