# closing run, part 1: suite + the rocprofv3 passes of the shipped sources, folded into profiles/traffic.json (part 2: tools/final_r04_bench.sh)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputests_final.log 2>&1; tail -3 gpurun_out/r04/gputests_final.log
bash tools/evidence_r04.sh > /dev/null 2>&1
for t in config2_16M config2_rc_16M config5_k64_16M config3_lognormal_rc_400k; do python3 profiles/make_traffic.py gpurun_out/prof_r04_$t profiles/r04/prof_r04_$t.txt > /dev/null; done
mkdir -p gpurun_out/r04/profiles_out; cp profiles/traffic.json gpurun_out/r04/profiles_out/; cp profiles/r04/prof_r04_*.txt gpurun_out/r04/profiles_out/
ls -la gpurun_out/r04/profiles_out
