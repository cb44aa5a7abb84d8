# rocprofv3 passes of config 2 at 4 M-read launches (the launch size of rounds 1-3), folded into profiles/traffic.json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
profiles/run_prof.sh r04_config2_4M --batch-reads 4000000 > gpurun_out/r04/p7.log 2>&1; echo "rc=$?"
python3 profiles/make_traffic.py gpurun_out/prof_r04_config2_4M profiles/r04/prof_r04_config2_4M.txt > /dev/null
cp profiles/traffic.json gpurun_out/r04/traffic_with_4M.json
python3 bench.py --batch-reads 4000000 --no-e2e --no-cpu-baseline > gpurun_out/r04/bench_r04_n1_4M_batches.json 2>/dev/null
python3 -c "
import json; j=json.loads(open('gpurun_out/r04/bench_r04_n1_4M_batches.json').read().strip().splitlines()[-1]); r=j['roofline']
print('4M: value %.4g ms/step %.3f kernel %.3f frac %.3f used %s traffic %s rl %s' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['profile']['used'], r.get('traffic'), r.get('random_line_frac')))"
