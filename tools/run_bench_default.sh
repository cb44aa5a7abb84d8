cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
python3 bench.py > gpurun_out/r04/bench_r04_n1.json 2> gpurun_out/r04/bench_r04_n1.err; echo rc=$?
python3 -c "
import json; j=json.loads(open('gpurun_out/r04/bench_r04_n1.json').read().strip().splitlines()[-1]); r=j['roofline']; e=j['e2e']
print('value %.4g ms/step %.3f kernel %.3f frac %.3f used %s' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['profile']['used']))
print('e2e %.4g parity %s db_load %s cpu %.4g parity %s' % (e['value'], e.get('parity_ok'), e['db_load']['runs_seconds'], j['cpu_baseline']['value'], j['cpu_baseline']['parity_ok']))
"
python3 -c "import __graft_entry__ as g; g.smoke()"
