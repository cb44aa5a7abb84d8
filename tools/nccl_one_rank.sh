mkdir -p gpurun_out/r03
MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 UTREE_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --nodes 300000000 > gpurun_out/r03/bench_nccl_1rank.json 2> gpurun_out/r03/bench_nccl_1rank.err
tail -c 400 gpurun_out/r03/bench_nccl_1rank.err
python3 -c "
import json; j=json.loads(open('gpurun_out/r03/bench_nccl_1rank.json').read().strip().splitlines()[-1]); print({k:j.get(k) for k in ('value','n_gpus','ranks','bcast_s','bcast')}); print(j['config']['parallelism'])"
