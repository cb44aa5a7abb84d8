# The N>1 code path of bench.py with ONE rank on a one-GPU lease: nccl process group, utree_rccl_unique_id, and -- UTREE_RCCL_FORCE=1 --
# utree_dev_replicate_rank's ncclCommInitRank, size broadcast and image broadcast (<= 1 GiB pieces) into a second allocation on the card;
# the steps then run on the RECEIVED copy.  Usage: bash tools/nccl_one_rank.sh [out-dir]
out=${1:-gpurun_out/r04}
mkdir -p $out
MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 UTREE_BENCH_FORCE_DIST=1 UTREE_RCCL_FORCE=1 python3 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-e2e > $out/bench_nccl_1rank.json 2> $out/bench_nccl_1rank.err
tail -c 400 $out/bench_nccl_1rank.err
python3 -c "
import json; j=json.loads(open('$out/bench_nccl_1rank.json').read().strip().splitlines()[-1]); print({k:j.get(k) for k in ('value','n_gpus','ranks','bcast_s','bcast_image_s','bcast','rccl_forced_one_rank')}); print(j['config']['parallelism'])"
