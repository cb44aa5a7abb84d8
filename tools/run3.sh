# round-3 evidence run (GPU box): tests of the new cases, hit-dense at full scale, config-3 lognormal, N=2 rehearsal
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r03/t_configs.log 2>&1; tail -3 gpurun_out/r03/t_configs.log
python3 tests/scale/hit_dense.py --rc 0 > gpurun_out/r03/hit_dense_fwd.json 2> gpurun_out/r03/hit_dense_fwd.err
python3 tests/scale/hit_dense.py --rc 1 > gpurun_out/r03/hit_dense_rc.json 2> gpurun_out/r03/hit_dense_rc.err
UTREE_LANE_PASS=0 python3 tests/scale/hit_dense.py --rc 0 --sample 1000 > gpurun_out/r03/hit_dense_fwd_wave_per_read.json 2> /dev/null
tools/bq.sh --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 100000 --len-dist lognormal > gpurun_out/r03/config3_lognormal.txt 2>&1
tools/bq.sh --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 100000 >> gpurun_out/r03/config3_lognormal.txt 2>&1
python3 bench.py --gpus 2 --backend gloo --share-gpu0 --replicate torch --steps 4 --warmup 1 --e2e-reads-per-rank 4000000 > gpurun_out/r03/bench_n2_rehearsal.json 2> gpurun_out/r03/bench_n2_rehearsal.err
tail -c 600 gpurun_out/r03/bench_n2_rehearsal.err
