#!/bin/bash
# rocprofv3 passes for bench.py's workload (run on the GPU box through gpurun).
# usage: profiles/run_prof.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# pass 1: the default bench command (10 steps, 2 warm-up) minus the CPU baseline legs, kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline $* > $OUT/trace.json 2> $OUT/trace.err || exit 1
# counter passes (each on its own, never together with a trace domain): fewer steps are enough
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || exit 2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 4
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $R/bench.py $ARGS > $OUT/pmc_tcc.json 2> $OUT/pmc_tcc.err || exit 5
# keep only small summaries
find $OUT -name "*.csv" -size +8M -delete
echo done
