#!/bin/bash
# rocprofv3 passes for bench.py's workload (run on the GPU box through gpurun).
# usage: profiles/run_prof.sh <tag> [bench args...]
# Leaves gpurun_out/prof_<tag>/: trace (kernel trace + stats), pmc_* (counter passes, each on its own), the bench lines, and
# kernel_source_sha256.txt -- the hash bench.py later checks a kept profile against (profiles/make_traffic.py).
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); from utree_amd import lib; print(lib.kernel_source_sha256())" > $OUT/kernel_source_sha256.txt || exit 9
echo "$*" > $OUT/bench_args.txt
LEGS="--no-cpu-baseline --no-e2e"
# pass 1: the default bench command (10 steps, 2 warm-up) minus the CPU / file legs, kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $LEGS $* > $OUT/trace.json 2> $OUT/trace.err || exit 1
# counter passes (each on its own, never together with a trace domain): fewer steps are enough
ARGS="--steps 3 --warmup 1 $LEGS $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || exit 2
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py $ARGS > $OUT/pmc_sq2.json 2> $OUT/pmc_sq2.err || echo "pmc_sq2 pass failed (optional)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 4
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $R/bench.py $ARGS > $OUT/pmc_tcc.json 2> $OUT/pmc_tcc.err || exit 5
python3 $R/profiles/summarize.py $OUT > $OUT/summary.txt 2> $OUT/summary.err || echo "summary failed"
# keep only small files (gpurun brings back at most 64 MiB): the raw traces and counter tables are folded into summary.txt / counters.json
find $OUT -name "*.csv" -size +256k -delete
find $OUT -name "*.err" -size +64k -delete
echo done
