# round-4 evidence: rocprofv3 passes (kernel trace + counter passes, each on its own) for the BASELINE configurations at bench.py's launch size
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
profiles/run_prof.sh r04_config2_16M > gpurun_out/r04/p1.log 2>&1
profiles/run_prof.sh r04_config2_rc_16M --rc 1 > gpurun_out/r04/p2.log 2>&1
profiles/run_prof.sh r04_config5_k64_16M --kmer 64 --nodes 568000000 > gpurun_out/r04/p3.log 2>&1
profiles/run_prof.sh r04_config3_lognormal_rc_400k --nodes 72000000 --read-len 10000 --rc 1 --batch-reads 400000 --len-dist lognormal --model-reads 2000 > gpurun_out/r04/p4.log 2>&1
du -sh gpurun_out; echo finished
