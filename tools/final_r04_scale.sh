#!/bin/bash
# closing run, part 3: the command lines against the GENUINE reference binaries at scale, with the shipped library (tests/scale): whole chain
# FASTA -> .ubt -> .ctr -> classifications, our CLI vs the reference's on big synthetic .ctr files (k = 32 +- RC, k = 64, long reads), and the full suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gputests_final.log 2>&1; tail -2 $O/gputests_final.log
python3 tests/scale/chain_check.py > $O/chain_check.json 2> $O/chain_check.err; echo "chain check rc=$?"
python3 tests/scale/e2e_scale.py --threads 16 > $O/e2e_scale_k32.json 2> $O/e2e_scale_k32.err; echo "e2e scale k32 rc=$?"
python3 tests/scale/e2e_scale.py --threads 16 --nodes 400000000 --reads 2000000 --rc 1 > $O/e2e_scale_k32_rc.json 2> $O/e2e_scale_k32_rc.err; echo "e2e scale k32 rc rc=$?"
python3 tests/scale/e2e_scale.py --threads 16 --kmer 64 --nodes 568000000 --reads 2000000 > $O/e2e_scale_k64.json 2> $O/e2e_scale_k64.err; echo "e2e scale k64 rc=$?"
python3 tests/scale/e2e_scale.py --threads 16 --nodes 72000000 --reads 50000 --read-len 10000 --rc 1 > $O/e2e_scale_long_rc.json 2> $O/e2e_scale_long_rc.err; echo "e2e scale long rc rc=$?"
UTREE_OVF_CHAINS=1 python3 tests/scale/hit_dense.py --rc 1 --sample 200000 --steps 2 > $O/hit_dense_chains_vs_reference.json 2> /dev/null; echo "hit dense (chains) vs reference rc=$?"
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04/e2e_scale_*.json") + ["gpurun_out/r04/chain_check.json", "gpurun_out/r04/hit_dense_chains_vs_reference.json"]):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], {k: v for k, v in j.items() if "identical" in k or "parity" in k or k in ("ours_seconds", "reference_seconds", "speedup", "kernel")})
    except Exception as ex:
        print(f, "ERR", ex)
PY
