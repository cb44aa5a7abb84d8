#!/usr/bin/env python3
"""Hit-dense secondary workload (run on the GPU box through gpurun; prints one JSON object).

bench.py's synthetic reads carry floor(150/32) = 4 planted k-mers; reads cut from indexed genomes hit
on nearly every window (itree.c:929-935 appends every hit), which stresses what those reads never reach: the > 64-hit tally,
many distinct labels per read, long vote lists.  This script makes such a case end to end with the product's own tools:

  related reference genomes (mutated copies of a few roots, GG-style 8-rank labels, so k-mers collide at every rank)
    --utree-buildGG (complevel 0: every k-mer)--> .ubt --xtree-compress--> .ctr --> device image
  reads: 150 bp slices of the references, 1 % substitutions, a quarter reverse-complemented

and reports reads/s of the resident-batch hot path (as bench.py does), the hit statistics, and parity of a sample against the
GENUINE reference binary (oracle/_ref/xtree-searchGG on the same .ctr and FASTA).  Under `rocprofv3 --kernel-trace --stats`
the per-kernel split (classify_short_k / vote_k) comes out of the same run.

usage: hit_dense.py [--refs 1000] [--ref-len 1000000] [--reads 4000000] [--rc 1] [--steps 5]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--refs", type=int, default=1000)
    ap.add_argument("--ref-len", type=int, default=1_000_000)
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--rc", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--complevel", type=int, default=0)
    ap.add_argument("--sample", type=int, default=200_000, help="reads of the parity check against the genuine reference")
    ap.add_argument("--oracle-reads", type=int, default=0, help="reads of batch 0 compared record by record with the CPU oracle")
    ap.add_argument("--dir", default="/dev/shm/utree_hitdense")
    args = ap.parse_args()
    import numpy as np
    import torch
    from utree_amd import lib
    from utree_amd.search import CtrDB, DeviceTree, search_gg
    dev = torch.device("cuda:0")
    os.makedirs(args.dir, exist_ok=True)
    d = args.dir
    out = {"refs": args.refs, "ref_len": args.ref_len, "reads": args.reads, "read_len": args.read_len, "rc": args.rc, "complevel": args.complevel}
    from utree_amd import synth
    rdb = synth.make_related_db(dev, d, refs=args.refs, ref_len=args.ref_len, complevel=args.complevel)
    db, tree = rdb.ctr, rdb.tree
    out["generate_refs_seconds"] = rdb.seconds["generate_refs"]
    out["build_seconds"] = rdb.seconds["build"]
    out["compress_seconds"] = rdb.seconds["compress"]
    out["upload_seconds"] = rdb.seconds["upload"]
    out["nodes"] = int(db.n_nodes)
    out["labels"] = int(db.n_labels)
    out["image_GiB"] = tree.info.image_bytes / 2**30
    out["bucket_bytes"] = int(tree.info.bucket_bytes)
    L = args.read_len

    def run(cmd):
        t = time.time()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        return time.time() - t, r

    def make_batch(seed):
        return synth.make_related_reads(rdb, args.reads, L, seed=seed).bases
    nb = min(3, args.steps)
    batches = [make_batch(100 + b) for b in range(nb)]
    off = torch.arange(args.reads, dtype=torch.int64, device=dev) * L
    ln = torch.full((args.reads,), L, dtype=torch.int32, device=dev)
    outs = [torch.empty((args.reads, 6), dtype=torch.int32, device=dev) for _ in range(nb)]
    ws = torch.empty(tree.workspace_bytes(args.reads, args.reads * L, L, bool(args.rc)), dtype=torch.uint8, device=dev)

    def step(i):
        tree.classify(batches[i % nb], off, ln, rc=bool(args.rc), total_bases=args.reads * L, max_len=L, out=outs[i % nb], workspace=ws)
    step(0)
    torch.cuda.synchronize()
    tree.kernel_time(reset=True)
    t0 = time.time()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    el = time.time() - t0
    k_ms, k_n = tree.kernel_time(reset=True)
    res = outs[0]
    found = res[:, 2].float()
    out["reads_per_second"] = args.reads * args.steps / el
    out["ms_per_step"] = 1e3 * el / args.steps
    out["kernel"] = tree.kernel_name()
    out["classify_kernel_avg_ms"] = k_ms / max(1, k_n)
    out["windows_per_second"] = out["reads_per_second"] * (L - 31) * (2 if args.rc else 1)
    out["hits_per_read_mean"] = float(found.mean())
    out["hits_per_read_max"] = int(found.max())
    out["reads_with_more_than_64_hits"] = float((found > 64).float().mean())
    out["distinct_labels_per_read_mean"] = float(res[:, 3].float().mean())
    out["distinct_labels_per_read_max"] = int(res[:, 3].max())
    out["classified_fraction"] = float((found > 0).float().mean())
    mc = tree.model_counts(batches[0], off[:100_000], ln[:100_000], rc=bool(args.rc))
    out["model_counts_100k_reads"] = mc

    # parity of a sample against the genuine reference on the same files
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "xtree-searchGG")
    ns = min(args.sample, args.reads)
    host = batches[0][: ns * L].cpu().numpy().reshape(ns, L)
    with open(d + "/sample.fa", "wb") as f:
        for i in range(ns):
            f.write(b">q%d\n" % i)
            f.write(host[i].tobytes())
            f.write(b"\n")
    rcarg = ["RC"] if args.rc else []
    code, st = search_gg(db, [tree], d + "/sample.fa", d + "/ours.txt", rc=bool(args.rc), threads=16)
    assert code == 0
    out["file_search_pipeline"] = "device text" if st.pipeline else "host text"

    def sorted_sha(p):
        l = open(p, "rb").read().split(b"\n")
        l.sort()
        return hashlib.sha256(b"\n".join(l)).hexdigest(), len(l) - 1
    if os.path.exists(ref_bin):
        open(d + "/empty.fa", "wb").close()
        t_load, r = run([ref_bin, d + "/db.ctr", d + "/empty.fa", d + "/e.txt", "16"] + rcarg)
        t_all, r = run([ref_bin, d + "/db.ctr", d + "/sample.fa", d + "/ref.txt", "16"] + rcarg)
        assert r.returncode == 0
        a, b = sorted_sha(d + "/ours.txt"), sorted_sha(d + "/ref.txt")
        out["parity_sample_reads"] = ns
        out["parity_sorted_lines_identical"] = (a == b)
        out["lines"] = a[1]
        out["reference_reads_per_second_16_threads"] = ns / max(1e-6, t_all - t_load)
    # ... and the batch results record by record against the CPU oracle (test infrastructure: the checker, not the thing measured)
    if args.oracle_reads:
        from oracle import orc
        no = min(args.oracle_reads, args.reads)
        o = orc.OracleDB.load(d + "/db.ctr")
        hostb = batches[0][: no * L].cpu().numpy()
        want = o.classify_batch(hostb, np.arange(no, dtype=np.uint64) * L, np.full(no, L, dtype=np.uint32), rc=bool(args.rc), threads=16)
        got = outs[0][:no].cpu().numpy()
        gu = got.view(np.uint32)
        hit = want["found"] > 0
        multi = hit & (want["uix"] > 1)
        out["oracle_reads"] = no
        out["oracle_records_identical"] = bool(np.array_equal(gu[:, 2], want["found"]) and np.array_equal(gu[hit, 3], want["uix"][hit]) and
                                               np.array_equal(gu[hit, 0], want["label"][hit]) and np.array_equal(got[hit, 1], want["cut"][hit]) and
                                               np.array_equal(gu[multi, 4], want["sl"][multi]) and np.array_equal(gu[multi, 5], want["ol"][multi]))
    print(json.dumps(out, indent=1))
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))


if __name__ == "__main__":
    main()
