#!/usr/bin/env python3
"""Batches of mixed read lengths through the lane-per-read pass: timings on the GPU box, one JSON object.
  A  4 M x 150 bp                                  B  the same with 1 % of the reads 300 bp
  C  2 M x 150 bp + 20 000 x 10 kb in ONE batch    D, E  the two halves as batches of their own
(--scale S multiplies the read counts; the keys keep the names of scale 1)
usage: mixed_check.py [--nodes 1217000000] [--scale 4]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_217_000_000)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--scale", type=int, default=1, help="multiplies every batch's read count (4: the 16 M-read launches of DESIGN.md section 0.12)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from utree_amd import synth
    dev = torch.device("cuda:0")
    sdb = synth.make_db(dev, args.nodes, W=8)
    tree = sdb.tree

    def timed(reads, rc=False):
        tot, mx = int(reads.length.sum().item()), int(reads.length.max().item())
        ws = torch.empty(tree.workspace_bytes(reads.n, tot, mx, rc), dtype=torch.uint8, device=dev)
        out = torch.empty((reads.n, 6), dtype=torch.int32, device=dev)
        tree.classify(reads.bases, reads.off, reads.length, rc=rc, total_bases=tot, max_len=mx, out=out, workspace=ws)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.reps):
            tree.classify(reads.bases, reads.off, reads.length, rc=rc, total_bases=tot, max_len=mx, out=out, workspace=ws)
        torch.cuda.synchronize()
        tree.poll()
        return 1e3 * (time.time() - t0) / args.reps, tree.kernel_name(), int((out[:, 2] > 0).sum().item())
    S = args.scale
    n = 4_000_000 * S
    a = synth.make_reads_var(sdb, np.full(n, 150, dtype=np.int32), seed=11)
    lb = np.full(n, 150, dtype=np.int32)
    lb[np.random.default_rng(5).choice(n, n // 100, replace=False)] = 300
    b = synth.make_reads_var(sdb, lb, seed=11)
    lc = np.concatenate([np.full(2_000_000 * S, 150, dtype=np.int32), np.full(20_000 * S, 10_000, dtype=np.int32)])
    np.random.default_rng(6).shuffle(lc)
    c = synth.make_reads_var(sdb, lc, seed=12)
    d = synth.make_reads_var(sdb, np.full(2_000_000 * S, 150, dtype=np.int32), seed=13)
    e = synth.make_reads_var(sdb, np.full(20_000 * S, 10_000, dtype=np.int32), seed=14)
    out = {"scale": S, "nodes": args.nodes, "image_GiB": tree.info.image_bytes / 2**30, "bucket_bytes": tree.info.bucket_bytes}
    for tag, r in (("A_4M_x_150bp", a), ("B_4M_x_150bp_with_1pct_300bp", b), ("C_2M_x_150bp_plus_20k_x_10kb_one_batch", c), ("D_2M_x_150bp", d), ("E_20k_x_10kb", e)):
        ms, kn, found = timed(r)
        out[tag] = {"ms_per_batch": ms, "kernel": kn, "reads_with_hits": found}
    out["B_over_A"] = out["B_4M_x_150bp_with_1pct_300bp"]["ms_per_batch"] / out["A_4M_x_150bp"]["ms_per_batch"]
    out["C_over_D_plus_E"] = out["C_2M_x_150bp_plus_20k_x_10kb_one_batch"]["ms_per_batch"] / (out["D_2M_x_150bp"]["ms_per_batch"] + out["E_20k_x_10kb"]["ms_per_batch"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
