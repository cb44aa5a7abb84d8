#!/usr/bin/env python3
"""Randomised cross-check of the lane-per-read pass against the wave-per-read kernels (UTREE_LANE_PASS=1 / 0) on the GPU box:
random database sizes (sparse to dense tables: UTREE_FINE_BITS), k = 32 / 64, read lengths and per-read length mixes within the
pass's range, both strand modes, random N / lowercase / other bytes, reads assembled from pieces of other reads (many labels),
unaligned buffers.  Records must be identical.  usage: lanes_fuzz.py [rounds] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from utree_amd import synth


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dev = torch.device("cuda:0")
    checked = lanes_used = 0
    for it in range(rounds):
        W = int(rng.choice([8, 8, 16]))
        k = 4 * W
        nodes = int(10 ** rng.uniform(5.0, 7.7))
        os.environ["UTREE_FINE_BITS"] = str(int(rng.choice([8, 8, 6, 4, 2, 0])))
        os.environ["UTREE_BUCKET_BYTES"] = str(int(rng.choice([64, 64, 128])))
        os.environ["UTREE_BUCKET_TARGET"] = "%.2f" % float(rng.choice([0, 0, 0.7, 1.5, 3, 6, 12, 30]))     # 0: the default load; small: few overflow runs, large: most buckets overflow
        # k = 64: several pairs of buckets per slot, picked by the four bases around the minimizer (image version 13; needs UTREE_FINE_BITS=8)
        ts = str(int(rng.choice([0, 2, 5]))) if W == 16 else "0"
        if ts != "0":
            os.environ["UTREE_TEST_SUB"] = ts
        else:
            os.environ.pop("UTREE_TEST_SUB", None)
        if rng.random() < 0.25:
            os.environ["UTREE_VOTE_BYTES"] = "1"
        else:
            os.environ.pop("UTREE_VOTE_BYTES", None)
        sdb = synth.make_db(dev, nodes, W=W)
        for sub in range(3):
            cap = (2095 if W == 8 else 1615)
            L = int(rng.choice([k, k + 1, 100, 150, 151, 160, 161, 250, 289 if W == 8 else 257, 300, 547 if W == 8 else 451, 700, 1063 if W == 8 else 839, 1200, cap, cap + 1, 2113, 3000, 5000, 12000]))
            n = int(rng.integers(1, 40_000 if L <= 600 else 8_000 if L <= 2200 else 1_500))
            reads = synth.make_reads(sdb, n, L, seed=int(rng.integers(1, 1 << 30)))
            bases = reads.bases.clone().view(n, L)
            g = torch.Generator(device=dev); g.manual_seed(int(rng.integers(1, 1 << 30)))
            # damage: N's, lowercase, foreign bytes, rows swapped in halves (chimeras)
            m = torch.rand((n, L), generator=g, device=dev)
            bases[m < 0.002] = ord("N")
            low = (m > 0.002) & (m < 0.01)
            bases[low] = bases[low] | 0x20
            bases[(m > 0.01) & (m < 0.0102)] = ord("-")
            half = torch.rand(n, generator=g, device=dev) < 0.3
            perm = torch.randperm(n, generator=g, device=dev)
            bases[half, L // 2:] = bases[perm][half, L // 2:]
            # per-read lengths: a mix of full and shorter reads, offsets with gaps
            length = torch.where(torch.rand(n, generator=g, device=dev) < 0.5, torch.full((n,), L, device=dev),
                                 torch.randint(1, L + 1, (n,), generator=g, device=dev)).to(torch.int32)
            shift = int(rng.integers(0, 4))
            buf = torch.zeros(n * L + shift, dtype=torch.uint8, device=dev)
            buf[shift:] = bases.reshape(-1)
            for rc in (False, True):
                os.environ["UTREE_LANE_PASS"] = "1"
                a = sdb.tree.classify(buf[shift:], reads.off, length, rc=rc)
                name = sdb.tree.kernel_name()
                os.environ["UTREE_LANE_PASS"] = "0"
                b = sdb.tree.classify(buf[shift:], reads.off, length, rc=rc)
                torch.cuda.synchronize()
                sdb.tree.poll()
                if not torch.equal(a, b):
                    bad = torch.nonzero((a != b).any(dim=1)).squeeze(1)
                    print("MISMATCH", dict(W=W, nodes=nodes, L=L, n=n, rc=rc, fine=os.environ["UTREE_FINE_BITS"], bucket=os.environ["UTREE_BUCKET_BYTES"], target=os.environ["UTREE_BUCKET_TARGET"], kernel=name, first=bad[:5].tolist()),
                          a[bad[0]].tolist(), b[bad[0]].tolist())
                    sys.exit(1)
                checked += n
                lanes_used += name.startswith("classify_lanes_k")
        sdb.tree.close()
        del sdb
        torch.cuda.empty_cache()
        print("round %d ok: W=%d nodes=%d fine=%s bucket=%s target=%s (%d reads so far, %d batches through the pass)" % (it, W, nodes, os.environ["UTREE_FINE_BITS"], os.environ["UTREE_BUCKET_BYTES"], os.environ["UTREE_BUCKET_TARGET"], checked, lanes_used), flush=True)
    print("fuzz ok: %d reads, %d batches through the lane-per-read pass" % (checked, lanes_used))


if __name__ == "__main__":
    main()
