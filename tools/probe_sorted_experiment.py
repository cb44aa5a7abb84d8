#!/usr/bin/env python3
"""Experiment (not product): how much faster is the table probe when a batch's k-mers arrive sorted by table
slot (every 128-B line fetched once, sequentially) than in read order (random lines)?  Upper bound for a
radix-partitioned lookup pipeline."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from utree_amd import synth

dev = torch.device("cuda:0")
sdb = synth.make_db(dev, 1_217_000_000)
tree = sdb.tree
for n in (476_000_000, 1_904_000_000):
    g = torch.Generator(device=dev); g.manual_seed(1)
    words = torch.randint(-(1 << 62), (1 << 62), (n,), generator=g, device=dev, dtype=torch.int64) * 2 + 1
    def run(w, tag):
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(3):
            out = tree.get_ix(None, w)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 3
        print("%-22s n=%d  %.2f ms  %.1f G lookups/s" % (tag, n, dt * 1e3, n / dt / 1e9), flush=True)
    run(words, "read order (random)")
    MIN = -(1 << 63)
    t0 = time.time(); s = torch.sort(words ^ MIN).values ^ MIN; torch.cuda.synchronize()
    print("torch.sort of the words: %.1f ms" % ((time.time() - t0) * 1e3))
    run(s, "sorted by slot")
    # partially sorted: bucketed by the top 12 bits only (4096 partitions of 8 MB of table), random inside
    key = (words >> 52) & 0xFFF
    t0 = time.time(); order = torch.argsort(key); p = words[order]; torch.cuda.synchronize()
    run(p, "4096 partitions")
    key = (words >> 56) & 0xFF
    order = torch.argsort(key); p = words[order]; torch.cuda.synchronize()
    run(p, "256 partitions")
    del words, s, p, order, key
