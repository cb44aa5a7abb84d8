import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from utree_amd import synth
dev = torch.device("cuda:0")
sdb = synth.make_db(dev, 72_000_000, W=8)
L = synth.lognormal_lengths(400_000, mean=10000.0, lo=1000, hi=100000, seed=synth.READ_SEED)
r = synth.make_reads_var(sdb, L, seed=synth.READ_SEED, device=dev)
for rc in (False, True):
    tot, mx = int(r.length.sum().item()), int(r.length.max().item())
    ws = torch.empty(sdb.tree.workspace_bytes(r.n, tot, mx, rc), dtype=torch.uint8, device=dev)
    out = sdb.tree.classify(r.bases, r.off, r.length, rc=rc, total_bases=tot, max_len=mx, workspace=ws)
    torch.cuda.synchronize()
    cur = ws[:512].view(torch.int64).cpu().numpy()
    print("rc", rc, "kernel", sdb.tree.kernel_name(), "long (after left_count)", cur[32], "left", cur[24], "pieces", cur[8], "mid-listed", cur[16], "reads", r.n, "mean len", tot / r.n)
