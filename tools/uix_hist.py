import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from utree_amd import synth
dev = torch.device('cuda:0')
sdb = synth.make_db(dev, 300_000_000)
r = synth.make_reads(sdb, 1_000_000, 150)
res = sdb.tree.classify(r.bases, r.off, r.length, rc=False).cpu().numpy().view(np.int32).reshape(-1, 6)
uix = res[:, 3]
print("found>0", (res[:, 2] > 0).mean(), "uix>=2", (uix >= 2).mean(), "hist", np.bincount(np.minimum(uix, 20))[:21].tolist())
print("mean uix among pending", uix[uix >= 2].mean())
