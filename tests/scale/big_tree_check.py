#!/usr/bin/env python3
"""A `.ctr` with MORE than 2^32-1 nodes (the reference then stores 8-byte bin-table entries, itree.c:756-759) at real scale:
write it as a file, classify reads with OUR command line and with the genuine reference, compare the outputs.
This is the only way to exercise the 8-byte bin-table reader and the 64-bit offset kernels on a genuine file
(`UTREE_FORCE_OFF64` covers the kernels on small trees).  Needs ~30 GB of /dev/shm and ~150 GB of HBM.
usage: big_tree_check.py [nodes] [reads]"""
import hashlib, json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from utree_amd import lib, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_400_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
d = "/dev/shm/utree_big"; os.makedirs(d, exist_ok=True)
dev = torch.device("cuda:0")
out = {"nodes": N, "reads": R}
t0 = time.time()
MIN = synth._s64(1 << 63)
NL = 1000
labels = ["k__B%d;p__P%d;c__C%d;o__O%d" % (i % 2, i % 5, i % 25, i) for i in range(NL)]
ctr = d + "/big.ctr"
with open(ctr, "wb") as f:
    f.write(np.array([8, 0, 2, N], dtype="<u8").tobytes())
    f.write(b"\0" * (8 * ((1 << 24) + 1)))                               # 8-byte bin table (N >= UINT32_MAX), filled in below
    # ascending by construction: word_i = i * stride + 31 pseudo-random bits.  Everything is done in slices of 2^28
    # nodes: this torch build does not get element-wise kernels or searchsorted right beyond 2^31 elements.
    stride = (1 << 64) // N
    assert stride > (1 << 31)
    bounds = (torch.arange(1 << 24, dtype=torch.int64, device=dev) << 40) ^ MIN
    binix = torch.zeros(1 << 24, dtype=torch.int64, device=dev)
    counts = torch.zeros(NL, dtype=torch.int64, device=dev)
    step = 1 << 28
    sample = []
    last = None
    for a in range(0, N, step):
        idx = torch.arange(a, min(N, a + step), dtype=torch.int64, device=dev)
        w = idx * synth._s64(stride) + (synth.mix64(idx ^ 12345) & 0x7FFFFFFF)      # wraps like uint64
        ws = w ^ MIN
        assert bool((ws[1:] > ws[:-1]).all()) and (last is None or int(ws[0]) > last)
        last = int(ws[-1])
        binix += torch.searchsorted(ws.contiguous(), bounds, right=False)
        ix = (synth.mix64(w) >> 20) % NL
        ix = torch.where(ix < 0, ix + NL, ix)
        counts += torch.bincount(ix, minlength=NL)
        wb = w.contiguous().view(torch.uint8).view(-1, 8)
        ixb = ix.to(torch.int16).view(torch.uint8).view(-1, 2)
        f.write(torch.cat([wb[:, :5], ixb], dim=1).contiguous().cpu().numpy().tobytes())
        sample.append(w[torch.randint(0, w.numel(), (R // (N // step + 1) + 1,), device=dev)].cpu())
        del idx, w, ws, ix, wb, ixb
    f.write(b"".join(l.encode() + b"\t%d\n" % int(c) for l, c in zip(labels, counts.cpu().tolist())))
    table = torch.cat([binix, torch.tensor([N], dtype=torch.int64, device=dev)]).cpu().numpy().astype("<u8")
    assert (np.diff(table.astype(np.int64)) >= 0).all() and N // 3 < int(table[1 << 23]) < 2 * N // 3
    out["bin_table_sample"] = [int(table[i]) for i in (0, 1, 2, 1 << 23, (1 << 24) - 1, 1 << 24)]
    f.seek(32)
    f.write(table.tobytes())
out["ctr_bytes"] = os.path.getsize(ctr)
# reads: a database k-mer (both ends of the tree are sampled) + 118 random bases, some with a second k-mer
sample = torch.cat(sample)[:R].numpy().astype(np.uint64)
rng = np.random.default_rng(3)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
sh = (np.arange(31, -1, -1, dtype=np.uint64) * np.uint64(2))
km = ACGT[((sample[:, None] >> sh[None, :]) & np.uint64(3)).astype(np.int64)]
rest = ACGT[rng.integers(0, 4, (R, 118))]
second = ACGT[((np.roll(sample, 1)[:, None] >> sh[None, :]) & np.uint64(3)).astype(np.int64)]
rest[: R // 2, 40:72] = second[: R // 2]
seqs = np.concatenate([km, rest], axis=1)
fa = d + "/reads.fa"
with open(fa, "wb") as f:
    for i in range(R):
        f.write(b">q%d\n" % i); f.write(seqs[i].tobytes()); f.write(b"\n")
out["generate_seconds"] = time.time() - t0
del counts
torch.cuda.empty_cache()

if os.environ.get("GENERATE_ONLY"):
    print(json.dumps(out)); sys.exit(0)

def run(cmd):
    t = time.time(); r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE); return time.time() - t, r
def sorted_sha(p):
    l = open(p, "rb").read().split(b"\n"); l.sort(); return hashlib.sha256(b"\n".join(l)).hexdigest(), len(l) - 1
t, r = run([lib.CLI_PATH, ctr, fa, d + "/ours.txt", "16"])
out["ours"] = {"exit": r.returncode, "seconds": t, "stdout": [x for x in r.stdout.decode().splitlines() if "counters" in x or "Nodes" in x or "Good" in x],
               "stderr": r.stderr.decode().strip().splitlines()[-3:]}
if r.returncode == 0:
    out["ours"]["sorted_sha256"], out["ours"]["lines"] = sorted_sha(d + "/ours.txt")
ref = os.path.join(os.path.dirname(lib.SO_PATH), "..", "oracle", "_ref", "xtree-searchGG")
if os.path.exists(ref) and not os.environ.get("SKIP_REFERENCE"):
    t, r = run([ref, ctr, fa, d + "/ref.txt", "16"])
    out["reference"] = {"exit": r.returncode, "seconds": t, "stdout": [x for x in r.stdout.decode().splitlines() if "counters" in x or "smokes" in x or "Nodes" in x or "Good" in x]}
    if r.returncode == 0:
        out["reference"]["sorted_sha256"], out["reference"]["lines"] = sorted_sha(d + "/ref.txt")
        out["parity_sorted_lines_identical"] = out["reference"]["sorted_sha256"] == out["ours"].get("sorted_sha256")
print(json.dumps(out, indent=1))
for f in os.listdir(d): os.remove(os.path.join(d, f))
