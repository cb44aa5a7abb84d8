"""Seeded synthetic databases and reads, generated on the GPU with torch.

The database is produced in the reference's ON-DISK layout (bin table + packed SZ-byte records + label text:
itree.c:1301-1313) as device tensors and handed to the product through utree_dev_build -- the same path a
`.ctr` file takes after its bytes reach HBM -- so nothing about the file format is bypassed.

    node i  ->  word = mix64(i ^ seed)        (splitmix64 finaliser: a bijection, so words are unique, uniform)
    k = 64  ->  word = mix64(i ^ seed) << 64 | mix64(~i ^ seed)
    labels  ->  8-rank GG strings on a fan-out-4 tree: 16 384 strain leaves + 5 460 interpolated ancestors
                (ranks p..s) = 21 844 labels; file order shuffled by the seed; node i carries label i // B.
    reads   ->  uniform bases with floor(L/k) non-overlapping planted k-mers: 70 % from the read's leaf label,
                20 % from one of its ancestors, 10 % from a sibling strain; 5 % of reads fully random; 1 % carry
                one 'N'.

torch is plumbing here (RNG, sort, byte shuffling); the search itself only ever runs in the HIP kernels.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import lib as _lib
from .search import CtrDB, DeviceTree

DB_SEED = 0x5EEDC0DE
READ_SEED = 0xC0FFEE
MASK64 = (1 << 64) - 1
M1, M2 = 0xBF58476D1CE4E5B9, 0x94D049BB133111EB
M1_INV, M2_INV = pow(M1, -1, 1 << 64), pow(M2, -1, 1 << 64)
N_LEAVES = 4 ** 7
ANC_SIZES = [4 ** m for m in range(1, 7)]                 # ranks p, c, o, f, g, s
ANC_OFFSETS = np.concatenate([[0], np.cumsum(ANC_SIZES)[:-1]]) + N_LEAVES
N_LABELS = N_LEAVES + sum(ANC_SIZES)                       # 21 844


def _s64(x: int) -> int:
    x &= MASK64
    return x - (1 << 64) if x >> 63 else x


def _lsr(x, s: int):
    """logical shift right of an int64 tensor"""
    return (x >> s) & _s64((1 << (64 - s)) - 1)


def mix64(x):
    x = (x ^ _lsr(x, 30)) * _s64(M1)
    x = (x ^ _lsr(x, 27)) * _s64(M2)
    return x ^ _lsr(x, 31)


def unmix64(x):
    x = x ^ _lsr(x, 31) ^ _lsr(x, 62)
    x = x * _s64(M2_INV)
    x = x ^ _lsr(x, 27) ^ _lsr(x, 54)
    x = x * _s64(M1_INV)
    return x ^ _lsr(x, 30) ^ _lsr(x, 60)


def mix64_py(x: int) -> int:
    x &= MASK64
    x = ((x ^ (x >> 30)) * M1) & MASK64
    x = ((x ^ (x >> 27)) * M2) & MASK64
    return x ^ (x >> 31)


def label_strings() -> list:
    """Tree-id order: leaves 0..16383 (7 base-4 digits p..t), then ancestors rank p, c, o, f, g, s."""
    ranks = "kpcofgst"
    names = "KPCOFGST"

    def tokens(digits):
        toks = ["k__K"]
        for m in range(1, len(digits) + 1):
            toks.append("%s__%s%s" % (ranks[m], names[m], "".join(str(d) for d in digits[:m])))
        return toks

    out = []
    for leaf in range(N_LEAVES):
        digits = [(leaf >> (2 * (6 - j))) & 3 for j in range(7)]
        out.append(";".join(tokens(digits)))
    for m in range(1, 7):
        for a in range(4 ** m):
            digits = [(a >> (2 * (m - 1 - j))) & 3 for j in range(m)]
            out.append(";".join(tokens(digits)))
    return out


@dataclass
class SynthDB:
    ctr: CtrDB
    tree: DeviceTree
    n_nodes: int
    W: int
    block: int                      # nodes per label (B)
    seed: int
    tree2file: "object"             # torch int64 [N_LABELS]: tree label id -> file label index
    label_text: bytes
    binix: "object" = None          # torch int32 (as uint32 bits) [2^24+1] on device, if kept
    records: "object" = None        # torch uint8 [N*SZ] on device, if kept
    build_seconds: float = 0.0


def make_db(device, n_nodes: int, seed: int = DB_SEED, W: int = 8, fine_bits: int = _lib.FINE_AUTO,
            keep_raw: bool = False, image=None) -> SynthDB:
    import time
    import torch
    assert W in (8, 16) and n_nodes < 0xFFFFFFFF
    t0 = time.time()
    dev = torch.device(device)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    perm = torch.randperm(N_LABELS, generator=g)                    # tree id -> file index
    labels_tree = label_strings()
    file_labels = [None] * N_LABELS
    for t, f in enumerate(perm.tolist()):
        file_labels[f] = labels_tree[t]
    block = max(1, n_nodes // N_LABELS)
    MIN = _s64(1 << 63)
    if n_nodes < (1 << 31) - 1:
        idx = torch.arange(n_nodes, dtype=torch.int64, device=dev)
        words = mix64(idx ^ _s64(seed))
        del idx
        words = torch.sort(words ^ MIN).values ^ MIN                 # unsigned ascending
    else:
        # torch sorts (and selects by mask) at most 2^31 - 1 elements at a time: the node indices are walked in chunks, every chunk's
        # words go to the eighth of the value range they fall into, and the eighths are sorted one by one
        CH = 1 << 30
        parts = [[] for _ in range(8)]
        for lo in range(0, n_nodes, CH):
            idx = torch.arange(lo, min(n_nodes, lo + CH), dtype=torch.int64, device=dev)
            w = mix64(idx ^ _s64(seed))
            del idx
            top = (w >> 61) & 7                                      # the unsigned value's top three bits
            for q in range(8):
                parts[q].append(w[top == q])
            del w, top
        words = torch.empty(n_nodes, dtype=torch.int64, device=dev)
        at = 0
        for q in range(8):
            e = torch.cat(parts[q])
            parts[q] = None
            e = torch.sort(e ^ MIN).values ^ MIN
            words[at:at + e.numel()] = e
            at += e.numel()
            del e
        assert at == n_nodes
    orig = unmix64(words) ^ _s64(seed)                               # node index of each sorted word
    tree_id = torch.clamp(orig // block, max=N_LABELS - 1)
    ix = perm.to(dev)[tree_id]
    del tree_id
    counts = torch.bincount(ix, minlength=N_LABELS).cpu().numpy()
    prefix_bounds = (torch.arange((1 << 24) + 1, dtype=torch.int64, device=dev) << 40) ^ MIN
    binix = torch.searchsorted(words ^ MIN, prefix_bounds[:-1], right=False)
    binix = torch.cat([binix, torch.tensor([n_nodes], dtype=torch.int64, device=dev)])
    del prefix_bounds
    wb = words.view(torch.uint8).view(n_nodes, 8)
    ixb = ix.to(torch.int16).view(torch.uint8).view(n_nodes, 2)
    if W == 8:
        records = torch.cat([wb[:, :5], ixb], dim=1).contiguous()
    else:
        lo = mix64((~orig) ^ _s64(seed))
        lob = lo.view(torch.uint8).view(n_nodes, 8)
        records = torch.cat([lob, wb[:, :5], ixb], dim=1).contiguous()   # LE 128-bit word: low 8 bytes first
        del lo, lob
    del wb, ixb, ix, orig, words
    binix32 = binix & 0xFFFFFFFF                                      # on-disk u32 entries (same low 32 bits)
    binix32 = torch.where(binix32 >= (1 << 31), binix32 - (1 << 32), binix32).to(torch.int32)
    label_text = b"".join(lab.encode() + b"\t" + str(int(c)).encode() + b"\n" for lab, c in zip(file_labels, counts))
    ctr = CtrDB.from_memory(W, 2, n_nodes, binix.cpu().numpy().astype(np.uint64), None, label_text)
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()                                          # (the image builder allocates with hipMalloc: what torch has freed must be free)
    tree = DeviceTree.build_from_device(ctr, binix32, records.view(-1), device=dev.index or 0, fine_bits=fine_bits,
                                        image=image)
    torch.cuda.synchronize(dev)
    db = SynthDB(ctr=ctr, tree=tree, n_nodes=n_nodes, W=W, block=block, seed=seed, tree2file=perm.to(dev),
                 label_text=label_text)
    if keep_raw:
        db.binix, db.records = binix32, records.view(-1)
    db.build_seconds = time.time() - t0
    return db


@dataclass
class SynthReads:
    bases: "object"       # uint8 [n * L] on device
    off: "object"         # int64 [n]
    length: "object"      # int32 [n]
    n: int
    read_len: int


def make_reads(db: SynthDB, n_reads: int, read_len: int = 150, seed: int = READ_SEED, device=None) -> SynthReads:
    import torch
    dev = db.tree2file.device if device is None else torch.device(device)
    k = 4 * db.W
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    seq = acgt[torch.randint(0, 4, (n_reads, read_len), generator=g, device=dev)]
    slots = read_len // k
    if slots:
        leaf = torch.randint(0, N_LEAVES, (n_reads, 1), generator=g, device=dev)
        u = torch.rand((n_reads, slots), generator=g, device=dev)
        depth = torch.randint(1, 7, (n_reads, slots), generator=g, device=dev)          # ancestor rank p..s
        anc_off = torch.from_numpy(ANC_OFFSETS.astype(np.int64)).to(dev)
        anc = anc_off[depth - 1] + (leaf >> (2 * (7 - depth)))
        sib = leaf ^ torch.randint(1, 4, (n_reads, slots), generator=g, device=dev)
        lab = torch.where(u < 0.7, leaf.expand(-1, slots), torch.where(u < 0.9, anc, sib))
        # a node of that label: labels own node blocks [t*B, (t+1)*B); the last label also owns the tail
        node = lab * db.block + torch.randint(0, db.block, (n_reads, slots), generator=g, device=dev)
        node = torch.clamp(node, max=db.n_nodes - 1)
        hi = mix64(node ^ _s64(db.seed))
        shifts = torch.arange(62, -2, -2, device=dev, dtype=torch.int64)               # first base most significant
        codes = (hi.unsqueeze(-1) >> shifts) & 3
        if db.W == 16:
            lo = mix64((~node) ^ _s64(db.seed))
            codes = torch.cat([codes, (lo.unsqueeze(-1) >> shifts) & 3], dim=-1)
        planted = acgt[codes].view(n_reads, slots * k)
        random_read = torch.rand((n_reads, 1), generator=g, device=dev) < 0.05
        seq[:, : slots * k] = torch.where(random_read, seq[:, : slots * k], planted)
    with_n = torch.rand(n_reads, generator=g, device=dev) < 0.01
    pos = torch.randint(0, read_len, (n_reads,), generator=g, device=dev)
    rows = torch.nonzero(with_n).squeeze(1)
    seq[rows, pos[rows]] = ord("N")
    off = torch.arange(n_reads, dtype=torch.int64, device=dev) * read_len
    length = torch.full((n_reads,), read_len, dtype=torch.int32, device=dev)
    return SynthReads(bases=seq.contiguous().view(-1), off=off, length=length, n=n_reads, read_len=read_len)


def lognormal_lengths(n_reads: int, mean: float = 10_000.0, sigma: float = 1.0, lo: int = 1_000, hi: int = 100_000, seed: int = READ_SEED):
    """read lengths ~ lognormal with the given mean, clipped to [lo, hi] (numpy int32 array)."""
    rng = np.random.default_rng(seed)
    mu = np.log(mean) - 0.5 * sigma * sigma
    return np.clip(rng.lognormal(mu, sigma, n_reads), lo, hi).astype(np.int32)


def make_reads_var(db: SynthDB, lengths, seed: int = READ_SEED, device=None) -> SynthReads:
    """Reads of DIFFERENT lengths with the content rule of make_reads: uniform bases, floor(L/k) non-overlapping planted k-mers (70 %
    from the read's leaf label, 20 % an ancestor's, 10 % a sibling's), 5 % of the reads fully random, 1 % with one 'N'.  Every read's
    storage starts at a multiple of k in the buffer (reads need not be adjacent: offsets say where they are), so the planted k-mers are
    rows of the buffer seen as [rows, k]."""
    import torch
    dev = db.tree2file.device if device is None else torch.device(device)
    k = 4 * db.W
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    ln = torch.as_tensor(np.asarray(lengths, dtype=np.int64), device=dev)
    n = ln.numel()
    rows = (ln + k - 1) // k
    row0 = torch.cumsum(rows, 0) - rows
    total_rows = int(rows.sum().item())
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    seq = torch.empty((total_rows, k), dtype=torch.uint8, device=dev)
    step = 1 << 22
    for lo in range(0, total_rows, step):                                     # (randint in pieces: the int64 draw is 8 x the bytes)
        hi = min(total_rows, lo + step)
        seq[lo:hi] = acgt[torch.randint(0, 4, (hi - lo, k), generator=g, device=dev)]
    slots = ln // k
    total_slots = int(slots.sum().item())
    if total_slots:
        read_of = torch.repeat_interleave(torch.arange(n, device=dev), slots)
        slot0 = torch.cumsum(slots, 0) - slots
        j = torch.arange(total_slots, device=dev) - slot0[read_of]
        leaf_r = torch.randint(0, N_LEAVES, (n,), generator=g, device=dev)
        random_read = torch.rand(n, generator=g, device=dev) < 0.05
        leaf = leaf_r[read_of]
        u = torch.rand(total_slots, generator=g, device=dev)
        depth = torch.randint(1, 7, (total_slots,), generator=g, device=dev)
        anc_off = torch.from_numpy(ANC_OFFSETS.astype(np.int64)).to(dev)
        anc = anc_off[depth - 1] + (leaf >> (2 * (7 - depth)))
        sib = leaf ^ torch.randint(1, 4, (total_slots,), generator=g, device=dev)
        lab = torch.where(u < 0.7, leaf, torch.where(u < 0.9, anc, sib))
        node = torch.clamp(lab * db.block + torch.randint(0, db.block, (total_slots,), generator=g, device=dev), max=db.n_nodes - 1)
        keep = ~random_read[read_of]
        shifts = torch.arange(62, -2, -2, device=dev, dtype=torch.int64)
        target = (row0[read_of] + j)[keep]
        node = node[keep]
        for lo in range(0, node.numel(), step):
            nd = node[lo:lo + step]
            codes = (mix64(nd ^ _s64(db.seed)).unsqueeze(-1) >> shifts) & 3
            if db.W == 16:
                codes = torch.cat([codes, (mix64((~nd) ^ _s64(db.seed)).unsqueeze(-1) >> shifts) & 3], dim=-1)
            seq[target[lo:lo + step]] = acgt[codes]
    off = row0 * k
    with_n = torch.rand(n, generator=g, device=dev) < 0.01
    pos = (torch.rand(n, generator=g, device=dev) * ln).long().clamp(max=int(ln.max().item()) - 1)
    pos = torch.minimum(pos, ln - 1)
    rws = torch.nonzero(with_n).squeeze(1)
    flat = seq.view(-1)
    flat[off[rws] + pos[rws]] = ord("N")
    return SynthReads(bases=flat, off=off.contiguous(), length=ln.to(torch.int32), n=n, read_len=int(ln.max().item()))


def fasta_tensor(reads: SynthReads, first_index: int = 0):
    """The same FASTA text as reads_to_fasta, assembled on the device (uint8 tensor): reads whose index has the same number of
    digits have records of one size, so each such group is one 2-D byte tensor."""
    import torch
    dev = reads.bases.device
    L = reads.read_len
    seq = reads.bases.view(reads.n, L)
    parts = []
    lo = first_index
    end = first_index + reads.n
    while lo < end:
        nd = len(str(lo))
        hi = min(end, 10 ** nd)
        cnt = hi - lo
        rec = torch.empty((cnt, 2 + nd + 1 + L + 1), dtype=torch.uint8, device=dev)
        rec[:, 0] = ord(">")
        rec[:, 1] = ord("r")
        idx = torch.arange(lo, hi, dtype=torch.int64, device=dev)
        for d in range(nd):
            rec[:, 2 + nd - 1 - d] = ((idx // (10 ** d)) % 10 + ord("0")).to(torch.uint8)
        rec[:, 2 + nd] = ord("\n")
        rec[:, 3 + nd: 3 + nd + L] = seq[lo - first_index: hi - first_index]
        rec[:, 3 + nd + L] = ord("\n")
        parts.append(rec.view(-1))
        lo = hi
    return torch.cat(parts) if len(parts) != 1 else parts[0]


def reads_to_fasta(reads: SynthReads, first_index: int = 0) -> bytes:
    """FASTA text (headers `>r<index>`) of a SynthReads batch, for the CLI / CPU baselines."""
    seq = reads.bases.view(reads.n, reads.read_len).cpu().numpy()
    out = bytearray()
    for i in range(reads.n):
        out += b">r%d\n" % (first_index + i)
        out += seq[i].tobytes()
        out += b"\n"
    return bytes(out)


@dataclass
class RelatedDB:
    """A database of RELATED genomes built by the product's own tools (utree-buildGG + xtree-compress), and the references reads are cut from."""
    ctr: CtrDB
    tree: DeviceTree
    kept: "object"            # torch uint8 [keep, ref_len] base codes 0..3 of the first `keep` references
    ref_len: int
    n_nodes: int
    W: int
    seconds: dict


def make_related_db(device, workdir: str, refs: int = 1000, ref_len: int = 1_000_000, complevel: int = 0, seed: int = 11) -> RelatedDB:
    """`refs` references = mutated copies (2 % substitutions) of refs/25 random roots with GG-style 8-rank labels -> utree-buildGG
    (complevel 0: every k-mer) -> xtree-compress -> device image.  k-mers of the ~25 relatives of a root crowd around each minimizer: the
    shape of a real reference database (README.md:2 of the reference), unlike make_db's independent k-mers."""
    import os
    import subprocess
    import time
    import torch
    dev = torch.device(device)
    os.makedirs(workdir, exist_ok=True)
    d = workdir
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    n_roots = max(4, refs // 25)
    roots = torch.randint(0, 4, (n_roots, ref_len), generator=g, device=dev, dtype=torch.uint8)
    ranks = "kpcofgst"
    keep = min(refs, 256)
    kept = torch.empty((keep, ref_len), dtype=torch.uint8, device=dev)
    secs = {}
    t0 = time.time()
    with open(d + "/refs.fa", "wb") as f, open(d + "/refs.map", "wb") as m:
        for i in range(refs):
            r = i % n_roots
            s_ = roots[r].clone()
            mut = torch.rand(ref_len, generator=g, device=dev) < 0.02
            s_[mut] = torch.randint(0, 4, (int(mut.sum()),), generator=g, device=dev, dtype=torch.uint8)
            if i < keep:
                kept[i] = s_
            f.write(b">ref%06d\n" % i)
            acgt[s_.long()].cpu().numpy().tofile(f)
            f.write(b"\n")
            path = [r % 2, r % 3, r % 5, r % 7, r % 11, r, i % 9, i]
            m.write(b"ref%06d\t" % i + ";".join("%s__%d" % (ranks[k], path[k]) for k in range(8)).encode() + b"\n")
    secs["generate_refs"] = time.time() - t0
    del roots

    def run(cmd):
        t = time.time()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            raise RuntimeError("%s failed: %s" % (cmd[0], r.stderr.decode()[-500:]))
        return time.time() - t, r
    secs["build"], r = run([_lib.BUILD_GG_CLI_PATH, d + "/refs.fa", d + "/refs.map", d + "/db.ubt", "0", str(complevel)])
    secs["compress"], r = run([_lib.COMPRESS_CLI_PATH, d + "/db.ubt", d + "/db.ctr"])
    os.remove(d + "/db.ubt")
    os.remove(d + "/refs.fa")
    db = CtrDB.open(d + "/db.ctr")
    t0 = time.time()
    tree = DeviceTree.upload(db, dev.index or 0)
    torch.cuda.synchronize(dev)
    secs["upload"] = time.time() - t0
    return RelatedDB(ctr=db, tree=tree, kept=kept, ref_len=ref_len, n_nodes=int(db.n_nodes), W=int(db.W), seconds=secs)


def make_related_reads(rdb: RelatedDB, n_reads: int, read_len: int = 150, seed: int = 100) -> SynthReads:
    """Reads cut from the kept references: 1 % substitutions, a quarter reverse-complemented -- they hit in most of their windows."""
    import torch
    dev = rdb.kept.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    L = read_len
    keep = rdb.kept.shape[0]
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    comp = torch.tensor([3, 2, 1, 0], dtype=torch.uint8, device=dev)
    which = torch.randint(0, keep, (n_reads,), generator=g, device=dev)
    pos = torch.randint(0, rdb.ref_len - L, (n_reads,), generator=g, device=dev)
    idx = (which * rdb.ref_len + pos).unsqueeze(1) + torch.arange(L, device=dev).unsqueeze(0)
    s_ = rdb.kept.view(-1)[idx]
    mm = torch.rand((n_reads, L), generator=g, device=dev) < 0.01
    s_ = torch.where(mm, torch.randint(0, 4, (n_reads, L), generator=g, device=dev, dtype=torch.uint8), s_)
    rcm = (torch.arange(n_reads, device=dev) & 3) == 0
    s_ = torch.where(rcm.unsqueeze(1), comp[s_.flip(1).long()], s_)
    off = torch.arange(n_reads, dtype=torch.int64, device=dev) * L
    ln = torch.full((n_reads,), L, dtype=torch.int32, device=dev)
    return SynthReads(bases=acgt[s_.long()].contiguous().view(-1), off=off, length=ln, n=n_reads, read_len=L)
