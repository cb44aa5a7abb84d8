"""The lane-per-read pass (csrc/lanes_kernel.hip: 64 reads per wavefront, k = 32 / u16 labels / forward strand / reads up to 160
bases) against the CPU oracle and against the wave-per-read kernels (UTREE_LANE_PASS=0) on the same inputs, with the cases that
decide which reads it keeps: one bad base (handled in the kernel), several (left to the wave-per-read kernel), reads shorter
than a window, the longest reads it holds, batches that are not a multiple of 64, buckets that overflow, reads with more hits
than a lane's slot holds, hit-dense reads, reads with more labels than its tally table, and a workload on which it gives up for good.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import orc
from utree_amd import ctrfile
from utree_amd.search import CtrDB, DeviceTree, classify_fasta_bytes
import util
from test_gpu_parity import fasta_bytes, random_reads, tree_for


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def oracle_text(o, data, tmp_path, threads=8, rc=False):
    fa = tmp_path / "r.fa"
    fa.write_bytes(data)
    out = tmp_path / "o.txt"
    code, nr, good, err = o.search_file(str(fa), str(out), threads=threads, rc=rc)
    assert code == 0
    return out.read_bytes()


class OwnDB:
    """60 000 random k-mers under a four-rank label tree, written with the package's own .ctr writer (every bin regular)."""

    def __init__(self, tmp_path, seed=3, k=32):
        self.k = k
        rng = np.random.default_rng(seed)
        lo = rng.integers(0, 1 << 63, 60_000, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 60_000, dtype=np.uint64)
        hi = (rng.integers(0, 1 << 63, 60_000, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 60_000, dtype=np.uint64)) if k == 64 \
            else np.zeros(60_000, np.uint64)
        order = np.lexsort((lo, hi))
        keep = np.ones(len(order), bool)
        keep[1:] = (hi[order][1:] != hi[order][:-1]) | (lo[order][1:] != lo[order][:-1])
        self.hi, self.lo = hi[order][keep], lo[order][keep]
        labels = ["k__A;p__P%d;c__C%d;o__O%d" % (a, b, c) for a in range(3) for b in range(3) for c in range(4)]
        labels += ["k__A;p__P%d;c__C%d" % (a, b) for a in range(3) for b in range(3)] + ["k__A;p__P%d" % a for a in range(3)]
        ix = rng.integers(0, len(labels), len(self.lo)).astype(np.uint32)
        self.ctr = str(tmp_path / ("own%d.ctr" % k))
        ctrfile.write_ctr(self.ctr, k // 4, 2, self.hi, self.lo, ix, labels)

    def words(self):
        return self.hi, self.lo


@pytest.mark.parametrize("name", ["own", "own64", "vote", "kat", "toy", "k64", "katq2"])
def test_mixed_reads_lanes_vs_wave_per_read_vs_oracle(torch_cuda, name, tmp_path, monkeypatch):
    if name.startswith("own"):
        d = OwnDB(tmp_path, k=64 if name == "own64" else 32)
        db = CtrDB.open(d.ctr)
        tree = DeviceTree.upload(db, 0)
        o = orc.OracleDB.load(d.ctr)
    else:
        d = util.load_db_fixture(name)
        db, tree = tree_for(name)
        o = orc.OracleDB.load(util.fixture_ctr(name))
    assert not tree.info.generic_mode
    rng = np.random.default_rng(77)
    reads = random_reads(rng, d, 3001, 1, 160) + random_reads(rng, d, 500, 150, 150, hit_frac=0.9)
    hi, lo = d.words()
    some = [ctrfile.decode_kmer(int(hi[j]), int(lo[j]), d.k) for j in rng.integers(0, len(lo), 40)]
    rnd = lambda n: "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))
    reads += [("short%d" % L, rnd(L)) for L in (1, 15, 16, 31, 63)]                                   # no window at all
    reads += [("k%d" % i, s) for i, s in enumerate(some[:10])]                                    # exactly one window
    reads += [("full%d" % i, (s + rnd(128))[:160]) for i, s in enumerate(some[10:20])]           # the longest this pass holds
    reads += [("tail%d" % i, (rnd(128) + s)[-160:]) for i, s in enumerate(some[20:30])]          # hit in the last window
    reads += [("n_first", "N" + some[30] + rnd(50)), ("n_last", rnd(50) + some[31] + "N"), ("n_mid", (some[32] + "N" + some[33])[:160]),
              ("n_two", (some[34] + "N" + rnd(5) + "n" + some[35])[:160]), ("n_many", ("NNNN" + some[36] + "NN" + some[37])[:160]),
              ("x_other", (some[38] + "-" + some[39])[:160]), ("all_n", "N" * 100), ("lower", (some[0] + rnd(30) + some[1]).lower()[:160])]
    reads = [reads[i] for i in rng.permutation(len(reads))]
    data = fasta_bytes(reads)
    # reverse complements of database k-mers: hits on the second strand only
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    reads += [("rev%d" % i, (rnd(int(rng.integers(0, 48))) + s.translate(comp)[::-1] + rnd(int(rng.integers(0, 48))))[:160]) for i, s in enumerate(some)]
    data = fasta_bytes(reads)
    for rc in (False, True):
        monkeypatch.setenv("UTREE_LANE_PASS", "1")
        got = classify_fasta_bytes(db, tree, data, rc=rc)
        assert tree.kernel_name().startswith("classify_lanes_")
        monkeypatch.setenv("UTREE_LANE_PASS", "0")
        plain = classify_fasta_bytes(db, tree, data, rc=rc)
        assert "classify_short_k" in tree.kernel_name()
        assert got == plain
        assert got == oracle_text(o, data, tmp_path, rc=rc)


@pytest.mark.parametrize("name,rc", [("vote", 0), ("kat", 0), ("toy", 0), ("toy", 1), ("k64", 0), ("k64", 1), ("katq", 0), ("katq2", 0), ("ix32", 0), ("ix32", 1)])
def test_reference_golden_lines_through_the_lane_pass(torch_cuda, name, rc, monkeypatch):
    """The genuine reference's committed output (tests/golden) for the fixture reads this pass takes, k = 32 and 64, both strand
    modes.  toy / k64 / katq2 carry COMPRESS' first-bin quirk: one or two irregular bins, whose reads the pass leaves to the
    wave-per-read kernel's exact-probe path."""
    db, tree = tree_for(name)
    assert not tree.info.generic_mode and tree.info.irregular_bins <= 4
    lines = util.fixture_bytes(util.READS_OF.get(name, name) + "_reads.fa.gz").split(b"\n")
    recs = [(lines[i], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]
    keep = [(h, q) for h, q in recs if len(q.rstrip(b"\r")) <= (1055 if rc else 2095 if db.k == 32 else 1615)]
    assert len(keep) > 100
    names = {h[1:].split(b" ")[0] for h, q in keep}
    out = util.fixture_bytes("%s_out%s.txt.gz" % (name, "_rc" if rc else ""))
    want = b"".join(l + b"\n" for l in out.split(b"\n") if l and l.split(b"\t")[0] in names)
    monkeypatch.setenv("UTREE_LANE_PASS", "1")
    got = classify_fasta_bytes(db, tree, b"".join(h + b"\n" + q + b"\n" for h, q in keep), rc=bool(rc))
    assert tree.kernel_name().startswith(("classify_lanes_k<%d, %d," % (db.k // 4, db.I), "classify_lanes_mixed_k<%d, %d," % (db.k // 4, db.I)))     # (ix32: the u32-label instantiation; reads beyond 160 bases: the classes in one launch)
    assert got == want and len(want) > 0


@pytest.mark.parametrize("k,max_len", [(32, 161), (32, 250), (32, 289), (32, 290), (32, 301), (32, 547), (32, 548), (32, 1063), (32, 1064), (32, 2095),
                                       (64, 257), (64, 258), (64, 451), (64, 839), (64, 1615)])
def test_longer_reads_take_two_to_sixteen_lanes(torch_cuda, k, max_len, tmp_path, monkeypatch):
    """A lane holds 160 bases = 129 windows (k = 64: 97); longer reads are cut into such pieces, one lane each, up to sixteen."""
    d = OwnDB(tmp_path, seed=8, k=k)
    db = CtrDB.open(d.ctr)
    tree = DeviceTree.upload(db, 0)
    o = orc.OracleDB.load(d.ctr)
    rng = np.random.default_rng(max_len)
    nr = 1200 if max_len <= 600 else 400
    reads = random_reads(rng, d, nr, 1, max_len, hit_frac=0.7) + random_reads(rng, d, nr // 4, max_len, max_len, hit_frac=0.9)
    hi, lo = d.words()
    some = [ctrfile.decode_kmer(int(hi[j]), int(lo[j]), d.k) for j in rng.integers(0, len(lo), 60)]
    rnd = lambda n: "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))
    # a database k-mer across every piece boundary (window 128 / 129 of k = 32, 96 / 97 of k = 64, and their multiples), one at the very end
    sw = 160 - k + 1
    for i, s in enumerate(some):
        at = (1 + i % 15) * sw - (i % 7)
        r = (rnd(max(0, at)) + s + rnd(max_len))[:max_len]
        reads.append(("edge%d" % i, r))
        reads.append(("end%d" % i, (rnd(max_len) + s)[-max_len:]))
    reads = [reads[i] for i in rng.permutation(len(reads))]
    data = fasta_bytes(reads)
    for rc in (False, True):
        monkeypatch.setenv("UTREE_LANE_PASS", "1")
        got = classify_fasta_bytes(db, tree, data, rc=rc)
        if 2 * max_len + 1 <= 2112 or not rc:
            assert tree.kernel_name().startswith("classify_lanes_")
        monkeypatch.setenv("UTREE_LANE_PASS", "0")
        assert got == classify_fasta_bytes(db, tree, data, rc=rc)
        assert got == oracle_text(o, data, tmp_path, rc=rc)
    tree.close()


def test_batch_sizes_around_the_grab_of_64(torch_cuda, tmp_path, monkeypatch):
    d = OwnDB(tmp_path, seed=4)
    db = CtrDB.open(d.ctr)
    tree = DeviceTree.upload(db, 0)
    o = orc.OracleDB.load(d.ctr)
    rng = np.random.default_rng(5)
    reads = random_reads(rng, d, 300, 32, 160, hit_frac=0.8)
    monkeypatch.setenv("UTREE_LANE_PASS", "1")
    for n in (1, 2, 63, 64, 65, 127, 129, 300):
        data = fasta_bytes(reads[:n])
        assert classify_fasta_bytes(db, tree, data, rc=False) == oracle_text(o, data, tmp_path), n
        assert tree.kernel_name().startswith("classify_lanes_")
    tree.close()


def genome_db(tmp_path, rng, n_contigs=12, contig=6000):
    """Every 32-mer of a few random contigs, one label per contig (plus shared ancestors): reads cut from them hit in every window."""
    contigs = ["".join("ACGT"[int(x)] for x in rng.integers(0, 4, contig)) for _ in range(n_contigs)]
    kmers, labs = {}, []
    for ci, s in enumerate(contigs):
        labs.append("k__B;p__P%d;c__C%d;o__O%d" % (ci % 2, ci % 4, ci))
        for i in range(len(s) - 31):
            kmers.setdefault(s[i:i + 32], ci)
    ks = sorted(kmers)
    hi, lo = ctrfile.encode_kmers(ks)
    ix = np.array([kmers[x] for x in ks], dtype=np.uint32)
    order = np.lexsort((lo, hi))
    ctr = str(tmp_path / "genome.ctr")
    ctrfile.write_ctr(ctr, 8, 2, hi[order], lo[order], ix[order], labs)
    return ctr, contigs


def related_db(tmp_path, rng, label_bytes, roots=3, length=3000, relatives=40, k=32):
    """Every 32-mer of `relatives` mutated copies (3 % substitutions, every fifth 10 %) of a few random roots: the k-mers of the copies
    crowd around each minimizer -- hundreds of records of one hash value, a HEAVY overflow run."""
    seqs, kmers, labs = [], {}, []
    for r in range(roots):
        root = rng.integers(0, 4, length)
        for c in range(relatives):
            s = root.copy()
            mut = rng.random(length) < (0.10 if c % 5 == 4 else 0.03)
            s[mut] = rng.integers(0, 4, int(mut.sum()))
            seqs.append("".join("ACGT"[int(x)] for x in s))
            labs.append("k__B;p__P%d;c__C%d;o__O%d" % (r, c % 4, r * relatives + c))
    for ci, s in enumerate(seqs):
        for i in range(len(s) - k + 1):
            kmers.setdefault(s[i:i + k], ci)
    ks = sorted(kmers)
    hi, lo = ctrfile.encode_kmers(ks)
    ix = np.array([kmers[x] for x in ks], dtype=np.uint32)
    order = np.lexsort((lo, hi))
    ctr = str(tmp_path / ("related%d_%d.ctr" % (k, label_bytes)))
    ctrfile.write_ctr(ctr, k // 4, label_bytes, hi[order], lo[order], ix[order], labs)
    return ctr, seqs


@pytest.mark.parametrize("label_bytes", [2, 4])
def test_heavy_overflow_runs_as_chains(torch_cuda, tmp_path, monkeypatch, label_bytes):
    """k = 32, opt-in UTREE_OVF_CHAINS=1: a heavy run is stored as chains of consecutive k-mers (image flag UTREE_F_OVF_CHAINS, device_common.hpp).
    Reads cut from related genomes (1 % errors, some reverse-complemented) and random reads, both strand modes, through the lane pass and through
    the wave-per-read kernels, against the oracle; the default image (records behind a position directory) gives the same bytes."""
    rng = np.random.default_rng(23 + label_bytes)
    ctr, seqs = related_db(tmp_path, rng, label_bytes)
    db = CtrDB.open(ctr)
    o = orc.OracleDB.load(ctr)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads = []
    for i in range(6000):
        s = seqs[int(rng.integers(0, len(seqs)))]
        p = int(rng.integers(0, len(s) - 150))
        r = list(s[p:p + 150])
        for x in np.nonzero(rng.random(150) < 0.01)[0]:
            r[int(x)] = "ACGT"[int(rng.integers(0, 4))]
        r = "".join(r)
        if i % 4 == 0:
            r = "".join(comp[c] for c in reversed(r))
        reads.append(("h%d" % i, r))
    data = fasta_bytes(reads + [("sparse%d" % i, "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 150))) for i in range(500)])
    want = {rc: oracle_text(o, data, tmp_path, rc=rc) for rc in (False, True)}
    assert want[False].count(b"\n") > 4000                                           # the reads do hit
    monkeypatch.setenv("UTREE_OVF_CHAINS", "1")
    tree = DeviceTree.upload(db, 0)
    assert tree.info.overflow_chains == 1, "the image carries no chains"
    n_ovf = tree.info.overflow_bytes
    for lane_pass in ("1", "0"):
        monkeypatch.setenv("UTREE_LANE_PASS", lane_pass)
        for rc in (False, True):
            got = classify_fasta_bytes(db, tree, data, rc=rc)
            assert tree.kernel_name().startswith("classify_lanes_" if lane_pass == "1" else "classify_short_k"), tree.kernel_name()
            assert got == want[rc], (lane_pass, rc)
    tree.close()
    monkeypatch.delenv("UTREE_OVF_CHAINS")
    tree = DeviceTree.upload(db, 0)
    assert tree.info.overflow_chains == 0 and tree.info.overflow_bytes > n_ovf                   # records take more room than chains
    monkeypatch.setenv("UTREE_LANE_PASS", "1")
    for rc in (False, True):
        assert classify_fasta_bytes(db, tree, data, rc=rc) == want[rc]
    tree.close()


@pytest.mark.parametrize("label_bytes", [2, 4])
def test_related_genomes_k64(torch_cuda, tmp_path, monkeypatch, label_bytes):
    """k = 64 databases of related genomes: heavy overflow runs behind a position directory, with u16 labels (the lane pass and the wave-per-read
    kernels) and with u32 labels (32-byte records: the wave-per-read kernels only, README.md:87-88 of the reference allows the combination), both
    strand modes, against the oracle."""
    rng = np.random.default_rng(64 + label_bytes)
    ctr, seqs = related_db(tmp_path, rng, label_bytes, roots=2, length=2500, relatives=40, k=64)
    db = CtrDB.open(ctr)
    o = orc.OracleDB.load(ctr)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads = []
    for i in range(4000):
        s = seqs[int(rng.integers(0, len(seqs)))]
        p = int(rng.integers(0, len(s) - 150))
        r = list(s[p:p + 150])
        for x in np.nonzero(rng.random(150) < 0.005)[0]:
            r[int(x)] = "ACGT"[int(rng.integers(0, 4))]
        r = "".join(r)
        if i % 4 == 0:
            r = "".join(comp[c] for c in reversed(r))
        reads.append(("h%d" % i, r))
    data = fasta_bytes(reads + [("sparse%d" % i, "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 150))) for i in range(300)])
    want = {rc: oracle_text(o, data, tmp_path, rc=rc) for rc in (False, True)}
    assert want[False].count(b"\n") > 2000
    tree = DeviceTree.upload(db, 0)
    assert tree.info.lane_pass == (1 if label_bytes == 2 else 0)
    for lane_pass in ("1", "0"):
        monkeypatch.setenv("UTREE_LANE_PASS", lane_pass)
        for rc in (False, True):
            assert classify_fasta_bytes(db, tree, data, rc=rc) == want[rc], (lane_pass, rc)
    tree.close()


def test_hit_dense_reads_and_reads_with_more_labels_than_the_tally_table(torch_cuda, tmp_path, monkeypatch):
    """Reads cut from the database's own contigs hit in every window (119 hits, buckets that overflow): tallied in the kernel.  Reads
    pieced together from 60 contigs carry more distinct labels than a read's tally table has slots (48): those are left to the
    wave-per-read kernel, and a workload made of them turns the pass off for the handle."""
    rng = np.random.default_rng(11)
    ctr, contigs = genome_db(tmp_path, rng, n_contigs=72, contig=3000)
    db = CtrDB.open(ctr)
    tree = DeviceTree.upload(db, 0)
    o = orc.OracleDB.load(ctr)
    monkeypatch.setenv("UTREE_LANE_PASS", "1")

    def cut(n):
        out = []
        for i in range(n):
            a, b = contigs[int(rng.integers(0, len(contigs)))], contigs[int(rng.integers(0, len(contigs)))]
            p, q = int(rng.integers(0, len(a) - 150)), int(rng.integers(0, len(b) - 150))
            s = a[p:p + 150] if i % 3 else a[p:p + 75] + b[q:q + 75]                 # a third are chimeras: two labels, a vote
            out.append(("g%d" % i, s))
        return out

    def patchwork(n):                                                                # 60 x 32 bases, each from another contig
        out = []
        for i in range(n):
            parts = []
            for c in rng.permutation(len(contigs))[:60]:
                p = int(rng.integers(0, len(contigs[c]) - 32))
                parts.append(contigs[c][p:p + 32])
            out.append(("w%d" % i, "".join(parts)))
        return out
    for rc in (False, True):
        data = fasta_bytes(cut(4000) + [("sparse%d" % i, "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 150))) for i in range(500)])
        got = classify_fasta_bytes(db, tree, data, rc=rc)
        assert tree.kernel_name().startswith("classify_lanes_")
        assert got == oracle_text(o, data, tmp_path, rc=rc)
    mixed = fasta_bytes(cut(300) + patchwork(300))
    got = classify_fasta_bytes(db, tree, mixed, rc=False)
    assert tree.kernel_name().startswith("classify_lanes_mixed_k<8, 2,")   # 60 x 32 = 1920 bases: sixteen lanes per read, the batch's other reads one -- one launch for both
    assert got == oracle_text(o, mixed, tmp_path)
    big = fasta_bytes(patchwork(30_000))
    for _ in range(10):                                                              # > 256 Ki reads, all of them left over
        classify_fasta_bytes(db, tree, big, rc=False)
    assert "classify_short_k" in tree.kernel_name()                                 # the pass sits such batches out ...
    assert classify_fasta_bytes(db, tree, mixed, rc=False) == got
    # ... but keeps probing (every eighth batch), and a batch it does well on brings it back: no latch for the handle's lifetime
    easy = fasta_bytes(cut(4000))
    want_easy = oracle_text(o, easy, tmp_path)
    names = []
    for _ in range(12):
        assert classify_fasta_bytes(db, tree, easy, rc=False) == want_easy
        names.append(tree.kernel_name())
    assert any("classify_short_k" in x for x in names[:4]) and all(x.startswith("classify_lanes_") for x in names[-3:]), names
    tree.close()


def test_synthetic_config2_shape_lanes_equals_wave_per_read(torch_cuda, monkeypatch):
    """40 M-node synthetic database (the bench generator), 300 000 x 150 bp reads: records identical with the pass on and off."""
    torch = torch_cuda
    from utree_amd import synth
    sdb = synth.make_db(torch.device("cuda:0"), 40_000_000, W=8)
    reads = synth.make_reads(sdb, 300_001, 150, seed=synth.READ_SEED + 9)
    for rc in (False, True):
        monkeypatch.setenv("UTREE_LANE_PASS", "1")
        a = sdb.tree.classify(reads.bases, reads.off, reads.length, rc=rc)
        assert sdb.tree.kernel_name().startswith("classify_lanes_")
        monkeypatch.setenv("UTREE_LANE_PASS", "0")
        b = sdb.tree.classify(reads.bases, reads.off, reads.length, rc=rc)
        assert torch.equal(a, b)
        assert int((a[:, 2] > 0).sum()) > 0.9 * 300_001
    sdb.tree.close()


@pytest.mark.parametrize("k", [32, 64])
def test_batch_of_mixed_lengths_is_split_by_lanes_per_read(torch_cuda, k, tmp_path, monkeypatch):
    """One file holds any mix of lengths (itree.c:866-901).  A batch with 150 bp reads, a few hundred longer ones and some of 3-12 kb is
    routed on the device: every read goes through the instantiation for the lanes it needs (1, 2, 4, 8, 16), the longest in pieces --
    the 150 bp reads do not fall back to the wave-per-read kernel because a long read shares their batch."""
    d = OwnDB(tmp_path, seed=21, k=k)
    db = CtrDB.open(d.ctr)
    tree = DeviceTree.upload(db, 0)
    o = orc.OracleDB.load(d.ctr)
    rng = np.random.default_rng(k)
    reads = (random_reads(rng, d, 3000, 1, 160, hit_frac=0.6) + random_reads(rng, d, 300, 161, 600, hit_frac=0.7) +
             random_reads(rng, d, 120, 601, 2095, hit_frac=0.7) + random_reads(rng, d, 24, 3000, 12000, hit_frac=0.8))
    reads = [reads[i] for i in rng.permutation(len(reads))]
    for sub in (reads, [r for r in reads if len(r[1]) <= 600], [r for r in reads if len(r[1]) <= 160 or len(r[1]) >= 3000]):
        data = fasta_bytes(sub)
        for rc in (False, True):
            monkeypatch.setenv("UTREE_LANE_PASS", "1")
            got = classify_fasta_bytes(db, tree, data, rc=rc)
            assert tree.kernel_name().startswith("classify_lanes_")          # (..._mixed_k: the classes in one launch; ..._k<.., 16, .., 2, ..>: the pieces of the long reads)
            monkeypatch.setenv("UTREE_LANE_PASS", "0")
            assert got == classify_fasta_bytes(db, tree, data, rc=rc)
            assert got == oracle_text(o, data, tmp_path, rc=rc)
    tree.close()


def test_long_reads_with_more_labels_than_their_tables_hold(torch_cuda, tmp_path, monkeypatch):
    """A long read's pieces add their tallies to a 64-slot table of the read in HBM; a piece keeps 48 labels.  Reads pieced together from
    72 contigs (72 labels) overflow one or the other: they are flagged and finished by classify_long_k, the others by finish_long_k."""
    rng = np.random.default_rng(5)
    ctr, contigs = genome_db(tmp_path, rng, n_contigs=72, contig=3000)
    db = CtrDB.open(ctr)
    tree = DeviceTree.upload(db, 0)
    o = orc.OracleDB.load(ctr)

    def patch(n_parts, part, pool):
        s = []
        for c in rng.permutation(pool)[:n_parts]:
            p = int(rng.integers(0, len(contigs[c]) - part))
            s.append(contigs[c][p:p + part])
        return "".join(s)
    reads = [("many%d" % i, patch(72, 140, 72)) for i in range(40)]                    # 10 080 bases, 72 labels: beyond the read's table
    reads += [("piece%d" % i, patch(60, 40, 72) + patch(10, 500, 10)) for i in range(40)]   # 60 labels inside the first piece (2 064 windows)
    reads += [("few%d" % i, patch(30, 300, 30)) for i in range(40)]                    # 9 000 bases, 30 labels: stays in the pass
    reads += [("short%d" % i, patch(1, 150, 72)) for i in range(500)]
    reads = [reads[i] for i in rng.permutation(len(reads))]
    data = fasta_bytes(reads)
    for rc in (False, True):
        monkeypatch.setenv("UTREE_LANE_PASS", "1")
        got = classify_fasta_bytes(db, tree, data, rc=rc)
        assert tree.kernel_name().startswith("classify_lanes_k<8, 2, 16, false, 2,")
        want = oracle_text(o, data, tmp_path, rc=rc)
        assert got == want
        uix = {l.split(b"\t")[0]: int(l.split(b"\t")[3]) for l in want.split(b"\n") if l}
        assert max(v for n, v in uix.items() if n.startswith(b"many")) > 64 and min(v for n, v in uix.items() if n.startswith(b"few")) <= 48
    tree.close()


def test_a_workspace_too_small_is_reported_not_overrun(torch_cuda, tmp_path, monkeypatch):
    """The kernels reserve (rank, count) list space from a cursor; the workspace's size makes an overrun impossible.  With the test
    hook that shrinks the capacity the kernels raise the batch's error word: utree_classify_poll turns it into UTREE_E_DEVICE, nothing
    is written past the buffer, and the next batch (hook off) is fine."""
    import torch
    from utree_amd import lib
    d = OwnDB(tmp_path, seed=4)
    db = CtrDB.open(d.ctr)
    tree = DeviceTree.upload(db, 0)
    o = orc.OracleDB.load(d.ctr)
    rng = np.random.default_rng(6)
    data = fasta_bytes(random_reads(rng, d, 100_000, 150, 150, hit_frac=0.9))
    want = oracle_text(o, data, tmp_path)
    for lane_pass in ("1", "0"):
        monkeypatch.setenv("UTREE_LANE_PASS", lane_pass)
        monkeypatch.setenv("UTREE_TEST_TALLY_CAP", "20000")
        with pytest.raises(lib.UtreeError) as ei:
            classify_fasta_bytes(db, tree, data)
        assert ei.value.code == lib.E_DEVICE
        monkeypatch.delenv("UTREE_TEST_TALLY_CAP")
        assert classify_fasta_bytes(db, tree, data) == want
    tree.close()


@pytest.mark.parametrize("hook,value", [("UTREE_TEST_LONG_CAP", "256"), ("UTREE_TEST_LONG_CAP", "100"), ("UTREE_TEST_PIECES_CAP", "300"), ("UTREE_TEST_PIECES_CAP", "7")])
def test_more_long_reads_or_pieces_than_the_tables_hold_is_reported(torch_cuda, hook, value, tmp_path, monkeypatch):
    """The tables of the pieces pass (long reads through the lane kernels) are sized from the batch's total_bases / max_len.  With the
    test hooks that shrink the capacities -- 256: a multiple of pieces_k's workgroup, where only a check that does not depend on the
    thread index fires -- the batch is reported (UTREE_E_DEVICE), every consumer of the lists stops at the capacity (the lists' areas
    keep their size, so a kernel that ran on would read unwritten entries: the results of the next batch would show it), and the next
    batch, hook off, equals the oracle."""
    from utree_amd import lib
    d = OwnDB(tmp_path, seed=5)
    db = CtrDB.open(d.ctr)
    tree = DeviceTree.upload(db, 0)
    o = orc.OracleDB.load(d.ctr)
    rng = np.random.default_rng(8)
    data = fasta_bytes(random_reads(rng, d, 400, 2200, 9000, hit_frac=0.8) + random_reads(rng, d, 3000, 100, 160, hit_frac=0.8))
    want = oracle_text(o, data, tmp_path, rc=True)
    monkeypatch.setenv("UTREE_LANE_PASS", "1")
    monkeypatch.setenv(hook, value)
    with pytest.raises(lib.UtreeError) as ei:
        classify_fasta_bytes(db, tree, data, rc=True)
    assert ei.value.code == lib.E_DEVICE
    monkeypatch.delenv(hook)
    assert classify_fasta_bytes(db, tree, data, rc=True) == want
    assert tree.kernel_name().startswith("classify_lanes_k<8, 2, 16, false, 2,")
    tree.close()


@pytest.mark.parametrize("name,rc", [("toy", 1), ("k64", 1), ("ix32", 0), ("vote", 0), ("katq2", 0)])
def test_line_sized_buckets_option(torch_cuda, name, rc, tmp_path, monkeypatch):
    """UTREE_BUCKET_BYTES=128 builds the image with one bucket per 128-byte HBM line (a third smaller; the kernels' NL = 2
    instantiations and the wave-per-read kernels' two-halves lookup): the reference's golden lines, and random reads of every
    length class against the oracle, through both kernel families."""
    monkeypatch.setenv("UTREE_BUCKET_BYTES", "128")
    d = util.load_db_fixture(name)
    db = CtrDB.open(util.fixture_ctr(name))
    tree = DeviceTree.upload(db, 0)
    assert tree.info.bucket_bytes == 128
    o = orc.OracleDB.load(util.fixture_ctr(name))
    data = util.fixture_bytes(util.READS_OF.get(name, name) + "_reads.fa.gz")
    want = util.fixture_bytes("%s_out%s.txt.gz" % (name, "_rc" if rc else ""))
    rng = np.random.default_rng(3)
    more = fasta_bytes(random_reads(rng, d, 2000, 1, 160, hit_frac=0.8) + random_reads(rng, d, 200, 161, 2095, hit_frac=0.8) + random_reads(rng, d, 8, 3000, 9000, hit_frac=0.8))
    for lane_pass in ("1", "0"):
        monkeypatch.setenv("UTREE_LANE_PASS", lane_pass)
        if b"\0" not in data:
            assert classify_fasta_bytes(db, tree, data, rc=bool(rc)) == want
        assert classify_fasta_bytes(db, tree, more, rc=bool(rc)) == oracle_text(o, more, tmp_path, rc=bool(rc))
        if lane_pass == "1":
            assert tree.kernel_name().startswith("classify_lanes_") and tree.kernel_name().endswith(", 2, false>")
    tree.close()
    monkeypatch.delenv("UTREE_BUCKET_BYTES")
    t64 = DeviceTree.upload(db, 0)
    assert t64.info.bucket_bytes == 64 and t64.info.image_bytes != 0
    t64.close()


def _rc16(x):
    r = 0
    for j in range(16):
        r = (r << 2) | (3 - ((x >> (2 * j)) & 3))
    return r


def _mix32(x):
    x = (x * 0x9E3779B1) & 0xFFFFFFFF
    x ^= x >> 15
    return (x * 0x85EBCA6B) & 0xFFFFFFFF


def _mer(x):
    return "".join("ACGT"[(x >> (30 - 2 * j)) & 3] for j in range(16))


@pytest.mark.parametrize("k", [32, 64])
def test_kmers_whose_two_minimizer_views_differ(torch_cuda, k, tmp_path, monkeypatch):
    """Image version 11 finds the reverse complement of a window from the window's own minimizer run: under the MIRRORED view of the
    database k-mer (device_common.hpp).  That view differs from the k-mer's own when two of its 16-mers tie in the 23 bits they are ranked
    by, when the minimizer occurs twice (tandem repeats), or when it is its own reverse complement -- such k-mers are stored twice.  A
    database made of exactly those, reads in both orientations, both strand modes, both kernel families and the one-pass / two-pass
    lane kernels: the oracle's lines every time."""
    rng = np.random.default_rng(100 + k)
    rs = lambda n: "".join("ACGT"[int(c)] for c in rng.integers(0, 4, n))
    comp = str.maketrans("ACGT", "TGCA")
    kmers = set()
    # (a) palindromic 16-mers with the smallest hashes: the minimizer wherever they occur
    pals = []
    for half in rng.integers(0, 1 << 16, 30000):
        x = int(half) << 16
        x |= _rc16(x) & 0xFFFF                                                         # second half = reverse complement of the first
        assert _rc16(x) == x
        pals.append(x)
    pals = sorted(set(pals), key=_mix32)[:6]
    for m in pals:
        for _ in range(40):
            p = int(rng.integers(0, k - 15))
            s = rs(k)
            kmers.add(s[:p] + _mer(m) + s[p + 16:])
    # (b) two different 16-mers that tie in the top 23 bits of their (tiny) canonical hashes, both in one k-mer
    cand = [int(c) for c in rng.integers(0, 1 << 32, 3_000_000, dtype=np.uint64)]
    tiny = {}
    for c in cand:
        h = _mix32(min(c, _rc16(c)))
        if h < (1 << 18):
            tiny.setdefault(h >> 9, []).append(c)
    pairs = [v[:2] for v in tiny.values() if len(v) >= 2 and min(v[0], _rc16(v[0])) != min(v[1], _rc16(v[1]))][:8]
    assert len(pairs) >= 4
    for a, b in pairs:
        for _ in range(30):
            s = rs(k)
            pa = int(rng.integers(0, k - 31))
            pb = int(rng.integers(pa + 16, k - 15))
            ma = _mer(a) if rng.integers(0, 2) else _mer(_rc16(a))
            mb = _mer(b) if rng.integers(0, 2) else _mer(_rc16(b))
            s = s[:pa] + ma + s[pa + 16:]
            kmers.add(s[:pb] + mb + s[pb + 16:])
    # (c) tandem repeats: the same 16-mer at several positions of a k-mer
    for period in (1, 2, 3, 4, 5, 7, 8, 11, 16):
        for _ in range(25):
            unit = rs(period)
            s = (unit * (k // period + 2))[:k]
            kmers.add(s)
            kmers.add(rs(3) + s[:k - 6] + rs(3))
    # (d) k-mers that hold a 16-mer AND its reverse complement (inverted repeats)
    for _ in range(60):
        m = rs(16)
        s = rs(k)
        pa = int(rng.integers(0, k - 31)); pb = int(rng.integers(pa + 16, k - 15))
        s = s[:pa] + m + s[pa + 16:]
        kmers.add(s[:pb] + m.translate(comp)[::-1] + s[pb + 16:])
    kmers |= {rs(k) for _ in range(3000)}                                             # ... and ordinary ones
    kmers = sorted(kmers)
    hi, lo = ctrfile.encode_kmers(kmers)
    labels = ["k__A;p__B;c__L%d" % i for i in range(12)]
    ix = rng.integers(0, 12, len(kmers)).astype(np.uint32)
    order = np.lexsort((lo, hi))
    ctr = str(tmp_path / ("views%d.ctr" % k))
    ctrfile.write_ctr(ctr, k // 4, 2, hi[order], lo[order], ix[order], labels)
    db = CtrDB.open(ctr)
    o = orc.OracleDB.load(ctr)
    reads = []
    for i, s in enumerate(kmers):
        t = rs(int(rng.integers(0, 40))) + s + rs(int(rng.integers(0, 40)))
        reads.append(("r%d" % i, t if i % 2 else t.translate(comp)[::-1]))
    reads += [("j%d" % i, "".join(kmers[int(a)] if rng.integers(0, 2) else kmers[int(a)].translate(comp)[::-1] for a in rng.integers(0, len(kmers), 2)))
              for i in range(600)]
    data = fasta_bytes(reads)
    want = {rc: oracle_text(o, data, tmp_path, rc=rc) for rc in (False, True)}
    assert want[True].count(b"\n") >= len(kmers)
    tree = DeviceTree.upload(db, 0)
    assert tree.info.strand_views == 1
    for lane_pass, bs in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("UTREE_LANE_PASS", lane_pass)
        monkeypatch.setenv("UTREE_LANES_BS", bs)
        for rc in (False, True):
            assert classify_fasta_bytes(db, tree, data, rc=rc) == want[rc], (lane_pass, bs, rc)
            if lane_pass == "1":
                assert tree.kernel_name().endswith(", 1, %s>" % ("true" if rc and bs == "1" else "false")), tree.kernel_name()
    tree.close()
    # without room for the second views the image says so, and both strands are walked as two sequences: same lines
    monkeypatch.setenv("UTREE_DUP_CAP", "3")
    monkeypatch.setenv("UTREE_LANE_PASS", "1")
    monkeypatch.setenv("UTREE_LANES_BS", "1")
    t2 = DeviceTree.upload(db, 0)
    assert t2.info.strand_views == 0
    assert classify_fasta_bytes(db, t2, data, rc=True) == want[True]
    assert t2.kernel_name().endswith(", 1, false>")
    t2.close()


@pytest.mark.parametrize("k", [32, 64])
def test_both_strands_when_nearly_every_bucket_overflows(torch_cuda, k, tmp_path, monkeypatch):
    """With both strands served from one pass a minimizer run can have TWO overflowing buckets (one per orientation), and the list of
    overflowing runs grows over the front of the run list, behind what phase B has read.  A table sized for 30 nodes per bucket
    (UTREE_BUCKET_TARGET) overflows nearly everywhere: the list must not outgrow its room -- a read whose overflowing run does not fit is
    left to the wave-per-read kernel -- and the lines stay the oracle's."""
    monkeypatch.setenv("UTREE_BUCKET_TARGET", "30")
    d = OwnDB(tmp_path, seed=31, k=k)
    db = CtrDB.open(d.ctr)
    tree = DeviceTree.upload(db, 0)
    o = orc.OracleDB.load(d.ctr)
    rng = np.random.default_rng(k + 1)
    comp = str.maketrans("ACGT", "TGCA")
    reads = random_reads(rng, d, 6000, 100, 160, hit_frac=0.9)
    reads = [(n, s if i % 2 else s.translate(comp)[::-1]) for i, (n, s) in enumerate(reads)]
    data = fasta_bytes(reads)
    monkeypatch.setenv("UTREE_LANE_PASS", "1")
    for rc in (True, False):
        assert classify_fasta_bytes(db, tree, data, rc=rc) == oracle_text(o, data, tmp_path, rc=rc)
        assert tree.kernel_name().startswith("classify_lanes_k<%d, 2, 1, false, 0, 1, %s>" % (k // 4, "true" if rc else "false"))
    tree.close()


@pytest.mark.parametrize("sub", [3, 7])
def test_k64_slots_split_by_the_four_bases_around_the_minimizer(torch_cuda, sub, tmp_path, monkeypatch):
    """Image version 13, k = 64: a window's minimizer keeps two bases' distance from the window's ends, and where one hash value holds more
    nodes than a bucket its slot is several pairs of buckets, picked by the four bases around the minimizer in its canonical orientation
    (utree_internal.h: UTREE_MIN_MARGIN; dev_image.c: compute_regions).  Databases of the test suite's size never get there by themselves
    (the full-size configs[4] test does): UTREE_TEST_SUB gives every slot `sub` pairs.  Reads in both orientations, both strand modes,
    both kernel families: the oracle's lines."""
    monkeypatch.setenv("UTREE_TEST_SUB", str(sub))
    d = OwnDB(tmp_path, seed=77 + sub, k=64)
    db = CtrDB.open(d.ctr)
    o = orc.OracleDB.load(d.ctr)
    rng = np.random.default_rng(sub)
    comp = str.maketrans("ACGT", "TGCA")
    reads = random_reads(rng, d, 5000, 64, 160, hit_frac=0.9)
    reads = [(n, s if i % 2 else s.translate(comp)[::-1]) for i, (n, s) in enumerate(reads)]
    reads += random_reads(rng, d, 300, 200, 2500, hit_frac=0.8)
    data = fasta_bytes(reads)
    want = {rc: oracle_text(o, data, tmp_path, rc=rc) for rc in (False, True)}
    tree = DeviceTree.upload(db, 0)
    for lane_pass, bs in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("UTREE_LANE_PASS", lane_pass)
        monkeypatch.setenv("UTREE_LANES_BS", bs)
        for rc in (False, True):
            assert classify_fasta_bytes(db, tree, data, rc=rc) == want[rc], (lane_pass, bs, rc)
    monkeypatch.delenv("UTREE_TEST_SUB")
    plain = DeviceTree.upload(db, 0)
    assert tree.info.image_bytes > plain.info.image_bytes + (sub - 1) * (1 << 31) - (1 << 28)      # 2^24 more pairs per pair and slot
    plain.close()
    tree.close()
