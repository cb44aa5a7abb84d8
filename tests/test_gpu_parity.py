"""GPU parity tests: the HIP path, called through the C-ABI, against (a) the committed outputs of the genuine
reference (tests/golden) and (b) the CPU oracle on seeded inputs.  Bit-exact: integer / byte work.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import orc
from utree_amd import ctrfile, lib
from utree_amd.search import CtrDB, DeviceTree, classify_fasta_bytes, frame_fasta, search_gg
import util


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


_TREES = {}


def tree_for(name, fine_bits=lib.FINE_AUTO):
    key = (name, fine_bits)
    if key not in _TREES:
        while len(_TREES) >= 6:                         # (handles keep their buffers until closed: the oldest go)
            _TREES.pop(next(iter(_TREES)))[1].close()
        db = CtrDB.open(util.fixture_ctr(name))
        _TREES[key] = (db, DeviceTree.upload(db, 0, fine_bits))
    return _TREES[key]


GOLDEN = [("toy", 0), ("toy", 1), ("k64", 0), ("k64", 1), ("ix32", 0), ("ix32", 1), ("k64ix32", 0), ("k64ix32", 1), ("vote", 0), ("kat", 0), ("katq", 0),
          ("katq2", 0), ("generic", 0), ("k16", 0), ("k16", 1)]


@pytest.mark.parametrize("name,rc", GOLDEN)
def test_golden_outputs_bit_exact(torch_cuda, name, rc):
    db, tree = tree_for(name)
    got = classify_fasta_bytes(db, tree, util.fixture_bytes(util.READS_OF.get(name, name) + "_reads.fa.gz"), rc=bool(rc))
    want = util.fixture_bytes("%s_out%s.txt.gz" % (name, "_rc" if rc else ""))
    assert got == want


@pytest.mark.parametrize("fine_bits", [0, 1, 3, 6])
@pytest.mark.parametrize("name", ["toy", "k64", "ix32", "katq", "katq2"])
def test_fine_index_width_does_not_change_results(torch_cuda, name, fine_bits):
    db, tree = tree_for(name, fine_bits)
    assert tree.info.fine_bits == fine_bits
    got = classify_fasta_bytes(db, tree, util.fixture_bytes(name + "_reads.fa.gz"), rc=False)
    assert got == util.fixture_bytes(name + "_out.txt.gz")


def test_irregular_bins_and_generic_mode_are_detected(torch_cuda):
    """katq2: the first-bin quirk put a LARGER record in front of a bin -> that bin is flagged and searched with the
    reference's probe order; generic: non-monotone bin table -> every bin takes that path.  Both golden-checked above."""
    _, tk = tree_for("kat")
    _, tq2 = tree_for("katq2")
    _, tg = tree_for("generic")
    assert tk.info.irregular_bins == 0 and tk.info.generic_mode == 0
    assert tq2.info.irregular_bins >= 1 and tq2.info.generic_mode == 0
    assert tg.info.generic_mode == 1 and tg.info.fine_bits == 0


def test_lookup_operator_matches_oracle(torch_cuda):
    torch = torch_cuda
    for name in ("kat", "katq", "toy", "k64", "ix32", "k16"):
        db, tree = tree_for(name, 4)
        o = orc.OracleDB.load(util.fixture_ctr(name))
        d = util.load_db_fixture(name)
        hi, lo = d.words() if name != "katq" else (np.zeros(0, np.uint64), np.zeros(0, np.uint64))
        rng = np.random.default_rng(1)
        if name == "katq":
            # words cannot be reconstructed through the quirk table; probe suffix x prefix combinations
            sh, sl = d.suffixes()
            pref = rng.integers(0, 1 << 24, size=len(sl)).astype(np.uint64)
            nz = np.flatnonzero(np.diff(d.binix.astype(np.int64)) > 0)
            pref[: len(nz)] = nz[: len(pref)].astype(np.uint64)[: len(nz)]
            lo = sl | (pref << np.uint64(40))
            hi = np.zeros_like(lo)
        # hits, near misses, random words
        far = np.uint64(1 << 8) if d.W == 4 else np.uint64(1 << 40)                       # the same suffix in the next bin
        qlo = np.concatenate([lo, lo ^ np.uint64(1), lo + far, rng.integers(0, 1 << 63, 5000).astype(np.uint64)])
        if d.W == 4:
            qlo &= np.uint64(0xFFFFFFFF)                                                 # PACKSIZE=16: a word is 32 bits
        qhi = np.concatenate([hi, hi, hi, rng.integers(0, 1 << 63, 5000).astype(np.uint64) if d.W == 16 else np.zeros(5000, np.uint64)])
        t_lo = torch.from_numpy(qlo.view(np.int64)).cuda()
        t_hi = torch.from_numpy(qhi.view(np.int64)).cuda()
        got = tree.get_ix(t_hi if d.W == 16 else None, t_lo).cpu().numpy().view(np.uint32)
        nl = o.n_labels
        for j in range(len(qlo)):
            w = o.lookup(int(qhi[j]), int(qlo[j]))
            w = w if w < nl else 0xFFFFFFFF
            assert int(got[j]) == w, (name, j, hex(int(qhi[j])), hex(int(qlo[j])))


def random_reads(rng, d, n, min_len, max_len, hit_frac=0.6):
    """Reads that concatenate DB k-mers (hits) with random stretches, N's and lowercase."""
    hi, lo = d.words()
    k = d.k
    out = []
    for i in range(n):
        L = int(rng.integers(min_len, max_len + 1))
        parts = []
        cur = 0
        while cur < L:
            if rng.random() < hit_frac and len(lo):
                j = int(rng.integers(0, len(lo)))
                s = ctrfile.decode_kmer(int(hi[j]), int(lo[j]), k)
                if rng.random() < 0.3:        # overlap-extend with a neighbour k-mer sharing a label region
                    s += "".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 9))))
            else:
                s = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 50))))
            r = rng.random()
            if r < 0.05:
                s = s.lower()
            elif r < 0.10:
                p = int(rng.integers(0, len(s)))
                s = s[:p] + "N" + s[p + 1:]
            parts.append(s)
            cur += len(s)
        out.append(("r%d" % i, "".join(parts)[:L]))
    return out


def fasta_bytes(reads):
    return b"".join(b">" + n.encode() + b"\n" + s.encode() + b"\n" for n, s in reads)


@pytest.mark.parametrize("name", ["toy", "k64", "ix32", "vote", "k16"])
@pytest.mark.parametrize("rc", [0, 1])
def test_random_reads_vs_oracle_short_and_long(torch_cuda, name, rc, tmp_path):
    """Seeded reads from 1 bp to 40 kb (short wave-per-read path, long workgroup-per-read path, both strands)."""
    d = util.load_db_fixture(name)
    db, tree = tree_for(name, 3)
    o = orc.OracleDB.load(util.fixture_ctr(name))
    rng = np.random.default_rng(1234 + rc)
    reads = (random_reads(rng, d, 1500, 1, 330) + random_reads(rng, d, 60, 300, 5000) +
             random_reads(rng, d, 6, 20000, 40000, hit_frac=0.9))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    data = fasta_bytes(reads)
    got = classify_fasta_bytes(db, tree, data, rc=bool(rc))
    fa = tmp_path / "r.fa"
    fa.write_bytes(data)
    out = tmp_path / "o.txt"
    code, nr, good, err = o.search_file(str(fa), str(out), threads=8, rc=bool(rc))
    assert code == 0 and nr == len(reads)
    assert got == out.read_bytes()


def test_result_records_match_oracle_records(torch_cuda):
    """Field-by-field (label, cut, found, uix, sl, ol), not only the formatted text."""
    torch = torch_cuda
    db, tree = tree_for("toy")
    o = orc.OracleDB.load(util.fixture_ctr("toy"))
    data = util.fixture_bytes("toy_reads.fa.gz")
    fr = frame_fasta(data)
    buf = np.frombuffer(data, dtype=np.uint8)
    for rc in (False, True):
        want = o.classify_batch(buf, fr["seq_off"], fr["seq_len"], rc=rc, threads=8)
        res = tree.classify(torch.from_numpy(buf.copy()).cuda(), torch.from_numpy(fr["seq_off"].astype(np.int64)).cuda(),
                            torch.from_numpy(fr["seq_len"].astype(np.int32)).cuda(), rc=rc).cpu().numpy()
        got = res.view(np.uint32)
        hit = want["found"] > 0
        assert np.array_equal(got[:, 2], want["found"])
        assert np.array_equal(got[hit, 3], want["uix"][hit])
        assert np.array_equal(got[hit, 0], want["label"][hit])
        assert np.array_equal(res[hit, 1], want["cut"][hit])
        multi = hit & (want["uix"] > 1)
        assert np.array_equal(got[multi, 4], want["sl"][multi]) and np.array_equal(got[multi, 5], want["ol"][multi])


@pytest.mark.parametrize("shift", [1, 2, 3])
def test_unaligned_caller_buffer(torch_cuda, shift):
    """The kernels fetch aligned dwords around the read's bytes: a bases pointer at any byte offset of an allocation, with
    the first read at the allocation's first byte and the last read ending at its last byte, gives the same records."""
    torch = torch_cuda
    db, tree = tree_for("toy")
    data = util.fixture_bytes("toy_reads.fa.gz")
    fr = frame_fasta(data)
    buf = np.frombuffer(data, dtype=np.uint8)
    lo, hi = int(fr["seq_off"][0]), int(fr["seq_off"][-1] + fr["seq_len"][-1])
    off = torch.from_numpy((fr["seq_off"] - lo).astype(np.int64)).cuda()
    ln = torch.from_numpy(fr["seq_len"].astype(np.int32)).cuda()
    for rc in (False, True):
        base = torch.from_numpy(buf[lo:hi].copy()).cuda()
        want = tree.classify(base, off, ln, rc=rc).cpu().numpy()
        big = torch.zeros(shift + hi - lo, dtype=torch.uint8, device="cuda")
        big[shift:] = base
        got = tree.classify(big[shift:], off, ln, rc=rc).cpu().numpy()
        assert np.array_equal(got, want)


def test_edge_case_files_through_search_file(torch_cuda, tmp_path):
    """Whole-file path (framing, sharding, formatting, exit conditions) on the parser edge cases."""
    cases = json.load(open(os.path.join(util.GOLD, "edge_cases.json")))
    db, tree = tree_for("toy")
    for nm, c in sorted(cases.items()):
        fa = tmp_path / ("%s.fa" % nm)
        fa.write_bytes(bytes.fromhex(c["input_hex"]))
        out = tmp_path / ("%s.out" % nm)
        code, st = search_gg(db, [tree], str(fa), str(out), rc=bool(c["rc"]), threads=2)
        exit_code = {lib.OK: 0, lib.E_FASTA: 2, lib.E_IO: 1}[code]
        assert exit_code == c["exit"], nm
        assert out.read_bytes() == bytes.fromhex(c["output_hex"]), nm


@pytest.mark.parametrize("name,rc", [("toy", 0), ("toy", 1), ("k64", 1), ("ix32", 0), ("k16", 1)])
def test_cli_drop_in(torch_cuda, name, rc, tmp_path):
    """The xtree-searchGG command line: same arguments, same output file, same banners, exit code 0."""
    out = tmp_path / "cls.txt"
    cmd = [lib.CLI_PATH, util.fixture_ctr(name), util.fixture_reads_path(name), str(out), "4"] + (["RC"] if rc else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    want = util.fixture_bytes("%s_out%s.txt.gz" % (name, "_rc" if rc else ""))
    assert out.read_bytes() == want
    so = r.stdout.decode()
    assert "This is UTree [v2.0RF SigNature Edition]" in so
    assert ("Reverse complement consideration is %sabled." % ("en" if rc else "dis")) in so
    assert "Tree read." in so and ("Good finds: %d" % want.count(b"\n")) in so
    m = util.manifest()
    assert ("Nodes in input tree: %d" % m[name + "_nodes"]) in so
    if name == "k16":
        assert "(PACKSIZE=16, CNTTYPE=NA, IXTYPE=uint16_t, SZ=3)" in so


def test_cli_usage_and_bad_db(tmp_path):
    # a tree no build of the reference reads (PACKSIZE=8 does not compile there): its own refusal text and exit code (itree.c:746-751)
    p = tmp_path / "p8.ctr"
    p.write_bytes(np.array([2, 0, 2, 5], dtype="<u8").tobytes() + b"\0" * 64)
    r = subprocess.run([lib.CLI_PATH, str(p), "a", "b"], stdout=subprocess.PIPE)
    assert r.returncode == 0 and b"ERROR. Input tree requires PACKSIZE=8, CNTTYPE=NA, IXTYPE=uint16_t" in r.stdout
    r = subprocess.run([lib.CLI_PATH], stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"usage: xtree-searchGG compTree.ctr fastaToSearch.fa output.txt [threads] [SPEED <X>] [RC]" in r.stdout
    r = subprocess.run([lib.CLI_PATH, str(tmp_path / "none.ctr"), "a", "b"], stdout=subprocess.PIPE)
    assert r.returncode == 0 and b"Invalid DB file" in r.stdout          # itree.c:735 exits 0


def test_size_independent_properties_at_scale(torch_cuda):
    """Properties that need no oracle: (1) a read and its reverse complement classify identically with RC on;
    (2) permuting the reads permutes the results; (3) classify is idempotent; (4) appending a no-hit read
    changes nothing else.  Run on 200k synthetic reads against a 4M-node synthetic DB."""
    torch = torch_cuda
    from utree_amd import synth
    sdb = synth.make_db(torch.device("cuda:0"), n_nodes=4_000_000, seed=synth.DB_SEED)
    tree = sdb.tree
    reads = synth.make_reads(sdb, n_reads=200_000, read_len=150, seed=7)
    base = tree.classify(reads.bases, reads.off, reads.length, rc=True).clone()
    again = tree.classify(reads.bases, reads.off, reads.length, rc=True).clone()
    assert torch.equal(base, again)
    assert int((base[:, 2] > 0).sum()) > 150_000
    # reverse complement of every read
    seq = reads.bases.view(-1, 150)
    comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=seq.device)
    for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
        comp[a] = b
    rcseq = comp[seq.long()].flip(1).contiguous()
    r2 = tree.classify(rcseq.view(-1), reads.off, reads.length, rc=True)
    assert torch.equal(base[:, 2:], r2[:, 2:])          # found, uix, sl, ol
    assert torch.equal(base[:, :2], r2[:, :2])          # label, cut
    perm = torch.randperm(reads.off.numel(), device=seq.device)
    r3 = tree.classify(reads.bases, reads.off[perm], reads.length[perm], rc=True)
    assert torch.equal(base[perm], r3)


def test_attached_image_copy_classifies_identically(torch_cuda):
    """The N>1 path: a byte copy of the flat device image (what the RCCL broadcast delivers to the other ranks)
    is adopted with utree_dev_attach and must give identical results; the image really is position independent."""
    torch = torch_cuda
    from utree_amd import synth
    sdb = synth.make_db(torch.device("cuda:0"), n_nodes=3_000_000, seed=synth.DB_SEED)
    reads = synth.make_reads(sdb, n_reads=100_000, read_len=150, seed=11)
    base = sdb.tree.classify(reads.bases, reads.off, reads.length, rc=True).clone()
    ptr, used = sdb.tree.image_ptr()
    src = sdb.tree.image_tensor()
    assert src.data_ptr() == ptr and used <= src.numel()
    pad = torch.empty(4096 + used, dtype=torch.uint8, device="cuda:0")      # a different address, differently aligned (mod 8 KiB)
    copy = pad[4096:4096 + used]
    copy.copy_(src[:used])
    ctr2 = CtrDB.from_memory(sdb.W, 2, sdb.n_nodes, np.zeros((1 << 24) + 1, dtype=np.uint64), None, sdb.label_text)
    t2 = DeviceTree.attach(ctr2, copy, 0)
    assert t2.info.fine_bits == sdb.tree.info.fine_bits and t2.info.image_bytes == used
    again = t2.classify(reads.bases, reads.off, reads.length, rc=True)
    assert torch.equal(base, again)
    assert int((base[:, 2] > 0).sum()) > 80_000


@pytest.mark.parametrize("name", ["toy", "katq2", "generic", "k64"])
def test_64bit_offset_instantiations(torch_cuda, name, monkeypatch):
    """Databases with N >= 2^32-1 nodes use 8-byte bin-table entries and 64-bit offsets on the device; no fixture can
    be that large, so a test hook builds the image of a small database with the 64-bit instantiations instead."""
    monkeypatch.setenv("UTREE_FORCE_OFF64", "1")
    db = CtrDB.open(util.fixture_ctr(name))
    tree = DeviceTree.upload(db, 0, 2)
    got = classify_fasta_bytes(db, tree, util.fixture_bytes(util.READS_OF.get(name, name) + "_reads.fa.gz"), rc=False)
    assert got == util.fixture_bytes(name + "_out.txt.gz")
    tree.close()


@pytest.mark.parametrize("name,parts", [("toy", 4), ("k64", 8), ("ix32", 64), ("katq2", 2), ("vote", 16)])
def test_image_built_in_parts(torch_cuda, name, parts, monkeypatch):
    """Trees of billions of nodes get their (hash, position, rest) order in 2..64 parts by the top hash bits so that the sort
    buffers fit beside the image (tests/scale/big_tree_check.py runs a 4.4 G-node tree); a hook forces that path on small ones."""
    monkeypatch.setenv("UTREE_BUILD_PARTS", str(parts))
    db = CtrDB.open(util.fixture_ctr(name))
    tree = DeviceTree.upload(db, 0, 3)
    got = classify_fasta_bytes(db, tree, util.fixture_bytes(util.READS_OF.get(name, name) + "_reads.fa.gz"), rc=False)
    assert got == util.fixture_bytes(name + "_out.txt.gz")
    tree.close()


@pytest.mark.parametrize("kind", ["fastq", "fastq_gz", "multiline", "multiline_gz", "auto_fastq", "auto_fasta_gz"])
def test_opt_in_input_formats_give_the_reference_output(torch_cuda, kind, tmp_path):
    """the toy reads rewritten as FASTQ / multi-line FASTA, plain or gzip, classify to exactly the
    golden output of the reference on the two-line FASTA (opt-in: the default framing stays the reference's)."""
    import gzip as gz
    data = util.fixture_bytes("toy_reads.fa.gz")
    names, off, ln = util.parse_fasta(data)
    reads = [(names[i], data[off[i]:off[i] + ln[i]]) for i in range(len(names))]
    if "fastq" in kind:
        blob = b"".join(b"@" + n + b" c\n" + s + b"\n+" + n + b"\n" + b"#" * len(s) + b"\n" for n, s in reads)
    else:
        blob = b"".join(b">" + n + b" c\n" + b"".join(s[a:a + 60] + b"\n" for a in range(0, len(s), 60)) for n, s in reads)
    path = tmp_path / ("reads." + kind)
    path.write_bytes(gz.compress(blob, 1) if kind.endswith("gz") else blob)
    fmt = lib.INPUT_AUTO if kind.startswith("auto") else (lib.INPUT_FASTQ if "fastq" in kind else lib.INPUT_FASTA_MULTILINE)
    db, tree = tree_for("toy")
    for rc in (0, 1):
        out = tmp_path / "o.txt"
        code, st = search_gg(db, [tree], str(path), str(out), rc=bool(rc), threads=4, input_format=fmt)
        assert code == lib.OK and st.n_reads == len(reads)
        assert out.read_bytes() == util.fixture_bytes("toy_out%s.txt.gz" % ("_rc" if rc else ""))
    # the command line: UTREE_INPUT
    out = tmp_path / "cli.txt"
    env = dict(os.environ, UTREE_INPUT="auto")
    r = subprocess.run([lib.CLI_PATH, util.fixture_ctr("toy"), str(path), str(out), "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, env=env)
    assert r.returncode == 0 and out.read_bytes() == util.fixture_bytes("toy_out.txt.gz")


@pytest.mark.parametrize("rc", [0, 1])
def test_more_long_reads_than_resident_workgroups(torch_cuda, rc, tmp_path):
    """classify_long_k hands long reads out one at a time; with more long reads than workgroups that fit the chip a
    workgroup takes a second, third ... read.  (A shared-memory slot reused between the work index and the vote made
    that path lose reads; bench.py's 10 kb configuration found it.)"""
    d = util.load_db_fixture("toy")
    db, tree = tree_for("toy")
    o = orc.OracleDB.load(util.fixture_ctr("toy"))
    rng = np.random.default_rng(77 + rc)
    reads = random_reads(rng, d, 2600, 2150, 2600, hit_frac=0.8) + random_reads(rng, d, 40, 5000, 9000, hit_frac=0.9)
    data = fasta_bytes(reads)
    got = classify_fasta_bytes(db, tree, data, rc=bool(rc))
    fa = tmp_path / "r.fa"
    fa.write_bytes(data)
    out = tmp_path / "o.txt"
    code, nr, good, err = o.search_file(str(fa), str(out), threads=8, rc=bool(rc))
    assert code == 0 and nr == len(reads)
    assert got == out.read_bytes()


@pytest.mark.parametrize("run_max", [0, 20])
def test_overflowing_buckets(torch_cuda, tmp_path, run_max, monkeypatch):
    """Hundreds of k-mers that share ONE minimizer land in one 64-byte bucket: 7 stay inline, the rest are reached through the
    bucket's overflow descriptor (binary search in the sorted records).  Such buckets are ~1 % at the design load and never
    occur in the small fixtures, so this database is made of them (k = 32 and k = 64).
    run_max = 20: the same database with the descriptor's range cut to 20 records (UTREE_BUCKET_RUN_MAX; the real limit is
    2^22, e.g. millions of k-mers containing A^16, whose hash is 0) -- the runs then saturate, the bins of their nodes are
    flagged and searched with the reference's probe sequence, and the database still loads and answers like the oracle."""
    if run_max:
        monkeypatch.setenv("UTREE_BUCKET_RUN_MAX", str(run_max))
    def rc16(x):                                                      # reverse complement of 16-mers packed first base first (A=0 .. T=3)
        x = x.astype(np.uint32)
        r = np.zeros_like(x)
        for j in range(16):
            r = (r << np.uint32(2)) | (np.uint32(3) - ((x >> np.uint32(2 * j)) & np.uint32(3)))
        return r
    rng = np.random.default_rng(5)
    cand = rng.integers(0, 1 << 32, 6_000_000, dtype=np.uint64).astype(np.uint32)
    canon = np.minimum(cand, rc16(cand))                              # image version 11: a 16-mer is ranked by the hash of its canonical form
    with np.errstate(over="ignore"):
        h = canon.copy(); h *= np.uint32(0x9E3779B1); h ^= h >> np.uint32(15); h *= np.uint32(0x85EBCA6B)
    best = np.argsort(h)[:3]
    assert all(int(h[b]) < 5000 for b in best)
    # three 16-mers with tiny hashes: minimizers wherever they occur; one as its canonical form, one reverse-complemented, one as drawn
    minis = [int(canon[best[0]]), int(rc16(canon[best[1]:best[1] + 1])[0]), int(cand[best[2]])]
    for W in (8, 16):
        k = 4 * W
        kmers = set()
        for m in minis:
            core = "".join("ACGT"[(m >> (30 - 2 * j)) & 3] for j in range(16))
            for _ in range(150):
                p = int(rng.integers(0, k - 15))
                s = "".join("ACGT"[c] for c in rng.integers(0, 4, k))
                kmers.add(s[:p] + core + s[p + 16:])
        kmers = sorted(kmers)
        hi, lo = ctrfile.encode_kmers(kmers)
        ix = rng.integers(0, 12, len(kmers)).astype(np.uint32)
        labels = ["k__A;p__B;c__L%d" % i for i in range(12)]
        order = np.lexsort((lo, hi))
        ctr = str(tmp_path / ("ovf%d.ctr" % k))
        ctrfile.write_ctr(ctr, W, 2, hi[order], lo[order], ix[order], labels)
        db = CtrDB.open(ctr)
        tree = DeviceTree.upload(db, 0)
        assert (tree.info.irregular_bins > 0) == bool(run_max)
        o = orc.OracleDB.load(ctr)
        reads = [("h%d" % i, s) for i, s in enumerate(kmers)]
        reads += [("m%d" % i, s[:-1] + "ACGT"[("ACGT".index(s[-1]) + 1) & 3]) for i, s in enumerate(kmers[::3])]   # near misses
        reads += [("j%d" % i, "".join(kmers[int(a)] for a in rng.integers(0, len(kmers), 4))) for i in range(200)]   # several per read
        data = "".join(">%s\n%s\n" % r for r in reads).encode()
        fa = tmp_path / "r.fa"; fa.write_bytes(data)
        out = tmp_path / "o.txt"
        code, nr, good, err = o.search_file(str(fa), str(out), threads=4, rc=False)
        assert code == 0 and good >= len(kmers)
        assert classify_fasta_bytes(db, tree, data, rc=False) == out.read_bytes()
        # both strands (one pass over canonical minimizer runs: the overflow runs of BOTH buckets of a pair), reads given as reverse complements too
        comp = str.maketrans("ACGT", "TGCA")
        data2 = "".join(">%s\n%s\n" % (n, sq if i % 2 else sq.translate(comp)[::-1]) for i, (n, sq) in enumerate(reads)).encode()
        fa.write_bytes(data2)
        code, nr, good2, err = o.search_file(str(fa), str(out), threads=4, rc=True)
        assert code == 0 and good2 >= len(kmers)
        assert classify_fasta_bytes(db, tree, data2, rc=True) == out.read_bytes()
        tree.close()


@pytest.mark.parametrize("seed,max_depth,table", [(1, 8, None), (2, 8, None), (3, 8, None), (4, 7, True), (5, 7, True), (6, 5, True), (4, 7, False), (6, 5, False)])
def test_vote_on_adversarial_label_sets(torch_cuda, seed, max_depth, table, tmp_path, monkeypatch):
    """vote_k skips the levels a whole group of labels shares with ONE scan of the group's first and last label; this
    checks it (and the level loop behind it) on label sets built to sit on the vote's edges (util.adversarial_vote_case).
    Oracle = the CPU restatement, itself held against the genuine reference on the same cases
    (test_oracle_golden.py::test_vote_adversarial_cases_oracle_vs_reference).
    Label sets of at most 8 tokens per label get the image's label table and vote_k's decisions from token ids (table=True); the same
    sets with the table switched off (UTREE_VOTE_BYTES) and the deeper sets go through the byte scans."""
    if table is False:
        monkeypatch.setenv("UTREE_VOTE_BYTES", "1")
    ctr_path, data, n_reads = util.adversarial_vote_case(seed, str(tmp_path), max_depth)
    fa = tmp_path / "r.fa"
    fa.write_bytes(data)
    o = orc.OracleDB.load(ctr_path)
    out = tmp_path / "o.txt"
    code, nr, good, err = o.search_file(str(fa), str(out), threads=8, rc=False)
    assert code == 0 and nr == n_reads and good > 0
    db = CtrDB.open(ctr_path)
    tree = DeviceTree.upload(db, 0)
    if table is not None:
        assert tree.info.vote_table == int(table)
    assert classify_fasta_bytes(db, tree, data, rc=False) == out.read_bytes()
    tree.close()


@pytest.mark.parametrize("name,with_gt", [("toy", False), ("toy", True), ("k64", False), ("ix32", False), ("vote", False)])
def test_byte_fuzz_vs_oracle(torch_cuda, name, with_gt, tmp_path):
    """Sequence lines with arbitrary 7-bit bytes (util.byte_fuzz_reads: the staging's byte-parallel base coding must call
    exactly the reference's eight letters good), both strands, whole-file path: output file and exit code equal the
    oracle's, which equals the genuine reference's on the same data (test_oracle_golden.py::test_byte_fuzz_oracle_vs_reference)."""
    data = util.byte_fuzz_reads(name, 21, with_gt)
    fa = tmp_path / "f.fa"
    fa.write_bytes(data)
    db, tree = tree_for(name)
    o = orc.OracleDB.load(util.fixture_ctr(name))
    for rc in (False, True):
        want, got = tmp_path / "orc.txt", tmp_path / "gpu.txt"
        code, nr, good, err = o.search_file(str(fa), str(want), threads=8, rc=rc)
        gcode, st = search_gg(db, [tree], str(fa), str(got), rc=rc, threads=2)
        assert {lib.OK: 0, lib.E_FASTA: 2, lib.E_IO: 1}[gcode] == code
        assert got.read_bytes() == want.read_bytes()


def test_framing_fuzz_vs_oracle(torch_cuda, tmp_path):
    """The same forty random framing files through the product's whole-file path: exit code and output equal the oracle's
    (which equals the genuine reference's: test_oracle_golden.py::test_framing_fuzz_oracle_vs_reference)."""
    db, tree = tree_for("toy")
    o = orc.OracleDB.load(util.fixture_ctr("toy"))
    for seed in range(40):
        fa = tmp_path / "f.fa"
        fa.write_bytes(util.framing_fuzz_case(seed))
        for rc in (False, True):
            want, got = tmp_path / "orc.txt", tmp_path / "gpu.txt"
            for f in (want, got):
                if f.exists():
                    f.unlink()
            code, nr, good, err = o.search_file(str(fa), str(want), threads=2, rc=rc)
            gcode, st = search_gg(db, [tree], str(fa), str(got), rc=rc, threads=2)
            assert {lib.OK: 0, lib.E_FASTA: 2, lib.E_IO: 1}[gcode] == code, seed
            assert (got.read_bytes() if got.exists() else b"") == (want.read_bytes() if want.exists() else b""), seed


def test_tally_workspace_under_label_diversity_and_mixed_lengths(torch_cuda):
    """The (rank, count) lists of a batch come out of one cursor shared by the 150-bp-class pass and the mid-length pass; the
    workspace bound (dev_image.c::carve) has to hold when nearly every window hits a different label and both passes run: reads of
    150 and 400 bp whose planted k-mers come from random labels, many reads, against the oracle."""
    torch = torch_cuda
    from utree_amd import synth
    dev = torch.device("cuda:0")
    sdb = synth.make_db(dev, 40_000_000, keep_raw=True)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    parts, offs, lens, at = [], [], [], 0
    for L, n in ((150, 300_000), (400, 120_000)):
        slots = L // 32
        node = torch.randint(0, sdb.n_nodes, (n, slots), generator=g, device=dev)                  # a random node = a random label per k-mer
        hi = synth.mix64(node ^ synth._s64(sdb.seed))
        shifts = torch.arange(62, -2, -2, device=dev, dtype=torch.int64)
        codes = (hi.unsqueeze(-1) >> shifts) & 3
        acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        seq = acgt[torch.randint(0, 4, (n, L), generator=g, device=dev)]
        seq[:, : slots * 32] = acgt[codes].view(n, slots * 32)
        parts.append(seq.contiguous().view(-1))
        offs.append(torch.arange(n, dtype=torch.int64, device=dev) * L + at)
        lens.append(torch.full((n,), L, dtype=torch.int32, device=dev))
        at += n * L
    bases, off, ln = torch.cat(parts), torch.cat(offs), torch.cat(lens)
    perm = torch.randperm(off.numel(), device=dev, generator=g)
    off, ln = off[perm].contiguous(), ln[perm].contiguous()
    got = sdb.tree.classify(bases, off, ln, rc=True).cpu().numpy()
    o = orc.OracleDB.from_memory(sdb.W, 2, sdb.binix.cpu().numpy().view(np.uint32).astype(np.uint64), sdb.records.cpu().numpy(), sdb.label_text)
    want = o.classify_batch(bases.cpu().numpy(), off.cpu().numpy().astype(np.uint64), ln.cpu().numpy().astype(np.uint32), rc=True, threads=16)
    g32 = got.view(np.uint32)
    assert np.array_equal(g32[:, 2], want["found"]) and np.array_equal(g32[:, 3], want["uix"])
    hit = want["found"] > 0
    assert np.array_equal(g32[hit, 0], want["label"][hit]) and np.array_equal(got[hit, 1], want["cut"][hit])
    multi = want["uix"] > 1
    assert np.array_equal(g32[multi, 4], want["sl"][multi]) and np.array_equal(g32[multi, 5], want["ol"][multi])
    assert float(want["uix"].mean()) > 3.5                                                     # nearly every hit a label of its own
    sdb.tree.close()


@pytest.mark.parametrize("case", ["quirk", "dups", "generic", "ix32"])
def test_packsize16_direct_table_on_irregular_trees(torch_cuda, case, tmp_path):
    """PACKSIZE=16 (README.md:88): a k-mer is a 32-bit word and the image holds XT_getIX32's answer for every word.  Where a bin is not
    strictly ascending -- COMPRESS' first-bin quirk, repeated or unsorted suffixes -- or the bin table is not monotone, the table is
    filled by the reference's own probe order: the oracle's lines (itself held to the genuine -D PACKSIZE=16 build on these very
    files, tests/test_oracle_golden.py), both strands, and the word operator on every probe."""
    torch = torch_cuda
    ctr, data = util.k16_table_cases(str(tmp_path))[case]
    db = CtrDB.open(ctr)
    tree = DeviceTree.upload(db, 0)
    o = orc.OracleDB.load(ctr)
    assert tree.info.bucket_bytes == 0 and tree.info.lane_pass == 0
    if case in ("quirk", "dups"):
        assert tree.info.irregular_bins >= 1
    if case == "generic":
        assert tree.info.generic_mode == 1
    fa = tmp_path / "r.fa"
    fa.write_bytes(data)
    for rc in (False, True):
        out = tmp_path / "o.txt"
        code, nr, good, err = o.search_file(str(fa), str(out), threads=4, rc=rc)
        assert code == 0 and good > 100
        assert classify_fasta_bytes(db, tree, data, rc=rc) == out.read_bytes()
    rng = np.random.default_rng(2)
    q = np.concatenate([rng.integers(0, 1 << 32, 20000, dtype=np.uint64), np.arange(0x10 << 8, (0x12 << 8), dtype=np.uint64)])
    got = tree.get_ix(None, torch.from_numpy(q.view(np.int64)).cuda()).cpu().numpy().view(np.uint32)
    nl = o.n_labels
    for j in range(len(q)):
        w = o.lookup(0, int(q[j]))
        assert int(got[j]) == (w if w < nl else 0xFFFFFFFF), (case, hex(int(q[j])))
    tree.close()
