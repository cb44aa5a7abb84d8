"""BASELINE.json's configurations at FULL database size, through the C-ABI on the MI355X, against the CPU oracle on a sample
and through size-independent properties (the fixtures in tests/golden pin the same code at toy size):

  configs[1]  8 GB L2 CTR (1 217 000 000 32-mer nodes), 150 bp reads
  configs[4]  PACKSIZE=64 build (568 000 000 64-mer nodes, the wide-key lookup path), 150 bp reads
  configs[2]  500 MB L4 CTR (72 000 000 nodes), long reads (1-100 kb, mean ~10 kb), RC on: the mid-length pass and classify_long_k

Databases and reads are seeded synthetic (utree_amd.synth); the oracle gets the same on-disk pieces (bin
table + packed records + labels).  Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import orc
from utree_amd import synth


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def oracle_of(sdb):
    return orc.OracleDB.from_memory(sdb.W, 2, sdb.binix.cpu().numpy().view(np.uint32).astype(np.uint64), sdb.records.cpu().numpy(),
                                    sdb.label_text)


def assert_records_equal(got, want, n):
    """got: int32 [n, 6] from the GPU; want: the oracle's record arrays."""
    g = got[:n].view(np.uint32)
    assert np.array_equal(g[:, 2], want["found"])
    hit = want["found"] > 0
    multi = hit & (want["uix"] > 1)
    assert np.array_equal(g[hit, 3], want["uix"][hit])
    assert np.array_equal(g[hit, 0], want["label"][hit])
    assert np.array_equal(got[:n][hit, 1], want["cut"][hit])
    assert np.array_equal(g[multi, 4], want["sl"][multi]) and np.array_equal(g[multi, 5], want["ol"][multi])
    return int(hit.sum()), int(multi.sum())


@pytest.mark.parametrize("W,nodes", [(8, 1_217_000_000), (16, 568_000_000)], ids=["config2_k32_1217M_nodes", "config5_k64_568M_nodes"])
def test_150bp_reads_on_the_full_size_database(torch_cuda, W, nodes):
    torch = torch_cuda
    dev = torch.device("cuda:0")
    sdb = synth.make_db(dev, nodes, W=W, keep_raw=True)
    n = 240_000
    reads = synth.make_reads(sdb, n, 150, seed=synth.READ_SEED + 77)
    o = oracle_of(sdb)
    host = reads.bases.cpu().numpy()
    for rc in (False, True):
        got = sdb.tree.classify(reads.bases, reads.off, reads.length, rc=rc)
        again = sdb.tree.classify(reads.bases, reads.off, reads.length, rc=rc)
        assert torch.equal(got, again)                                                  # idempotent
        if rc:
            # strand symmetry: with RC on, the reverse complement of every read classifies identically
            comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=dev)
            for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
                comp[a] = b
            rcseq = comp[reads.bases.view(n, 150).long()].flip(1).contiguous().view(-1)
            other = sdb.tree.classify(rcseq, reads.off, reads.length, rc=True)
            assert torch.equal(got, other)
        # both strand modes against the oracle at the full database size
        want = o.classify_batch(host, np.arange(n, dtype=np.uint64) * 150, np.full(n, 150, dtype=np.uint32), rc=rc, threads=16)
        hits, multi = assert_records_equal(got.cpu().numpy(), want, n)
        assert hits > 0.9 * n and multi > 0.3 * n                                       # the sample exercises the vote
    del o
    perm = torch.randperm(n, device=dev)
    shuffled = sdb.tree.classify(reads.bases, reads.off[perm], reads.length[perm], rc=False)
    assert torch.equal(sdb.tree.classify(reads.bases, reads.off, reads.length, rc=False)[perm], shuffled)   # reads are independent
    sdb.tree.close()


@pytest.mark.parametrize("W,nodes", [(8, 150_000_000), (16, 100_000_000)], ids=["k32", "k64"])
def test_the_automatic_choice_of_line_sized_buckets(torch_cuda, monkeypatch, W, nodes):
    """Beyond 2.2 G 32-mer nodes (1.25 G 64-mer nodes) dev_image.c builds 128-byte buckets by itself (bucket_words_default).  The
    threshold is lowered here (UTREE_BUCKET128_NODES) so that the AUTOMATIC branch -- no UTREE_BUCKET_BYTES -- is the one taken at a size the
    suite can afford, and the image it builds is held to the oracle in both strand modes."""
    torch = torch_cuda
    monkeypatch.delenv("UTREE_BUCKET_BYTES", raising=False)
    monkeypatch.setenv("UTREE_BUCKET128_NODES", str(nodes - 1))
    sdb = synth.make_db(torch.device("cuda:0"), nodes, W=W, keep_raw=True)
    assert sdb.tree.info.bucket_bytes == 128
    monkeypatch.setenv("UTREE_BUCKET128_NODES", str(nodes))
    small = synth.make_db(torch.device("cuda:0"), 2_000_000, W=W)
    assert small.tree.info.bucket_bytes == 64                                           # at or below the threshold: 64-byte buckets
    small.tree.close()
    n = 120_000
    reads = synth.make_reads(sdb, n, 150, seed=synth.READ_SEED + 78)
    o = oracle_of(sdb)
    host = reads.bases.cpu().numpy()
    for rc in (False, True):
        got = sdb.tree.classify(reads.bases, reads.off, reads.length, rc=rc)
        want = o.classify_batch(host, np.arange(n, dtype=np.uint64) * 150, np.full(n, 150, dtype=np.uint32), rc=rc, threads=16)
        hits, multi = assert_records_equal(got.cpu().numpy(), want, n)
        assert hits > 0.9 * n and multi > 0.3 * n
    assert ", 2, " in sdb.tree.kernel_name() and sdb.tree.kernel_name().startswith("classify_lanes_k<%d, 2, 1," % W), sdb.tree.kernel_name()
    sdb.tree.poll()
    sdb.tree.close()


def test_a_batch_of_sixteen_million_reads_equals_its_slices(torch_cuda):
    """bench.py's default launch is 16 M reads (2.4 GB of bases: offsets beyond 2^31).  Every read is independent of its batch: slices from
    the head, the middle and the very end of the big batch, classified as small batches of their own, must give the records the big batch
    gave them, and the last slice also equals the CPU oracle."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    sdb = synth.make_db(dev, 150_000_000, W=8, keep_raw=True)
    n, L, m = 16_000_000, 150, 60_000
    reads = synth.make_reads(sdb, n, L, seed=synth.READ_SEED + 99)
    big = sdb.tree.classify(reads.bases, reads.off, reads.length, rc=False)
    sdb.tree.poll()
    assert sdb.tree.kernel_name().startswith("classify_lanes_k<8, 2, 1,")
    assert int((big[:, 2] > 0).sum().item()) > 0.9 * n
    for lo in (0, 7_999_968, n - m):
        part = sdb.tree.classify(reads.bases[lo * L:(lo + m) * L].contiguous(), reads.off[:m].contiguous(), reads.length[:m].contiguous(), rc=False)
        assert torch.equal(part, big[lo:lo + m]), "slice at %d" % lo
    o = oracle_of(sdb)
    host = reads.bases[(n - m) * L:].cpu().numpy()
    want = o.classify_batch(host, np.arange(m, dtype=np.uint64) * L, np.full(m, L, dtype=np.uint32), rc=False, threads=16)
    hits, multi = assert_records_equal(big[n - m:].cpu().numpy(), want, m)
    assert hits > 0.9 * m and multi > 0.3 * m
    # both strands: the same invariance on the last slice
    big_rc = sdb.tree.classify(reads.bases, reads.off, reads.length, rc=True)
    part = sdb.tree.classify(reads.bases[(n - m) * L:].contiguous(), reads.off[:m].contiguous(), reads.length[:m].contiguous(), rc=True)
    assert torch.equal(part, big_rc[n - m:])
    sdb.tree.poll()
    sdb.tree.close()


def test_long_reads_with_both_strands_on_the_l4_size_database(torch_cuda):
    """configs[2]: 72 M nodes; reads of 1 kb ... 100 kb (hit-dense: a planted k-mer every 32 bases), RC on."""
    torch = torch_cuda
    dev = torch.device("cuda:0")
    sdb = synth.make_db(dev, 72_000_000, W=8, keep_raw=True)
    classes = [(1_000, 24_000), (2_000, 12_000), (3_000, 8_000), (10_000, 12_000), (30_000, 3_000), (100_000, 400)]   # (length, reads): ~25 kb... mean 5 kb
    bases, offs, lens = [], [], []
    at = 0
    for i, (L, cnt) in enumerate(classes):
        r = synth.make_reads(sdb, cnt, L, seed=synth.READ_SEED + 500 + i)
        bases.append(r.bases)
        offs.append(r.off + at)
        lens.append(r.length)
        at += cnt * L
    bases, off, ln = torch.cat(bases), torch.cat(offs), torch.cat(lens)
    n = off.numel()
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(3))      # lengths interleaved, as a file would have them
    off, ln = off[perm].contiguous(), ln[perm].contiguous()
    got = sdb.tree.classify(bases, off, ln, rc=True)
    assert "classify_long_k" in sdb.tree.kernel_name() or sdb.tree.kernel_name().startswith("classify_lanes_k<8, 2, 16, false, 2,")   # long reads: in pieces through the lane pass
    again = sdb.tree.classify(bases, off, ln, rc=True)
    assert torch.equal(got, again)
    o = oracle_of(sdb)
    want = o.classify_batch(bases.cpu().numpy(), off.cpu().numpy().astype(np.uint64), ln.cpu().numpy().astype(np.uint32), rc=True, threads=16)
    hits, multi = assert_records_equal(got.cpu().numpy(), want, n)
    assert hits > 0.9 * n and multi > 0.5 * n
    assert int(want["found"].max()) > 3000                                              # thousands of hits per read: the long tally paths
    sdb.tree.close()


@pytest.mark.parametrize("rc", [0, 1])
def test_hit_dense_related_genomes_database(tmp_path, rc):
    """The shape of a real reference database (README.md:2): related genomes, built by the product's own utree-buildGG +
    xtree-compress -- k-mers of 25 relatives crowd around each minimizer, so half of a read's buckets continue in overflow runs of
    dozens to hundreds of records, and reads hit in a quarter of their windows.  240 000 reads cut from the references: batch
    records against the CPU oracle, the file search's sorted lines against the genuine reference where its binary travelled."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "scale", "hit_dense.py"), "--refs", "100", "--ref-len", "200000",
                        "--reads", "240000", "--sample", "60000", "--oracle-reads", "240000", "--rc", str(rc), "--steps", "2",
                        "--dir", str(tmp_path / "hd")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    j = json.loads(p.stdout[p.stdout.index("{"):])
    assert j["oracle_records_identical"] and j["oracle_reads"] == 240000
    assert j.get("parity_sorted_lines_identical", True)
    assert j["kernel"].startswith("classify_lanes_k<8, 2, 1,")
    assert j["hits_per_read_mean"] > 15 and j["model_counts_100k_reads"]["overflow_buckets"] > 100_000     # hit-dense, and the buckets do overflow
