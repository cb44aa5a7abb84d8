"""CPU-only tests of the product's host side: C-ABI surface, loader, framing, formatting.
(No compute entry point is called here: those need a GPU and are covered by -m gpu tests.)"""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from oracle import orc
from utree_amd import ctrfile, lib
from utree_amd.search import CtrDB, frame_fasta
import util

ROOT = util.ROOT


def test_library_exports_every_declared_symbol():
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "utree_amd.h")).read()
    declared = set(re.findall(r"\b(utree_[a-z0-9_]+)\(", hdr))
    # types that look like calls in comments are not functions
    declared -= {"utree_ctr_from_memory_"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), "libutree_amd.so does not export %s" % name
        assert name in lib.SYMBOLS, "python binding lacks %s" % name
    assert L.utree_abi_version() == 4
    assert L.utree_strerror(lib.E_FORMAT) == b"Tree malformatted."


def test_product_does_not_reference_oracle():
    # the product must never import / link / execute anything under oracle/
    for base, _, files in os.walk(os.path.join(ROOT, "utree_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "liboracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f
                assert "orc_" not in txt, f


@pytest.mark.parametrize("name", ["toy", "k64", "ix32", "k16", "vote", "kat", "katq"])
def test_loader_matches_oracle_and_fixture(name):
    p = util.fixture_ctr(name)
    db = CtrDB.open(p)
    o = orc.OracleDB.load(p)
    d = util.load_db_fixture(name)
    assert (db.W, db.I, db.n_nodes) == (o.W, o.I, o.n_nodes) == (d.W, d.I, d.n_nodes)
    assert db.info.SZ == d.SZ and db.info.k == d.k and db.info.binix_width == 4
    assert db.n_labels == o.n_labels == len(d.labels())
    for ix in range(0, db.n_labels, max(1, db.n_labels // 50)):
        assert db.label(ix) == o.label(ix) == d.labels()[ix].encode("latin-1")
    assert db.label(db.n_labels) is None
    assert db.info.bin_total == int(d.binix[-1])


def test_loader_errors(tmp_path):
    L = lib.load()
    h = C.c_void_p()
    assert L.utree_ctr_open(str(tmp_path / "nope.ctr").encode(), C.byref(h)) == lib.E_IO      # itree.c:735
    p = tmp_path / "short.ctr"
    p.write_bytes(b"\x08" + b"\0" * 10)
    assert L.utree_ctr_open(str(p).encode(), C.byref(h)) == lib.E_FORMAT                         # itree.c:738
    p.write_bytes(np.array([8, 0, 2, 0], dtype="<u8").tobytes())
    assert L.utree_ctr_open(str(p).encode(), C.byref(h)) == lib.E_FORMAT                         # zero nodes
    p.write_bytes(np.array([2, 0, 2, 5], dtype="<u8").tobytes())
    assert L.utree_ctr_open(str(p).encode(), C.byref(h)) == lib.E_UNSUPPORTED                    # PACKSIZE=8
    p.write_bytes(np.array([8, 4, 2, 5], dtype="<u8").tobytes())
    assert L.utree_ctr_open(str(p).encode(), C.byref(h)) == lib.E_UNSUPPORTED                    # counts
    p.write_bytes(np.array([8, 0, 2, 5], dtype="<u8").tobytes() + b"\0" * 100)
    assert L.utree_ctr_open(str(p).encode(), C.byref(h)) == lib.E_FORMAT                         # short bin table
    # truncated node dump (itree.c:768) and missing labels (itree.c:776)
    d = util.load_db_fixture("kat")
    full = open(util.fixture_ctr("kat"), "rb").read()
    body = 32 + 4 * ctrfile.NUMBINS + d.n_nodes * d.SZ
    p.write_bytes(full[: body - 3])
    assert L.utree_ctr_open(str(p).encode(), C.byref(h)) == lib.E_FORMAT
    p.write_bytes(full[:body])
    assert L.utree_ctr_open(str(p).encode(), C.byref(h)) == lib.E_NOLABELS


def test_duplicate_labels_collapse_to_first_index():
    d = util.load_db_fixture("kat")
    text = b"lab_b\t1\nlab_a\t2\nlab_b\t3\nlab_c\t9\nlab_a\n"
    db = CtrDB.from_memory(d.W, d.I, d.n_nodes, d.binix, d.records, text)
    o = orc.OracleDB.from_memory(d.W, d.I, d.binix, d.records, text)
    assert db.n_labels == o.n_labels == 3
    assert [db.label(i) for i in range(3)] == [b"lab_b", b"lab_a", b"lab_c"] == [o.label(i) for i in range(3)]


def test_framing_matches_reference_edge_cases():
    cases = json.load(open(os.path.join(util.GOLD, "edge_cases.json")))
    code_to_exit = {0: 0, 1: 2, 2: 2, 3: 2, 4: 2}
    for nm, c in sorted(cases.items()):
        if c["rc"]:
            continue
        data = bytes.fromhex(c["input_hex"])
        fr = frame_fasta(data, final=True)
        assert code_to_exit[fr["error_code"]] == c["exit"], nm
        # names of the lines the reference printed must be a subsequence of the framed names
        names = [data[o:o + l] for o, l in zip(fr["name_off"], fr["name_len"])]
        out_names = [ln.rsplit(b"\t", 4)[0] for ln in bytes.fromhex(c["output_hex"]).split(b"\n") if ln]
        it = iter(names)
        assert all(any(n == x for x in it) for n in out_names), nm


def test_framing_chunked_equals_whole():
    data = util.fixture_bytes("toy_reads.fa.gz")[:200000]
    whole = frame_fasta(data, final=True)
    # feed in two pieces at an arbitrary split: first non-final, then the rest final
    for split in (1, 77, 5000, 123457):
        a = frame_fasta(data[:split], final=False)
        rest = data[a["consumed"]:]
        b = frame_fasta(rest, final=True)
        n1 = len(a["seq_off"])
        assert n1 + len(b["seq_off"]) == len(whole["seq_off"])
        assert np.array_equal(a["seq_off"], whole["seq_off"][:n1])
        assert np.array_equal(b["seq_off"] + a["consumed"], whole["seq_off"][n1:])
        assert np.array_equal(np.concatenate([a["seq_len"], b["seq_len"]]), whole["seq_len"])


def test_format_matches_oracle_format():
    p = util.fixture_ctr("toy")
    db = CtrDB.open(p)
    o = orc.OracleDB.load(p)
    data = util.fixture_bytes("toy_reads.fa.gz")
    fr = frame_fasta(data, final=True)
    buf = np.frombuffer(data, dtype=np.uint8)
    res = o.classify_batch(buf, fr["seq_off"], fr["seq_len"], rc=False, threads=4)      # oracle results as input
    got = db.format(buf, fr["name_off"], fr["name_len"], res)
    assert got == util.fixture_bytes("toy_out.txt.gz")


def test_rank_format_matches_reference_lines():
    """Host formatting of the rank-specific lines (itree.c:1002, "%f" and "%d") with the oracle's records as input:
    reproduces the genuine xtree-search output."""
    p = util.fixture_ctr("toy")
    db = CtrDB.open(p)
    o = orc.OracleDB.load(p)
    data = util.fixture_bytes("toy_reads.fa.gz")
    fr = frame_fasta(data, final=True)
    buf = np.frombuffer(data, dtype=np.uint8)
    rs = orc.RankSearch(o)
    res = np.zeros((len(fr["seq_off"]), 6), dtype=np.int32)
    for i in range(len(res)):
        s, l = int(fr["seq_off"][i]), int(fr["seq_len"][i])
        r = rs.read(data[s:s + l])
        res[i] = (r.label, -2 if r.printed else -4, r.found, 0, r.most, r.second)
    assert db.format(buf, fr["name_off"], fr["name_len"], res, rank=True) == util.fixture_bytes("toy_rank.txt.gz")


def test_synthetic_ctr_roundtrip(tmp_path):
    rng = np.random.default_rng(5)
    lo = np.unique(rng.integers(0, 1 << 63, size=5000, dtype=np.uint64))
    ix = rng.integers(0, 7, size=len(lo)).astype(np.uint32)
    labels = ["k__X;p__%d" % i for i in range(7)]
    p = str(tmp_path / "s.ctr")
    ctrfile.write_ctr(p, 8, 2, np.zeros_like(lo), lo, ix, labels)
    d = ctrfile.read_ctr(p)
    hi2, lo2 = d.words()
    assert np.array_equal(lo2, lo) and np.array_equal(d.ix(), ix) and d.labels() == labels
    o = orc.OracleDB.load(p)
    for j in range(0, len(lo), 97):
        assert o.lookup(0, int(lo[j])) == int(ix[j])


def _frame_both(data, final):
    """parallel (default) and serial (UTREE_FRAME_SERIAL) framing of the same bytes"""
    os.environ.pop("UTREE_FRAME_SERIAL", None)
    a = frame_fasta(data, final=final)
    os.environ["UTREE_FRAME_SERIAL"] = "1"
    try:
        b = frame_fasta(data, final=final)
    finally:
        os.environ.pop("UTREE_FRAME_SERIAL", None)
    return a, b


def test_parallel_framing_equals_serial_framing():
    rng = np.random.default_rng(11)
    parts = []
    for i in range(150_000):
        L = int(rng.integers(1, 400)) if rng.random() < 0.98 else int(rng.integers(2000, 60000))
        seq = bytes(rng.integers(65, 85, L, dtype=np.uint8))
        nl = b"\r\n" if rng.random() < 0.05 else b"\n"
        hdr = b">r%d" % i + (b" desc with spaces" if rng.random() < 0.1 else b"") + (b"\tx" if rng.random() < 0.02 else b"")
        if rng.random() < 0.002:
            seq = b""                                   # blank sequence line: framed with length 0, no error
        parts.append(hdr + nl + seq + nl)
    data = b"".join(parts)
    assert len(data) > (32 << 20)
    for final, blob in ((True, data), (True, data[:-1]), (False, data[:-777]), (False, data)):
        a, b = _frame_both(blob, final)
        for k in ("seq_off", "seq_len", "name_off", "name_len"):
            assert np.array_equal(a[k], b[k]), (k, final)
        assert a["consumed"] == b["consumed"] and a["error_code"] == b["error_code"] == 0
    # injected errors: the earliest one wins, reads before it are framed identically
    for where, junk in ((len(data) // 3, b"\n"), (len(data) // 2, b">oops\n"), (len(data) * 3 // 4, b"xx\n")):
        cut = data.index(b"\n>", where) + 1
        blob = data[:cut] + junk + data[cut:]
        a, b = _frame_both(blob, True)
        assert a["error_code"] == b["error_code"] != 0 and a["error_read"] == b["error_read"]
        assert len(a["seq_off"]) == len(b["seq_off"]) and np.array_equal(a["seq_off"], b["seq_off"])
    # odd number of lines at the end: "can't read sequence"
    a, b = _frame_both(data + b">last\n", True)
    assert a["error_code"] == b["error_code"] == 1 and len(a["seq_off"]) == len(b["seq_off"]) == 150_000


# ---- opt-in input formats: same reads as FASTQ / multi-line FASTA frame to the same sequences ----
def _as_fastq(reads):
    return b"".join(b"@" + n + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n" for n, s in reads)


def _as_multiline(reads, width=37, crlf=False):
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for n, s in reads:
        out.append(b">" + n + nl)
        for a in range(0, max(len(s), 1), width):
            out.append(s[a:a + width] + nl)
    return b"".join(out)


def test_fastq_and_multiline_fasta_frame_like_two_line_fasta():
    from utree_amd import lib
    from utree_amd.search import frame_reads
    data = util.fixture_bytes("toy_reads.fa.gz")
    fr = frame_fasta(data, final=True)
    reads = []
    for i in range(0, len(fr["seq_off"]), 7):
        s, l = int(fr["seq_off"][i]), int(fr["seq_len"][i])
        no, nl = int(fr["name_off"][i]), int(fr["name_len"][i])
        reads.append((data[no:no + nl] + b" some comment", data[s:s + l]))
    for fmt, blob in ((lib.INPUT_FASTQ, _as_fastq(reads)), (lib.INPUT_FASTA_MULTILINE, _as_multiline(reads)),
                      (lib.INPUT_FASTA_MULTILINE, _as_multiline(reads, 1000)), (lib.INPUT_FASTA_MULTILINE, _as_multiline(reads, 11, True))):
        g = frame_reads(blob, fmt)
        assert g["error_code"] == 0 and len(g["seq_off"]) == len(reads) and g["consumed"] == len(blob)
        b = g["buf"].tobytes()
        for i, (n, s) in enumerate(reads):
            so, sl = int(g["seq_off"][i]), int(g["seq_len"][i])
            no, nl = int(g["name_off"][i]), int(g["name_len"][i])
            assert b[so:so + sl] == s and b[no:no + nl] == n.split(b" ")[0]
    # chunked input: nothing is consumed past the last complete record, and the rest frames the same later
    blob = _as_multiline(reads)
    cut = len(blob) // 2
    g = frame_reads(blob[:cut], lib.INPUT_FASTA_MULTILINE, final=False)
    assert 0 < g["consumed"] <= cut and blob[g["consumed"]:g["consumed"] + 1] == b">"
    g2 = frame_reads(blob[g["consumed"]:], lib.INPUT_FASTA_MULTILINE, final=True)
    assert len(g["seq_off"]) + len(g2["seq_off"]) == len(reads)
    q = _as_fastq(reads)
    g = frame_reads(q[: len(q) // 3], lib.INPUT_FASTQ, final=False)
    assert q[g["consumed"]:g["consumed"] + 1] == b"@"
    # malformed
    assert frame_reads(b"ACGT\nACGT\n", lib.INPUT_FASTQ)["error_code"] == 2
    assert frame_reads(b"@r\nACGT\nACGT\nIIII\n", lib.INPUT_FASTQ)["error_code"] == 3
    assert frame_reads(b"@r\nACGT\n", lib.INPUT_FASTQ)["error_code"] in (1, 3)


def test_fasta_tensor_equals_the_python_loop():
    """synth.fasta_tensor (the bench's FASTA writer, torch ops) == synth.reads_to_fasta, across a change of the index's digit count."""
    import torch
    from utree_amd import synth
    g = torch.Generator().manual_seed(3)
    L, n = 37, 250
    bases = torch.tensor(list(b"ACGTN"), dtype=torch.uint8)[torch.randint(0, 5, (n * L,), generator=g)]
    r = synth.SynthReads(bases=bases, off=None, length=None, n=n, read_len=L)
    for first in (0, 95, 999_900):
        assert bytes(synth.fasta_tensor(r, first).numpy().tobytes()) == synth.reads_to_fasta(r, first)


def test_kernel_source_hash_follows_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py uses a kept profile only for the library built from exactly the profiled kernel sources: the hash must change with them."""
    import shutil
    from utree_amd import lib as ulib
    a = ulib.kernel_source_sha256()
    assert a == ulib.kernel_source_sha256() and len(a) == 64
    fake = tmp_path / "pkg"
    shutil.copytree(os.path.join(os.path.dirname(ulib.__file__), "csrc"), fake / "csrc", ignore=shutil.ignore_patterns("*.o"))
    monkeypatch.setattr(ulib, "_HERE", str(fake))
    assert ulib.kernel_source_sha256() == a
    with open(fake / "csrc" / "kernels.hip", "a") as f:
        f.write("// touched\n")
    assert ulib.kernel_source_sha256() != a


def test_profile_entries_name_their_kernel_and_sources():
    """profiles/traffic.json (what bench.py reads roofline.traffic from): every entry carries the kernel signature, the hash of
    the kernel sources it was profiled from and the counters the fractions are computed from."""
    import json
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert tj
    for key, e in tj.items():
        assert key.startswith("nodes=") and ",k=" in key and ",rc=" in key
        assert "classify_" in e["kernel"] and len(e["kernel_source_sha256"]) == 64
        assert e["hbm_bytes_per_launch"] > 0 and e["avg_launch_ms"] > 0 and e["SQ_INSTS_VALU_per_launch"] > 0


def test_fanout_falls_back_to_uploads_when_the_broadcast_fails(capfd):
    """utree_dev_fanout (what the command line calls with more than one GPU): the RCCL broadcast, and when that
    returns an error a warning and one utree_dev_upload per other device -- driven here, without a GPU, through the seam
    utree_dev_fanout_with with the two steps passed in."""
    L = lib.load()
    n = 4
    devs = (C.c_int * n)(0, 1, 2, 3)
    calls = {"rep": 0, "up": []}
    dev0 = C.c_void_p(0x1000)

    def rep_fail(ctr, d0, devices, k, out):
        calls["rep"] += 1
        assert d0 == dev0.value and k == n and [devices[i] for i in range(k)] == [0, 1, 2, 3]
        return lib.E_RCCL

    def rep_ok(ctr, d0, devices, k, out):
        calls["rep"] += 1
        for i in range(1, k):
            out[i] = 0x2000 + i
        return lib.OK

    def up(ctr, device, fine_bits, out):
        calls["up"].append((device, fine_bits))
        out[0] = 0x3000 + device
        return lib.OK

    out = (C.c_void_p * n)()
    how = C.c_int(-1)
    rc = L.utree_dev_fanout_with(None, dev0, devs, n, lib.FINE_AUTO, out, C.byref(how), C.cast(lib.REPLICATE_FN(rep_fail), C.c_void_p),
                                 C.cast(lib.UPLOAD_FN(up), C.c_void_p))
    assert rc == lib.OK and how.value == lib.FANOUT_UPLOAD and calls["rep"] == 1
    assert calls["up"] == [(1, lib.FINE_AUTO), (2, lib.FINE_AUTO), (3, lib.FINE_AUTO)]
    assert [out[i] for i in range(n)] == [0x1000, 0x3001, 0x3002, 0x3003]
    assert "RCCL broadcast of the tree failed" in capfd.readouterr().err
    calls["up"].clear()
    rc = L.utree_dev_fanout_with(None, dev0, devs, n, lib.FINE_AUTO, out, C.byref(how), C.cast(lib.REPLICATE_FN(rep_ok), C.c_void_p),
                                 C.cast(lib.UPLOAD_FN(up), C.c_void_p))
    assert rc == lib.OK and how.value == lib.FANOUT_BROADCAST and calls["up"] == []
    assert [out[i] for i in range(n)] == [0x1000, 0x2001, 0x2002, 0x2003]
    assert "failed" not in capfd.readouterr().err
    # a bad argument is not a transport failure: no fallback
    rc = L.utree_dev_fanout_with(None, dev0, devs, n, lib.FINE_AUTO, out, C.byref(how),
                                 C.cast(lib.REPLICATE_FN(lambda *a: lib.E_ARG), C.c_void_p), C.cast(lib.UPLOAD_FN(up), C.c_void_p))
    assert rc == lib.E_ARG and calls["up"] == []


def test_image_size_follows_the_region_model(monkeypatch):
    """utree_dev_image_bytes = the layout the loader builds in (dev_image.c: compute_regions + layout), evaluated on the host.  The table is
    sized region by region from the minimizer-hash density and, since image version 11, from the lumps that half-occupied hash values make
    (only every other hash value is some canonical 16-mer's): BASELINE's config 2 needs 40-odd GiB built (its final image: 40.8), without
    the lump model a third less at four times the overflow, a PACKSIZE=16 tree 8 GiB whatever its size."""
    L = lib.load()
    bins = np.zeros((1 << 24) + 1, dtype=np.uint64)

    def image_gib(W, n):
        bins[-1] = n
        db = CtrDB.from_memory(W, 2, n, bins, None, b"k__A;p__B\t1\n")
        g = L.utree_dev_image_bytes(db._h, lib.FINE_AUTO) / 2.0**30
        db.close()
        return g
    monkeypatch.delenv("UTREE_BUCKET_BYTES", raising=False)
    monkeypatch.delenv("UTREE_LUMP_SLACK", raising=False)
    c2 = image_gib(8, 1_217_000_000)
    assert 55 < c2 < 66                                    # table 40.6 + build areas (all nodes' records twice)
    monkeypatch.setenv("UTREE_LUMP_SLACK", "0")
    assert c2 - 18 < image_gib(8, 1_217_000_000) < c2 - 12  # the naive sizing: 15 GiB less table
    monkeypatch.delenv("UTREE_LUMP_SLACK")
    assert image_gib(8, 2_000_000) < 3.0                   # a toy: the floor of 2^16 pairs per region
    sizes = [image_gib(8, n) for n in (100_000_000, 300_000_000, 600_000_000, 1_217_000_000)]
    assert sizes == sorted(sizes)
    assert 8.0 < image_gib(4, 1000) < 8.3 and 8.0 < image_gib(4, 3_000_000_000) < 8.0 + 3 * 8 * 3.1   # PACKSIZE=16: 2^32 ranks + the records while building


def test_k64_sub_slices_in_the_size_model(monkeypatch):
    """Image version 13, k = 64: where one hash value holds more nodes than a bucket, its slot is several pairs of buckets (dev_image.c:
    compute_regions) -- as long as the table stays within UTREE_TABLE_MAX_GB; beyond the cap the slots stay single pairs, and only then is
    the table coarsened.  BASELINE's configs[4] (568 M 64-mers): 21 GiB more table with the sub-slices; none at a tenth of the size,
    where no hash value is that crowded."""
    L = lib.load()
    bins = np.zeros((1 << 24) + 1, dtype=np.uint64)

    def image_gib(n):
        bins[-1] = n
        db = CtrDB.from_memory(16, 2, n, bins, None, b"k__A;p__B\t1\n")
        g = L.utree_dev_image_bytes(db._h, lib.FINE_AUTO) / 2.0**30
        db.close()
        return g
    for v in ("UTREE_BUCKET_BYTES", "UTREE_LUMP_SLACK", "UTREE_SUB_SLICES", "UTREE_TABLE_MAX_GB", "UTREE_TEST_SUB"):
        monkeypatch.delenv(v, raising=False)
    with_sub = image_gib(568_000_000)
    monkeypatch.setenv("UTREE_SUB_SLICES", "0")
    without = image_gib(568_000_000)
    assert 19 < with_sub - without < 23                    # table 47.1 against 26.6 GiB
    small_without = image_gib(56_000_000)
    monkeypatch.delenv("UTREE_SUB_SLICES")
    assert image_gib(56_000_000) == small_without
    monkeypatch.setenv("UTREE_TABLE_MAX_GB", "40")         # the sub-sliced table (47 GiB) is over the cap, the plain one is not
    assert image_gib(568_000_000) == without
    monkeypatch.setenv("UTREE_TABLE_MAX_GB", "20")         # ... and now the plain one is too: coarser slots
    assert image_gib(568_000_000) < without - 4
