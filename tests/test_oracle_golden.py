"""Pin the CPU restatement (oracle/) against the genuine reference's outputs in tests/golden/.

Every expected byte here was produced by itree.c -D SEARCH_GG (oracle/_ref, 1 thread) at golden time
(tests/golden/make_golden.py).  CPU only.
"""
import json
import os

import numpy as np
import pytest

from oracle import orc
from utree_amd import ctrfile
import util


@pytest.fixture(scope="module")
def tmpdir_mod(tmp_path_factory):
    return tmp_path_factory.mktemp("orc")


def run_oracle(name, rc, tmpdir, threads=1):
    db = orc.OracleDB.load(util.fixture_ctr(name))
    out = os.path.join(str(tmpdir), "%s_%d.txt" % (name, rc))
    code, nr, good, err = db.search_file(util.fixture_reads_path(util.READS_OF.get(name, name)), out, threads=threads, rc=bool(rc))
    return code, nr, good, open(out, "rb").read()


@pytest.mark.parametrize("name,rcs", [("toy", (0, 1)), ("k64", (0, 1)), ("ix32", (0, 1)), ("k64ix32", (0, 1)), ("k16", (0, 1)), ("vote", (0,)),
                                      ("kat", (0,)), ("katq", (0,)), ("katq2", (0,)), ("generic", (0,))])
def test_search_file_matches_reference(name, rcs, tmpdir_mod):
    for rc in rcs:
        code, nr, good, got = run_oracle(name, rc, tmpdir_mod)
        want = util.fixture_bytes("%s_out%s.txt.gz" % (name, "_rc" if rc else ""))
        assert code == 0
        assert got == want, "%s rc=%d: oracle output differs from the reference's" % (name, rc)
        assert good == want.count(b"\n")


def test_threads_do_not_change_output(tmpdir_mod):
    _, _, _, a = run_oracle("toy", 1, tmpdir_mod, threads=1)
    _, _, _, b = run_oracle("toy", 1, tmpdir_mod, threads=4)
    assert a == b


def test_edge_cases_match_reference(tmpdir_mod):
    cases = json.load(open(os.path.join(util.GOLD, "edge_cases.json")))
    db = orc.OracleDB.load(util.fixture_ctr("toy"))
    assert len(cases) == util.manifest()["edge_cases"]
    for nm, c in sorted(cases.items()):
        fa = os.path.join(str(tmpdir_mod), "edge_%s.fa" % nm)
        open(fa, "wb").write(bytes.fromhex(c["input_hex"]))
        out = fa + ".out"
        code, nr, good, err = db.search_file(fa, out, threads=1, rc=bool(c["rc"]))
        assert code == c["exit"], (nm, code, c["exit"], err)
        assert open(out, "rb").read() == bytes.fromhex(c["output_hex"]), nm


def test_ctr_regenerated_by_our_writer_matches_reference_sha():
    # fixture_ctr asserts the SHA-256 of the reference-built file
    for name in ("toy", "k64", "ix32", "k64ix32", "k16", "vote", "kat", "katq", "katq2", "generic"):
        p = util.fixture_ctr(name)
        d = ctrfile.read_ctr(p)
        assert d.n_nodes == util.manifest()[name + "_nodes"]
        assert len(d.labels()) == util.manifest()[name + "_labels"]


def test_compress_style_bin_table_reproduces_reference_compress():
    # toy.ctr came out of the reference's COMPRESS: our like_compress bin table must equal its table
    for name in ("toy", "k64", "ix32", "k64ix32"):
        d = util.load_db_fixture(name)
        hi, lo = d.words()
        pref = ctrfile.word_prefix(d.W, hi, lo)
        # words() needs an exact table to recover prefixes; for reference-built DBs the quirk only
        # matters when the first bin holds one record, so compare through the generic routine
        b = ctrfile.binix_like_compress(pref, d.n_nodes)
        assert np.array_equal(b, d.binix)


def test_lookup_kat_direct():
    d = util.load_db_fixture("kat")
    db = orc.OracleDB.load(util.fixture_ctr("kat"))
    hi, lo = d.words()
    ix = d.ix()
    for j in range(0, len(lo), 7):
        assert db.lookup(0, int(lo[j])) == int(ix[j])
        miss = int(lo[j]) ^ 1
        if miss not in set(int(x) for x in lo[max(0, j - 2):j + 3]):
            assert db.lookup(0, miss) == orc.BAD_IX


def test_windows_skip_bad_bases():
    seq = b"ACGT" * 10 + b"N" + b"ACGT" * 10
    pos, hi, lo = orc.windows(seq, 32)
    # windows ending at 31..39 (9 of them) then from 41+31=72 .. 80
    assert list(pos) == list(range(31, 40)) + list(range(72, 81))
    assert int(lo[0]) == int("".join(format("ACGT".index(c), "02b") for c in (b"ACGT" * 8).decode()), 2)
    assert len(orc.windows(b"ACGT" * 7 + b"ACG", 32)[0]) == 0
    # lowercase and high bytes
    pos2, _, lo2 = orc.windows((b"acgt" * 8) + bytes([200]) + b"A" * 32, 32)
    assert list(pos2) == [31, 64] and int(lo2[1]) == 0


# ---- rank-specific search (`xtree-search`, itree.c -D SEARCH): SURVEY §8(f) rank 1 ---------------------
RANK_TAGS = sorted(util.manifest().get("rank_outputs", {}))


@pytest.mark.parametrize("tag", RANK_TAGS)
def test_rank_search_file_matches_reference(tag, tmpdir_mod):
    """Every byte of the genuine binaries' outputs (5 SLACK/SPARSITY/TOLERANCE builds, k=64 and u32-label builds,
    +RC), including the order dependence through the hit array the reference never clears (itree.c:982)."""
    v = util.manifest()["rank_outputs"][tag]
    db = orc.OracleDB.load(util.fixture_ctr(v["db"]))
    out = os.path.join(str(tmpdir_mod), tag + ".txt")
    sl, sp, tol = v["params"]
    code, nr, good, err = orc.rank_search_file(db, util.fixture_reads_path(v["reads"]), out, rc=bool(v["rc"]),
                                               slack=sl, sparsity=sp, tolerance=tol)
    want = util.fixture_bytes(tag + ".txt.gz")
    assert code == 0
    assert open(out, "rb").read() == want
    assert good == want.count(b"\n") == v["lines"]


def test_rank_vote_reads_one_stale_entry():
    """The carried entry in isolation: a read with ONE hit is printed iff the entry an earlier, longer hit list
    left at index 1 (or the initial 0) is the same label -- (most, second) = (2, 0)."""
    d = util.load_db_fixture("vote")
    db = orc.OracleDB.load(util.fixture_ctr("vote"))
    hi, lo = d.suffixes()
    ixs = d.ix()
    # full words: bin prefix (24 bits) + stored suffix
    b = d.binix
    prefix = np.searchsorted(b, np.arange(d.n_nodes), side="right") - 1
    words = (prefix.astype(np.uint64) << np.uint64(40)) | lo
    kmer_of = {}
    for w, i in zip(words, ixs):
        kmer_of.setdefault(int(i), ctrfile.decode_kmer(0, int(w), 32))
    labs = sorted(kmer_of)
    a, c = [l for l in labs if l != 0][:2]
    rs = orc.RankSearch(db)
    r = rs.read(kmer_of[a].encode())                      # first read: entry [1] is still 0 -> label 0 gets the vote
    assert (r.found, r.most, r.second, r.printed) == (1, 1, 1, 0)
    two = (kmer_of[c] + "N" + kmer_of[a]).encode()        # leaves [c, a] in the array
    r = rs.read(two)
    assert r.found == 2
    r = rs.read(kmer_of[a].encode())                      # one hit (a) + stale entry [1] = a  -> printed, 2 votes
    assert (r.found, r.label, r.most, r.second, r.printed) == (1, a, 2, 0, 1)
    assert rs.format(b"q", r) == b"q\t" + db.label(a) + b"\t1.000000\t2\n"
    r = rs.read(kmer_of[c].encode())                      # one hit (c) + stale a -> 1 : 1, not printed
    assert (r.found, r.most, r.second, r.printed) == (1, 1, 1, 0)


# ---- database BUILD (`utree-build`, `utree-buildGG`): SURVEY §8(f) rank 3 ------------------------------
BUILD_TAGS = sorted(util.manifest().get("build_outputs", {}))


def build_inputs(setname, tmpdir):
    fa = os.path.join(str(tmpdir), setname + ".fa")
    mp = os.path.join(str(tmpdir), setname + ".map")
    if not os.path.exists(fa):
        open(fa, "wb").write(util.fixture_bytes("build_%s.fa.gz" % setname))
        open(mp, "wb").write(util.fixture_bytes("build_%s.map.gz" % setname))
    return fa, mp


@pytest.mark.parametrize("tag", BUILD_TAGS)
def test_build_matches_reference(tag, tmpdir_mod):
    """`.ubt` and `[.gg].log` byte-identical (SHA-256) to what the genuine builders wrote: complevel 0/1/2/4, the
    rank-specific BUILD, PACKSIZE=64, IXTYPE=uint32_t, collision chains down to BAD, and the exit codes of bad inputs."""
    v = util.manifest()["build_outputs"][tag]
    fa, mp = build_inputs(v["set"], tmpdir_mod)
    ubt = os.path.join(str(tmpdir_mod), tag + ".ubt")
    code, ns, nn, nl, err = orc.build_file(fa, mp, ubt, W=v["W"], I=v["I"], complevel=v["complevel"], gg=bool(v["gg"]))
    assert code == v["exit"], err
    if code == 0:
        assert ctrfile.sha256_file(ubt) == v["ubt_sha256"]
        assert ctrfile.sha256_file(ubt + (".gg.log" if v["gg"] else ".log")) == v["log_sha256"]
        assert ("Total nodes in tree: %d [%d labels]" % (nn, nl)) in v["stdout_tail"]


@pytest.mark.skipif(not util.have_ref(), reason="needs oracle/_ref (the genuine reference, built from /root/reference by `make -C oracle ref`)")
@pytest.mark.parametrize("seed,max_depth", [(1, 8), (2, 8), (3, 8), (4, 7), (5, 7), (6, 5)])
def test_vote_adversarial_cases_oracle_vs_reference(seed, max_depth, tmp_path):
    """The label sets the GPU vote is stressed with (util.adversarial_vote_case) go beyond the committed vote fixture, so
    the oracle is first held against the genuine reference binary on them, where that binary exists."""
    import subprocess
    ctr_path, data, n_reads = util.adversarial_vote_case(seed, str(tmp_path), max_depth)
    fa = tmp_path / "r.fa"
    fa.write_bytes(data)
    want = tmp_path / "ref.txt"
    r = subprocess.run([os.path.join(util.REF_DIR, "xtree-searchGG"), ctr_path, str(fa), str(want), "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    o = orc.OracleDB.load(ctr_path)
    got = tmp_path / "orc.txt"
    code, nr, good, err = o.search_file(str(fa), str(got), threads=4, rc=False)
    assert code == 0 and nr == n_reads
    assert got.read_bytes() == want.read_bytes() and good > 1000


@pytest.mark.skipif(not util.have_ref(), reason="needs oracle/_ref (the genuine reference, built from /root/reference by `make -C oracle ref`)")
def test_rank_search_adversarial_case_oracle_vs_reference(tmp_path):
    """Same hostile label set through the rank-specific search (`xtree-search`): whole output files identical."""
    import subprocess
    ctr_path, data, n_reads = util.adversarial_vote_case(2, str(tmp_path))
    fa = tmp_path / "r.fa"
    fa.write_bytes(data)
    want = tmp_path / "ref.txt"
    r = subprocess.run([os.path.join(util.REF_DIR, "xtree-search"), ctr_path, str(fa), str(want), "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    got = tmp_path / "orc.txt"
    code, nr, good, err = orc.rank_search_file(orc.OracleDB.load(ctr_path), str(fa), str(got))
    assert code == 0 and nr == n_reads and good > 100
    assert got.read_bytes() == want.read_bytes()


@pytest.mark.skipif(not util.have_ref(), reason="needs oracle/_ref (the genuine reference, built from /root/reference by `make -C oracle ref`)")
@pytest.mark.parametrize("name,with_gt", [("toy", False), ("toy", True), ("k64", False), ("ix32", False), ("k64ix32", False), ("vote", False), ("k16", False)])
def test_byte_fuzz_oracle_vs_reference(name, with_gt, tmp_path):
    """Sequence lines with arbitrary 7-bit bytes, both strands: output file AND exit code of the oracle equal the genuine
    reference's (with '>' allowed, both stop at the same record with the format error)."""
    import subprocess
    data = util.byte_fuzz_reads(name, 21, with_gt)
    fa = tmp_path / "f.fa"
    fa.write_bytes(data)
    exe = "xtree-searchGG" + {"k64": "-k64", "ix32": "-ix32", "k64ix32": "-k64-ix32", "k16": "-k16"}.get(name, "")
    o = orc.OracleDB.load(util.fixture_ctr(name))
    for rc in (False, True):
        want, got = tmp_path / "ref.txt", tmp_path / "orc.txt"
        r = subprocess.run([os.path.join(util.REF_DIR, exe), util.fixture_ctr(name), str(fa), str(want), "1"] + (["RC"] if rc else []),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        code, nr, good, err = o.search_file(str(fa), str(got), threads=4, rc=rc)
        assert code == r.returncode and (code == 2) == with_gt
        assert got.read_bytes() == want.read_bytes() and good > 10


@pytest.mark.skipif(not util.have_ref(), reason="needs oracle/_ref (the genuine reference, built from /root/reference by `make -C oracle ref`)")
@pytest.mark.parametrize("exe,W,gg", [("utree-buildGG", 8, 1), ("utree-build", 8, 0), ("utree-buildGG-k64", 16, 1)])
def test_build_hostile_labels_oracle_vs_reference(exe, W, gg, tmp_path):
    """BUILD / BUILD_GG with hostile labels (util.hostile_build_case): `.ubt` and log of the oracle equal the genuine reference's
    at every compression level."""
    import subprocess
    for seed in (1, 2):
        fa_b, mp_b = util.hostile_build_case(seed)
        fa = tmp_path / "i.fa"; mp = tmp_path / "i.map"
        fa.write_bytes(fa_b); mp.write_bytes(mp_b)
        for cl in (0, 1, 2):
            want, got = str(tmp_path / "r.ubt"), str(tmp_path / "o.ubt")
            r = subprocess.run([os.path.join(util.REF_DIR, exe), str(fa), str(mp), want, "1", str(cl)], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=600)
            assert r.returncode == 0
            code, ns, nn, nl, err = orc.build_file(str(fa), str(mp), got, W=W, I=2, complevel=cl, gg=bool(gg))
            assert code == 0, err
            ext = ".gg.log" if gg else ".log"
            assert open(got, "rb").read() == open(want, "rb").read()
            assert open(got + ext, "rb").read() == open(want + ext, "rb").read()


@pytest.mark.skipif(not util.have_ref(), reason="needs oracle/_ref (the genuine reference, built from /root/reference by `make -C oracle ref`)")
def test_framing_fuzz_oracle_vs_reference(tmp_path):
    """Sixteen random files from the grammar of framing corner cases (util.framing_fuzz_case; 150 were run once while writing
    this test), both strands: the oracle's exit code and output file equal the genuine reference's."""
    import subprocess
    o = orc.OracleDB.load(util.fixture_ctr("toy"))
    for seed in range(16):
        fa = tmp_path / "f.fa"
        fa.write_bytes(util.framing_fuzz_case(seed))
        for rc in (False, True):
            want, got = tmp_path / "ref.txt", tmp_path / "orc.txt"
            for f in (want, got):
                if f.exists():
                    f.unlink()
            r = subprocess.run([os.path.join(util.REF_DIR, "xtree-searchGG"), util.fixture_ctr("toy"), str(fa), str(want), "1"] + (["RC"] if rc else []),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            code, nr, good, err = o.search_file(str(fa), str(got), threads=2, rc=rc)
            assert code == r.returncode, seed
            assert (got.read_bytes() if got.exists() else None) == (want.read_bytes() if want.exists() else None), seed


@pytest.mark.skipif(not util.have_ref(), reason="needs oracle/_ref (the genuine reference, built from /root/reference by `make -C oracle ref`)")
@pytest.mark.parametrize("case", ["quirk", "dups", "generic"])
def test_packsize16_irregular_tables_oracle_vs_reference(case, tmp_path):
    """PACKSIZE=16 trees off the regular shape (util.k16_table_cases): the oracle equals the genuine reference built with -D PACKSIZE=16,
    both strand modes -- which pins the oracle the GPU's direct-address table is held to in tests/test_gpu_parity.py."""
    import subprocess
    ctr, data = util.k16_table_cases(str(tmp_path))[case]
    fa = tmp_path / "r.fa"
    fa.write_bytes(data)
    o = orc.OracleDB.load(ctr)
    for rc in (False, True):
        want, got = tmp_path / "ref.txt", tmp_path / "orc.txt"
        r = subprocess.run([os.path.join(util.REF_DIR, "xtree-searchGG-k16"), ctr, str(fa), str(want), "1"] + (["RC"] if rc else []),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        code, nr, good, err = o.search_file(str(fa), str(got), threads=4, rc=rc)
        assert r.returncode == 0 and code == 0
        assert got.read_bytes() == want.read_bytes() and good > 100
