"""GPU parity of the database BUILD (`utree-build`, `utree-buildGG`) through the C-ABI:
`.ubt` and `[.gg].log` byte-identical (SHA-256) to what the genuine builders wrote, and the reference's exit conditions.
Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import orc
from utree_amd import ctrfile, lib
from utree_amd.search import build, compress
import util

BUILDS = util.manifest().get("build_outputs", {})
EXIT_OF = {lib.BUILD_E_MAP_EMPTY: 1, lib.BUILD_E_MAP: 2, lib.BUILD_E_FASTA: 2, lib.BUILD_E_NO_KMERS: 2, lib.BUILD_E_NAME: 4}


def inputs(setname, tmp_path):
    fa = tmp_path / (setname + ".fa")
    mp = tmp_path / (setname + ".map")
    fa.write_bytes(util.fixture_bytes("build_%s.fa.gz" % setname))
    mp.write_bytes(util.fixture_bytes("build_%s.map.gz" % setname))
    return str(fa), str(mp)


@pytest.mark.parametrize("tag", sorted(BUILDS))
def test_build_golden(tag, tmp_path):
    v = BUILDS[tag]
    fa, mp = inputs(v["set"], tmp_path)
    ubt = str(tmp_path / "o.ubt")
    code, st = build(fa, mp, ubt, W=v["W"], I=v["I"], complevel=v["complevel"], gg=bool(v["gg"]))
    if v["exit"] == 0:
        assert code == lib.OK
        assert ctrfile.sha256_file(ubt) == v["ubt_sha256"]
        assert ctrfile.sha256_file(ubt + (".gg.log" if v["gg"] else ".log")) == v["log_sha256"]
        assert ("Total nodes in tree: %d [%d labels]" % (st.n_nodes, st.n_labels)) in v["stdout_tail"]
    else:
        assert code in (lib.E_BUILD, lib.E_IO) and EXIT_OF[st.error_kind] == v["exit"]
        assert not os.path.exists(ubt)                       # the reference exits before it opens the output


def random_refs(rng, n_refs, lo, hi, n_leaves):
    """related references with GG-style labels, some labels short (< 2 ';'), N's and lowercase sprinkled in"""
    ranks = "kpcofgst"
    leaves = []
    for _ in range(n_leaves):
        depth = int(rng.integers(1, 9))
        leaves.append(";".join("%s__%d" % (ranks[d], int(rng.integers(0, 3))) for d in range(depth)))
    roots = [rng.integers(0, 4, hi) for _ in range(5)]
    fa, mp = [], []
    for i in range(n_refs):
        L = int(rng.integers(lo, hi))
        s = roots[int(rng.integers(0, 5))][:L].copy()
        mut = rng.random(L) < 0.01
        s[mut] = rng.integers(0, 4, int(mut.sum()))
        b = np.frombuffer(b"ACGT", dtype=np.uint8)[s].copy()
        if rng.random() < 0.2:
            b[rng.integers(0, L, 3)] = ord("N")
        if rng.random() < 0.1:
            b = np.frombuffer(bytes(b).lower(), dtype=np.uint8)
        name = "ref %d x" % i
        fa.append(b">" + name.encode() + b"\n" + bytes(b) + b"\n")
        mp.append(name.encode() + b"\t" + leaves[int(rng.integers(0, n_leaves))].encode() + b"\n")
    order = rng.permutation(n_refs)
    return b"".join(fa), b"".join(mp[i] for i in order)


@pytest.mark.parametrize("W,I,cl,gg,seed", [(8, 2, 0, 1, 1), (8, 2, 1, 1, 2), (8, 2, 2, 0, 3), (16, 2, 0, 1, 4), (8, 4, 1, 1, 5),
                                            (16, 4, 3, 1, 6)])
def test_build_random_refs_vs_oracle(W, I, cl, gg, seed, tmp_path):
    """Heavily colliding references (5 root sequences, 1 % mutations): long collision chains, many labels created by cuts."""
    rng = np.random.default_rng(seed)
    fa_b, mp_b = random_refs(rng, 400, 40, 3000, 60)
    fa = tmp_path / "r.fa"; mp = tmp_path / "r.map"
    fa.write_bytes(fa_b); mp.write_bytes(mp_b)
    want = str(tmp_path / "w.ubt"); got = str(tmp_path / "g.ubt")
    code, ns, nn, nl, err = orc.build_file(str(fa), str(mp), want, W=W, I=I, complevel=cl, gg=bool(gg))
    assert code == 0, err
    gcode, st = build(str(fa), str(mp), got, W=W, I=I, complevel=cl, gg=bool(gg))
    assert gcode == lib.OK
    assert (st.n_seqs, st.n_nodes, st.n_labels) == (ns, nn, nl)
    assert open(got, "rb").read() == open(want, "rb").read()
    ext = ".gg.log" if gg else ".log"
    assert open(got + ext, "rb").read() == open(want + ext, "rb").read()


def test_build_then_compress_then_search_chain(tmp_path):
    """utree-buildGG -> xtree-compress -> xtree-searchGG, all on the GPU path, equals the reference-built toy DB chain:
    the `.ubt` of the `rel` set compresses to a `.ctr` the search accepts, and the label set survives."""
    v = BUILDS["rel_buildGG_c1"]
    fa, mp = inputs("rel", tmp_path)
    ubt = str(tmp_path / "o.ubt"); ctr = str(tmp_path / "o.ctr")
    code, st = build(fa, mp, ubt, complevel=1)
    assert code == lib.OK and ctrfile.sha256_file(ubt) == v["ubt_sha256"]
    ccode, cst = compress(ubt, ctr)
    assert ccode == lib.OK and cst.n_nodes == st.n_nodes
    d = ctrfile.read_ctr(ctr)
    assert d.n_nodes == st.n_nodes and len(d.labels()) == st.n_labels


@pytest.mark.parametrize("gg", [1, 0])
def test_build_cli_drop_in(gg, tmp_path):
    tag = "rel_buildGG_c2" if gg else "rel_build_c1"
    v = BUILDS[tag]
    fa, mp = inputs("rel", tmp_path)
    ubt = str(tmp_path / "o.ubt")
    cli = lib.BUILD_GG_CLI_PATH if gg else lib.BUILD_CLI_PATH
    r = subprocess.run([cli, fa, mp, ubt, "0", str(v["complevel"])], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    assert ctrfile.sha256_file(ubt) == v["ubt_sha256"]
    so = r.stdout.decode()
    assert v["stdout_tail"][1] in so and "Tree written." in so
    r = subprocess.run([cli], stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"usage: utree-build" in r.stdout
    efa, emp = inputs("err_missing", tmp_path)
    r = subprocess.run([cli, efa, emp, ubt + "2", "0", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 4 and b"taxon map incomplete" in r.stdout


@pytest.mark.parametrize("tag,limit", [("rel_buildGG_c0", 3000), ("rel_buildGG-k64_c1", 700), ("corner_buildGG_c0", 400),
                                       ("rel_build_c1", 1000), ("rel_buildGG-ix32_c2", 100000)])
def test_build_in_kmer_range_passes(tag, limit, tmp_path, monkeypatch):
    """Large inputs are built in passes over ranges of k-mers (each sort below 2^31 items, buffers sized to the free HBM);
    a hook forces many small passes here.  Label numbering must not notice: its clock is the position in the input."""
    monkeypatch.setenv("UTREE_BUILD_PASS_KMERS", str(limit))
    v = BUILDS[tag]
    fa, mp = inputs(v["set"], tmp_path)
    ubt = str(tmp_path / "o.ubt")
    code, st = build(fa, mp, ubt, W=v["W"], I=v["I"], complevel=v["complevel"], gg=bool(v["gg"]))
    assert code == lib.OK
    assert ctrfile.sha256_file(ubt) == v["ubt_sha256"]
    assert ctrfile.sha256_file(ubt + (".gg.log" if v["gg"] else ".log")) == v["log_sha256"]


@pytest.mark.parametrize("W,gg", [(8, 1), (8, 0), (16, 1)])
def test_build_hostile_labels_vs_oracle(W, gg, tmp_path):
    """Hostile label sets through the GPU BUILD (util.hostile_build_case; the oracle is held against the genuine reference on
    them in test_oracle_golden.py): `.ubt` and log identical at every compression level."""
    for seed in (1, 2):
        fa_b, mp_b = util.hostile_build_case(seed)
        fa = tmp_path / "i.fa"; mp = tmp_path / "i.map"
        fa.write_bytes(fa_b); mp.write_bytes(mp_b)
        for cl in (0, 1, 2):
            want, got = str(tmp_path / "w.ubt"), str(tmp_path / "g.ubt")
            code, ns, nn, nl, err = orc.build_file(str(fa), str(mp), want, W=W, I=2, complevel=cl, gg=bool(gg))
            assert code == 0, err
            gcode, st = build(str(fa), str(mp), got, W=W, I=2, complevel=cl, gg=bool(gg))
            assert gcode == lib.OK and (st.n_seqs, st.n_nodes, st.n_labels) == (ns, nn, nl)
            ext = ".gg.log" if gg else ".log"
            assert open(got, "rb").read() == open(want, "rb").read()
            assert open(got + ext, "rb").read() == open(want + ext, "rb").read()
