"""GPU parity of the rank-specific search (`xtree-search`, itree.c -D SEARCH), called through
the C-ABI, against (a) the committed outputs of seven builds of the genuine reference and (b) the CPU oracle.
Bit-exact, including the reference's order dependence (the hit-array entry a read's vote picks up from an earlier
read, itree.c:982) and the words its register holds after a hit (itree.c:920, 950).

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import orc
from utree_amd import lib
from utree_amd.search import CtrDB, DeviceTree, frame_fasta, search_rank
import util

RANK = util.manifest().get("rank_outputs", {})


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


_TREES = {}


def tree_for(name, fine_bits=lib.FINE_AUTO):
    key = (name, fine_bits)
    if key not in _TREES:
        db = CtrDB.open(util.fixture_ctr(name))
        _TREES[key] = (db, DeviceTree.upload(db, 0, fine_bits))
    return _TREES[key]


@pytest.mark.parametrize("tag", sorted(RANK))
def test_rank_golden_file_to_file(torch_cuda, tag, tmp_path):
    """utree_rank_search_file == the genuine xtree-search builds, byte for byte (17 outputs)."""
    v = RANK[tag]
    db, tree = tree_for(v["db"])
    out = tmp_path / "o.txt"
    sl, sp, tol = v["params"]
    code, st = search_rank(db, tree, util.fixture_reads_path(v["reads"]), str(out), rc=bool(v["rc"]), slack=sl, sparsity=sp,
                           tolerance=tol, threads=4)
    want = util.fixture_bytes(tag + ".txt.gz")
    assert code == lib.OK
    assert out.read_bytes() == want
    assert st.good_finds == want.count(b"\n") == v["lines"]


def run_batches(torch, db, tree, data, batch, rc, **prm):
    """The batch operator, `batch` reads at a time in file order; returns (formatted bytes, records)."""
    fr = frame_fasta(data)
    buf = np.frombuffer(data, dtype=np.uint8)
    d_buf = torch.from_numpy(buf.copy()).cuda()
    n = len(fr["seq_off"])
    tree.rank_reset()
    recs = []
    for a in range(0, n, batch):
        b = min(n, a + batch)
        off = torch.from_numpy(fr["seq_off"][a:b].astype(np.int64)).cuda()
        ln = torch.from_numpy(fr["seq_len"][a:b].astype(np.int32)).cuda()
        recs.append(tree.rank_search(d_buf, off, ln, rc=rc, **prm).cpu().numpy())
    res = np.concatenate(recs) if recs else np.zeros((0, 6), np.int32)
    return db.format(buf, fr["name_off"], fr["name_len"], res, rank=True), res


@pytest.mark.parametrize("batch", [1, 7, 64, 65, 1000, 4097])
def test_rank_state_is_carried_across_batches(torch_cuda, batch):
    """Any batching of the same file gives the same bytes: what a read picks up from an earlier read's hit list crosses
    batch boundaries through the device-side array (incl. batches of one read, and reads of 120 kb)."""
    db, tree = tree_for("rk")
    data = util.fixture_bytes("rk_reads.fa.gz")
    if batch == 1:                                        # keep the one-read-per-batch case short
        fr = frame_fasta(data)
        cut = int(fr["seq_off"][400])
        data = data[: data.rfind(b">", 0, cut)]
        o = orc.OracleDB.load(util.fixture_ctr("rk"))
        rs = orc.RankSearch(o)
        fr = frame_fasta(data)
        want = b""
        for i in range(len(fr["seq_off"])):
            s, l = int(fr["seq_off"][i]), int(fr["seq_len"][i])
            no, nl = int(fr["name_off"][i]), int(fr["name_len"][i])
            want += rs.format(data[no:no + nl], rs.read(data[s:s + l]))
    else:
        want = util.fixture_bytes("rk_rank.txt.gz")
    got, _ = run_batches(torch_cuda, db, tree, data, batch, False)
    assert got == want


@pytest.mark.parametrize("name,rc,prm", [("rk", True, dict(slack=2, sparsity=2, tolerance=2)),
                                         ("toy", False, dict(slack=3, sparsity=8, tolerance=2)),
                                         ("k64", True, dict()), ("ix32", False, dict())])
def test_rank_records_match_oracle(torch_cuda, name, rc, prm):
    """Field by field (hits kept, most, secondMost, printed, label) against the oracle run read by read."""
    db, tree = tree_for(name, 2)
    data = util.fixture_bytes(name + "_reads.fa.gz")
    _, res = run_batches(torch_cuda, db, tree, data, 3000, rc, **prm)
    o = orc.OracleDB.load(util.fixture_ctr(name))
    rs = orc.RankSearch(o, **prm)
    fr = frame_fasta(data)
    for i in range(len(fr["seq_off"])):
        s, l = int(fr["seq_off"][i]), int(fr["seq_len"][i])
        w = rs.read(data[s:s + l], rc=rc)
        g = res[i]
        assert int(g[2]) == w.found, i
        if w.found:
            assert (int(g[4]), int(g[5]), int(g[1]) == -2) == (w.most, w.second, bool(w.printed)), i
            if w.printed:
                assert int(g[0]) == w.label, i


def test_rank_random_reads_vs_oracle(torch_cuda, tmp_path):
    """Seeded reads of 1 bp .. 200 kb from the dense DB in random order, both strands, vs the oracle's whole-file run."""
    d = util.load_db_fixture("rk")
    hi, lo = d.suffixes()
    b = d.binix
    prefix = np.searchsorted(b, np.arange(d.n_nodes), side="right") - 1
    words = (prefix.astype(np.uint64) << np.uint64(40)) | lo
    rng = np.random.default_rng(99)
    from utree_amd import ctrfile
    kmers = [ctrfile.decode_kmer(0, int(w), 32) for w in words[rng.integers(0, len(words), 4000)]]
    reads = []
    for i in range(2500):
        L = int(rng.choice([int(rng.integers(1, 80)), int(rng.integers(80, 400)), int(rng.integers(400, 3000))], p=[0.2, 0.7, 0.1]))
        if i in (700, 1900):
            L = 200000 if i == 700 else 70000
        parts, n = [], 0
        while n < L:
            if rng.random() < 0.6:
                s = kmers[int(rng.integers(0, len(kmers)))]
            else:
                s = "".join("ACGT"[c] for c in rng.integers(0, 4, int(rng.integers(1, 50))))
            if rng.random() < 0.05:
                s += "N"
            parts.append(s); n += len(s)
        reads.append((">r%d" % i, "".join(parts)[:L]))
    data = "".join("%s\n%s\n" % r for r in reads).encode()
    fa = tmp_path / "r.fa"
    fa.write_bytes(data)
    o = orc.OracleDB.load(util.fixture_ctr("rk"))
    db, tree = tree_for("rk", 1)
    for rc in (False, True):
        want = tmp_path / "w.txt"
        code, nr, good, err = orc.rank_search_file(o, str(fa), str(want), rc=rc)
        assert code == 0 and nr == len(reads)
        out = tmp_path / "g.txt"
        code, st = search_rank(db, tree, str(fa), str(out), rc=rc, threads=4)
        assert code == lib.OK and st.n_reads == len(reads)
        assert out.read_bytes() == want.read_bytes()
        got, _ = run_batches(torch_cuda, db, tree, data, 333, rc)
        assert got == want.read_bytes()


@pytest.mark.parametrize("name,rc", [("toy", 0), ("rk", 1)])
def test_rank_cli_drop_in(torch_cuda, name, rc, tmp_path):
    """The `xtree-search` command line: same arguments, output file and banners as the reference's binary."""
    out = tmp_path / "cls.txt"
    cmd = [lib.RANK_CLI_PATH, util.fixture_ctr(name), util.fixture_reads_path(name), str(out), "4"] + (["RC"] if rc else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    want = util.fixture_bytes("%s_rank%s.txt.gz" % (name, "_rc" if rc else ""))
    assert out.read_bytes() == want
    so = r.stdout.decode()
    assert "This is UTree [v2.0RF SigNature Edition]" in so and "Tree read." in so
    assert ("Good finds: %d" % want.count(b"\n")) in so
    # the compile-time knobs of the reference arrive through the environment
    out2 = tmp_path / "cls2.txt"
    env = dict(os.environ, UTREE_SLACK="3", UTREE_SPARSITY="8")
    r = subprocess.run(cmd[:3] + [str(out2)] + cmd[4:5] + (["RC"] if rc else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, env=env)
    assert r.returncode == 0
    if name == "toy" and not rc:
        assert out2.read_bytes() == util.fixture_bytes("toy_rank-s3p8.txt.gz")


def test_rank_cli_usage():
    r = subprocess.run([lib.RANK_CLI_PATH], stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"usage: xtree-search compTree.ctr fastaToSearch.fa output.txt [threads] [SPEED <X>] [RC]" in r.stdout


def test_rank_bad_parameters_are_refused(torch_cuda):
    import torch
    db, tree = tree_for("toy")
    z = torch.zeros(64, dtype=torch.uint8, device="cuda")
    off = torch.zeros(1, dtype=torch.int64, device="cuda")
    ln = torch.full((1,), 40, dtype=torch.int32, device="cuda")
    for bad in (dict(sparsity=0), dict(sparsity=33)):          # PACKSIZE/SPARSITY must be >= 1 window
        with pytest.raises(lib.UtreeError):
            tree.rank_search(z, off, ln, **bad)


def test_rank_search_on_adversarial_label_set(torch_cuda, tmp_path):
    """The hostile label set of the GG vote test through the rank-specific search: GPU file == oracle file (the oracle is
    held against the genuine reference on this case in test_oracle_golden.py)."""
    ctr_path, data, n_reads = util.adversarial_vote_case(2, str(tmp_path))
    fa = tmp_path / "r.fa"
    fa.write_bytes(data)
    want = tmp_path / "w.txt"
    code, nr, good, err = orc.rank_search_file(orc.OracleDB.load(ctr_path), str(fa), str(want))
    assert code == 0 and nr == n_reads
    db = CtrDB.open(ctr_path)
    tree = DeviceTree.upload(db, 0)
    out = tmp_path / "g.txt"
    code, st = search_rank(db, tree, str(fa), str(out), threads=4)
    assert code == lib.OK and st.n_reads == n_reads
    assert out.read_bytes() == want.read_bytes()
