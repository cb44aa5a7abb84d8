"""`.ubt` -> `.ctr` on the GPU (utree_compress_file / xtree-compress) must write byte-identical
files to the reference's xtree-compress (itree.c:1234-1315).  Expected SHA-256s come from the genuine reference at
golden time (tests/golden/make_golden.py)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from utree_amd import ctrfile, lib
from utree_amd.search import compress
import util


def _ubt_from_db_fixture(name, path):
    d = util.load_db_fixture(name)
    hi, lo = d.words()
    ctrfile.write_ubt(path, d.W, d.I, hi, lo, d.ix(), d.label_text)
    return d


@pytest.mark.parametrize("name", ["toy", "k64", "ix32", "k64ix32", "k16"])
def test_compress_reference_built_databases(name, tmp_path):
    """The toy databases went through the reference's BUILD_GG + COMPRESS; re-compressing their `.ubt` must give the
    same `.ctr` bytes."""
    ubt, ctr = str(tmp_path / "a.ubt"), str(tmp_path / "a.ctr")
    d = _ubt_from_db_fixture(name, ubt)
    code, st = compress(ubt, ctr)
    assert code == lib.OK and st.n_nodes == d.n_nodes and (st.W, st.I) == (d.W, d.I)
    assert ctrfile.sha256_file(ctr) == util.manifest()[name + "_ctr_sha256"]


@pytest.mark.parametrize("name", ["cq_single_first", "cq_multi_first", "cq_dup_labels", "cq_unsorted"])
def test_compress_corner_cases_match_reference(name, tmp_path):
    z = np.load(os.path.join(util.GOLD, name + "_ubt.npz"))
    ubt, ctr = str(tmp_path / "a.ubt"), str(tmp_path / "a.ctr")
    ctrfile.write_ubt(ubt, 8, 2, np.zeros_like(z["lo"]), z["lo"], z["ix"], z["tail"].tobytes())
    r = subprocess.run([lib.COMPRESS_CLI_PATH, ubt, ctr], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    m = util.manifest()
    assert ctrfile.sha256_file(ctr) == m[name + "_ctr_sha256"]
    assert m[name + "_stdout_tail"] in r.stdout.decode()              # "Total nodes in tree: N [L labels]"


def test_compress_large_random_matches_numpy_model(tmp_path):
    """2 M nodes over many chunks: against the numpy model of COMPRESS (ctrfile.binix_like_compress), which
    tests/test_oracle_golden.py pins to the reference's table on the reference-built databases."""
    rng = np.random.default_rng(3)
    lo = np.unique(rng.integers(0, 1 << 63, size=6_000_000, dtype=np.uint64) << np.uint64(1))
    ix = rng.integers(0, 50, size=len(lo)).astype(np.uint32)
    labels = ["k__R;p__%d" % i for i in range(50)]
    cnt = np.bincount(ix, minlength=50)
    tail = b"".join(("%s\t%d\n" % (l, c)).encode() for l, c in zip(labels, cnt))
    ubt, ctr, want = str(tmp_path / "r.ubt"), str(tmp_path / "r.ctr"), str(tmp_path / "w.ctr")
    ctrfile.write_ubt(ubt, 8, 2, np.zeros_like(lo), lo, ix, tail)
    code, st = compress(ubt, ctr)
    assert code == lib.OK and st.n_nodes == len(lo) and st.label_count_total == len(lo)
    ctrfile.write_ctr(want, 8, 2, np.zeros_like(lo), lo, ix, labels, label_counts=cnt, like_compress=True)
    assert ctrfile.sha256_file(ctr) == ctrfile.sha256_file(want)


def test_compress_errors(tmp_path):
    code, _ = compress(str(tmp_path / "missing.ubt"), str(tmp_path / "o.ctr"))
    assert code == lib.E_IO
    p = tmp_path / "bad.ubt"
    p.write_bytes(np.array([8, 0, 2, 0], dtype="<u8").tobytes())
    assert compress(str(p), str(tmp_path / "o.ctr"))[0] == lib.E_FORMAT
    p.write_bytes(np.array([2, 0, 2, 3], dtype="<u8").tobytes() + b"\0" * 64)            # PACKSIZE=8: no build of the reference reads it either
    assert compress(str(p), str(tmp_path / "o.ctr"))[0] == lib.E_UNSUPPORTED
