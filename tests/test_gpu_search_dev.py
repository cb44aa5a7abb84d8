"""GPU tests of the whole-file search with framing and formatting on the device (csrc/search_dev.c, text_kernels.hip),
through the C-ABI (utree_search_file): the reference's committed outputs, the oracle on seeded files, chunk boundaries
at every position class, two device handles (the multi-GPU sharding and ordered concatenation of search_dev.c on one
card), and the hand-over to the host framing for input the kernels do not take.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import orc
from utree_amd import lib
from utree_amd.search import CtrDB, DeviceTree, search_gg
import util


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


_TREES = {}


def tree_for(name):
    if name not in _TREES:
        while len(_TREES) >= 4:                         # (a handle that has searched a file keeps its lanes' buffers, ~14 GB of HBM: the oldest go)
            _TREES.pop(next(iter(_TREES)))[1].close()
        db = CtrDB.open(util.fixture_ctr(name))
        _TREES[name] = (db, DeviceTree.upload(db, 0))
    return _TREES[name]


def run(db, trees, data, tmp_path, rc=False, threads=4):
    fa, out = tmp_path / "in.fa", tmp_path / "out.txt"
    fa.write_bytes(data)
    if out.exists():
        out.unlink()
    code, st = search_gg(db, trees, str(fa), str(out), rc=rc, threads=threads)
    return code, st, out.read_bytes()


@pytest.mark.parametrize("name,rc", [("toy", 0), ("toy", 1), ("k64", 1), ("ix32", 0), ("k64ix32", 1), ("vote", 0), ("katq2", 0), ("generic", 0), ("k16", 1)])
def test_golden_files_take_the_device_pipeline(torch_cuda, name, rc, tmp_path):
    db, tree = tree_for(name)
    data = util.fixture_bytes(util.READS_OF.get(name, name) + "_reads.fa.gz")
    code, st, got = run(db, [tree], data, tmp_path, rc=bool(rc))
    assert code == lib.OK
    want = util.fixture_bytes("%s_out%s.txt.gz" % (name, "_rc" if rc else ""))
    assert got == want
    if b"\0" not in data:
        assert st.pipeline == 1 and st.n_lanes >= 1                  # framed and formatted on the GPU, not handed to the host path
    assert st.good_finds == want.count(b"\n") and st.n_reads == data.count(b"\n") // 2
    assert st.bytes_out == len(want) and st.bytes_in == len(data)


@pytest.mark.parametrize("chunk", [300, 1000, 4096, 65536 + 17])
def test_chunk_boundaries(torch_cuda, chunk, tmp_path, monkeypatch):
    """Small chunks put a boundary after almost every record class of the toy reads (N, lowercase, CRLF, short, no-hit);
    chunks go to the lanes in turn and the texts are concatenated in input order."""
    monkeypatch.setenv("UTREE_CHUNK_BYTES", str(chunk))
    db, tree = tree_for("toy")
    data = util.fixture_bytes("toy_reads.fa.gz")[: 400_000 if chunk < 4096 else None]
    data = data[: data.rfind(b"\n>") + 1]
    o = orc.OracleDB.load(util.fixture_ctr("toy"))
    fa, want = tmp_path / "w.fa", tmp_path / "want.txt"
    fa.write_bytes(data)
    ocode, nr, good, err = o.search_file(str(fa), str(want), threads=4, rc=True)
    code, st, got = run(db, [tree], data, tmp_path, rc=True)
    assert code == lib.OK and ocode == 0
    assert st.pipeline == 1
    assert got == want.read_bytes()
    assert st.n_reads == nr and st.good_finds == good


def test_two_device_handles_share_the_chunks(torch_cuda, tmp_path, monkeypatch):
    """n_dev = 2 on one card: the second handle adopts a byte copy of the image (what utree_dev_replicate's broadcast
    delivers), lanes of both handles take chunks in turn, the output is the one-device output (= the reference's)."""
    torch = torch_cuda
    monkeypatch.setenv("UTREE_CHUNK_BYTES", "20000")
    db, tree = tree_for("toy")
    ptr, used = tree.image_ptr()

    class _Raw:                                   # the library-owned image as a torch view (plumbing: a device-to-device copy)
        __cuda_array_interface__ = {"shape": (used,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    src = torch.as_tensor(_Raw(), device="cuda:0")
    copy = torch.empty(used + 4096, dtype=torch.uint8, device="cuda:0")[4096:]
    copy.copy_(src)
    torch.cuda.synchronize()
    t2 = DeviceTree.attach(db, copy, 0)
    for rc in (False, True):
        data = util.fixture_bytes("toy_reads.fa.gz")
        code, st, got = run(db, [tree, t2], data, tmp_path, rc=rc)
        assert code == lib.OK and st.pipeline == 1 and st.n_lanes >= 2
        assert got == util.fixture_bytes("toy_out%s.txt.gz" % ("_rc" if rc else ""))
    # the host pipeline (UTREE_HOST_TEXT) shards every framed chunk over the handles instead: same file
    monkeypatch.setenv("UTREE_HOST_TEXT", "1")
    code, st, got = run(db, [tree, t2], util.fixture_bytes("toy_reads.fa.gz"), tmp_path, rc=True)
    assert code == lib.OK and st.pipeline == 0
    assert got == util.fixture_bytes("toy_out_rc.txt.gz")
    t2.close()


def test_eight_device_handles_on_one_card(torch_cuda, tmp_path, monkeypatch):
    """The command line's shape on an 8-GPU node -- one process, eight handles, one input, one output -- rehearsed on one card: seven byte copies
    of the image attached beside the built one, three lanes per handle (the pipeline's choice beyond two devices): 24 lanes share the chunks,
    the output is the reference's, with one output file and with the output in parts."""
    torch = torch_cuda
    monkeypatch.setenv("UTREE_CHUNK_BYTES", "20000")
    db, tree = tree_for("toy")
    ptr, used = tree.image_ptr()

    class _Raw:
        __cuda_array_interface__ = {"shape": (used,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    src = torch.as_tensor(_Raw(), device="cuda:0")
    copies, trees = [], [tree]
    for _ in range(7):
        c = torch.empty(used + 4096, dtype=torch.uint8, device="cuda:0")[4096:]
        c.copy_(src)
        copies.append(c)
    torch.cuda.synchronize()
    trees += [DeviceTree.attach(db, c, 0) for c in copies]
    data = util.fixture_bytes("toy_reads.fa.gz")
    code, st, got = run(db, trees, data, tmp_path, rc=True, threads=16)
    assert code == lib.OK and st.pipeline == 1 and st.n_lanes == 24
    assert got == util.fixture_bytes("toy_out_rc.txt.gz")
    monkeypatch.setenv("UTREE_OUTPUT_PARTS", "8")
    fa, out = tmp_path / "in.fa", tmp_path / "parts.txt"
    fa.write_bytes(data)
    code, st = search_gg(db, trees, str(fa), str(out), rc=True, threads=16)
    assert code == lib.OK
    assert b"".join((tmp_path / ("parts.txt.part%03d" % i)).read_bytes() for i in range(8)) == util.fixture_bytes("toy_out_rc.txt.gz")
    for t in trees[1:]:
        t.close()


def test_last_line_without_newline_crlf_and_empty_file(torch_cuda, tmp_path):
    db, tree = tree_for("toy")
    o = orc.OracleDB.load(util.fixture_ctr("toy"))
    base = util.fixture_bytes("toy_reads.fa.gz")[:30_000]
    base = base[: base.rfind(b"\n>") + 1]
    cases = {
        "no_final_newline": base[:-1],
        "crlf": base.replace(b"\n", b"\r\n"),
        "crlf_no_final_newline": base.replace(b"\n", b"\r\n")[:-2],
        "empty": b"",
        "spaces_and_tabs_in_headers": base.replace(b">", b">x y\tz ", 50),
    }
    for nm, data in cases.items():
        fa, want = tmp_path / "w.fa", tmp_path / "want.txt"
        fa.write_bytes(data)
        ocode, nr, good, err = o.search_file(str(fa), str(want), threads=2, rc=False)
        code, st, got = run(db, [tree], data, tmp_path)
        assert code == lib.OK and ocode == 0, nm
        assert st.pipeline == 1, nm
        assert got == want.read_bytes(), nm
        assert st.n_reads == nr, nm


def test_input_for_the_host_framing_is_handed_over(torch_cuda, tmp_path, monkeypatch):
    """Malformed or unusual input leaves the device pipeline (stats.pipeline == 0) and gets the reference's behaviour from
    the host framing: NUL bytes, a header without '>', a sequence line starting with '>', a trailing header, a record
    larger than a chunk."""
    db, tree = tree_for("toy")
    o = orc.OracleDB.load(util.fixture_ctr("toy"))
    base = util.fixture_bytes("toy_reads.fa.gz")[:20_000]
    base = base[: base.rfind(b"\n>") + 1]
    mid = base.find(b"\n>", 10_000) + 1
    cases = {
        "nul_in_sequence": base[:mid + 30] + b"\0" + base[mid + 31:],
        "no_header": base[:mid] + b"r_without_gt\nACGT\n" + base[mid:],
        "sequence_begins_gt": base[:mid] + b">a\n>b\n" + base[mid:],
        "trailing_header": base + b">last\n",
        "blank_line": base[:mid] + b"\n" + base[mid:],
    }
    for nm, data in cases.items():
        fa, want = tmp_path / "w.fa", tmp_path / "want.txt"
        fa.write_bytes(data)
        ocode, nr, good, err = o.search_file(str(fa), str(want), threads=2, rc=True)
        code, st, got = run(db, [tree], data, tmp_path, rc=True)
        assert {lib.OK: 0, lib.E_FASTA: 2, lib.E_IO: 1}[code] == ocode, nm
        assert st.pipeline == 0, nm
        assert got == want.read_bytes(), nm
    # a record that does not fit a chunk
    monkeypatch.setenv("UTREE_CHUNK_BYTES", "200")
    long_rec = b">long\n" + b"ACGT" * 200 + b"\n"
    data = base[:mid] + long_rec + base[mid:]
    fa, want = tmp_path / "w.fa", tmp_path / "want.txt"
    fa.write_bytes(data)
    ocode, nr, good, err = o.search_file(str(fa), str(want), threads=2, rc=False)
    code, st, got = run(db, [tree], data, tmp_path)
    assert code == lib.OK and st.pipeline == 0 and got == want.read_bytes()


def test_context_is_reused_and_strand_mode_may_change(torch_cuda, tmp_path):
    db, tree = tree_for("toy")
    data = util.fixture_bytes("toy_reads.fa.gz")
    for rc in (False, True, False, True):
        code, st, got = run(db, [tree], data, tmp_path, rc=rc)
        assert code == lib.OK and st.pipeline == 1
        assert got == util.fixture_bytes("toy_out%s.txt.gz" % ("_rc" if rc else ""))
    assert lib.load().utree_search_prepare(db._h, (__import__("ctypes").c_void_p * 1)(tree._h), 1, 1) == lib.OK


def test_seeded_multichunk_file_vs_oracle(torch_cuda, tmp_path):
    """A 150 MB file (three 64 MiB chunks, four lanes) on a synthetic database: output equals the oracle's file."""
    torch = torch_cuda
    from utree_amd import synth
    sdb = synth.make_db(torch.device("cuda:0"), n_nodes=3_000_000, seed=synth.DB_SEED, keep_raw=True)
    n = 900_000
    reads = synth.make_reads(sdb, n_reads=n, read_len=150, seed=5)
    seq = reads.bases.cpu().numpy().reshape(n, 150)
    names = np.char.add(">read_", np.arange(n).astype("U9")).astype("S")
    data = b"".join(names[i] + b"\n" + seq[i].tobytes() + b"\n" for i in range(n))
    assert len(data) > 2 * (64 << 20)
    o = orc.OracleDB.from_memory(sdb.W, 2, sdb.binix.cpu().numpy().view(np.uint32).astype(np.uint64), sdb.records.cpu().numpy(), sdb.label_text)
    fa, want = tmp_path / "w.fa", tmp_path / "want.txt"
    fa.write_bytes(data)
    ocode, nr, good, err = o.search_file(str(fa), str(want), threads=16, rc=False)
    code, st, got = run(sdb.ctr, [sdb.tree], data, tmp_path, threads=8)
    assert code == lib.OK and ocode == 0 and st.pipeline == 1
    assert st.n_reads == n == nr and st.good_finds == good
    assert got == want.read_bytes()


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs: utree_dev_replicate issues one ncclBroadcast over all visible devices")
def test_replicate_rccl_over_all_visible_devices(torch_cuda, tmp_path):
    """The C host path of the multi-GPU search (north_star: host orchestration in C): utree_dev_replicate (ncclCommInitAll + one
    ncclBroadcast of the flat image, csrc/rccl_replicate.c), then utree_search_file over all the handles -- lanes of every device
    take chunks in turn, the output is the reference's -- and each replica classifies a batch exactly like the original."""
    import ctypes as C
    torch = torch_cuda
    n = torch.cuda.device_count()
    db = CtrDB.open(util.fixture_ctr("toy"))
    t0 = DeviceTree.upload(db, 0)
    devs = (C.c_int * n)(*range(n))
    out = (C.c_void_p * n)()
    assert lib.load().utree_dev_replicate(db._h, t0._h, devs, n, out) == lib.OK
    trees = [t0] + [DeviceTree(out[i], db) for i in range(1, n)]
    data = util.fixture_bytes("toy_reads.fa.gz")
    for rc in (False, True):
        code, st, got = run(db, trees, data, tmp_path, rc=rc, threads=8)
        assert code == lib.OK and st.pipeline == 1 and st.n_lanes >= n
        assert got == util.fixture_bytes("toy_out%s.txt.gz" % ("_rc" if rc else ""))
    from utree_amd.search import frame_fasta
    fr = frame_fasta(data)
    buf = np.frombuffer(data, dtype=np.uint8)
    want = None
    for g, t in enumerate(trees):
        dev = "cuda:%d" % g
        res = t.classify(torch.from_numpy(buf.copy()).to(dev), torch.from_numpy(fr["seq_off"].astype(np.int64)).to(dev),
                         torch.from_numpy(fr["seq_len"].astype(np.int32)).to(dev), rc=True).cpu()
        want = res if want is None else want
        assert torch.equal(res, want), "replica on device %d differs" % g


def test_a_lane_without_buffers_steps_aside(torch_cuda, tmp_path, monkeypatch):
    """A lane of the file pipeline whose buffers cannot be allocated (pinned host memory, HBM) leaves the chunks to the other lanes; the
    search fails only when no lane is left.  UTREE_TEST_LANE_NOMEM=1: every lane but the first gets UTREE_E_NOMEM instead of its buffers."""
    db = CtrDB.open(util.fixture_ctr("toy"))
    tree = DeviceTree.upload(db, 0)                                 # a fresh handle: no lane has its buffers yet
    data = util.fixture_bytes("toy_reads.fa.gz")
    monkeypatch.setenv("UTREE_CHUNK_BYTES", "20000")
    monkeypatch.setenv("UTREE_TEST_LANE_NOMEM", "1")
    code, st, got = run(db, [tree], data, tmp_path, rc=True)
    assert code == lib.OK and st.pipeline == 1 and st.n_lanes >= 2
    assert got == util.fixture_bytes("toy_out_rc.txt.gz")
    monkeypatch.delenv("UTREE_TEST_LANE_NOMEM")
    code, st, got = run(db, [tree], data, tmp_path, rc=True)         # the lanes that stepped aside get their buffers now
    assert code == lib.OK and got == util.fixture_bytes("toy_out_rc.txt.gz")
    tree.close()


def test_without_pinned_host_memory_the_search_still_runs(torch_cuda, tmp_path, monkeypatch):
    """Pinned host memory is a resource the users of a machine share: when hipHostMalloc fails a lane takes ordinary memory for its chunk and
    text buffers (the copies are then staged by the runtime).  UTREE_TEST_NO_PINNED=1 takes that branch."""
    db = CtrDB.open(util.fixture_ctr("toy"))
    tree = DeviceTree.upload(db, 0)
    data = util.fixture_bytes("toy_reads.fa.gz")
    monkeypatch.setenv("UTREE_CHUNK_BYTES", "20000")
    monkeypatch.setenv("UTREE_TEST_NO_PINNED", "1")
    code, st, got = run(db, [tree], data, tmp_path, rc=True)
    assert code == lib.OK and st.pipeline == 1
    assert got == util.fixture_bytes("toy_out_rc.txt.gz")
    tree.close()


def test_rccl_on_a_communicator_of_one_rank(torch_cuda, tmp_path, monkeypatch):
    """UTREE_RCCL_FORCE=1 (csrc/rccl_replicate.c): with one device / one rank the early returns are skipped, so ncclCommInitAll,
    ncclCommInitRank, the 8-byte size broadcast and the image's ncclBroadcast pieces all execute on this lease's single card --
    the image arrives in a second allocation, is attached like any received copy, and must answer like the one that was built:
    the reference's golden lines through the whole-file search, and a batch's records."""
    torch = torch_cuda
    monkeypatch.setenv("UTREE_RCCL_FORCE", "1")
    db = CtrDB.open(util.fixture_ctr("toy"))
    t0 = DeviceTree.upload(db, 0)
    # one process, all devices (the command line's shape): utree_dev_replicate
    trees = DeviceTree.replicate(db, t0, [0])
    assert len(trees) == 1 and trees[0] is not t0 and trees[0].image_ptr()[0] != t0.image_ptr()[0]
    assert trees[0].image_ptr()[1] == t0.image_ptr()[1] and DeviceTree.replicate_seconds() > 0
    # one process per GPU (bench.py's shape): utree_rccl_unique_id + utree_dev_replicate_rank
    t_rank = DeviceTree.replicate_rank(db, t0, 0, 0, 1, 0, DeviceTree.rccl_unique_id())
    assert t_rank is not t0 and t_rank.image_ptr()[0] not in (t0.image_ptr()[0], trees[0].image_ptr()[0])
    data = util.fixture_bytes("toy_reads.fa.gz")
    for t in (trees[0], t_rank):
        for rc in (False, True):
            code, st, got = run(db, [t], data, tmp_path, rc=rc)
            assert code == lib.OK and st.pipeline == 1
            assert got == util.fixture_bytes("toy_out%s.txt.gz" % ("_rc" if rc else ""))
    from utree_amd.search import frame_fasta
    fr = frame_fasta(data)
    buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    off, ln = torch.from_numpy(fr["seq_off"].astype(np.int64)).cuda(), torch.from_numpy(fr["seq_len"].astype(np.int32)).cuda()
    want = t0.classify(buf, off, ln, rc=True).cpu()
    for t in (trees[0], t_rank):
        assert torch.equal(t.classify(buf, off, ln, rc=True).cpu(), want)
        t.close()
    # both handles of a forced two-"device" search on one card: the built image and its broadcast copy share the chunks
    monkeypatch.setenv("UTREE_CHUNK_BYTES", "20000")
    t2 = DeviceTree.replicate(db, t0, [0])[0]
    code, st, got = run(db, [t0, t2], data, tmp_path, rc=True)
    assert code == lib.OK and st.n_lanes >= 2 and got == util.fixture_bytes("toy_out_rc.txt.gz")
    t2.close()
    t0.close()


def test_rccl_broadcast_in_pieces_of_a_gibibyte(torch_cuda, tmp_path, monkeypatch):
    """An image beyond 1 GiB goes through ncclBroadcast in several pieces (the last one ragged); the copy is the image, byte for
    byte, and classifies like it."""
    torch = torch_cuda
    from utree_amd import synth
    monkeypatch.setenv("UTREE_RCCL_FORCE", "1")
    sdb = synth.make_db(torch.device("cuda:0"), 120_000_000, W=8)
    ptr, used = sdb.tree.image_ptr()
    assert used > (2 << 30) + 4096, used
    t2 = DeviceTree.replicate(sdb.ctr, sdb.tree, [0])[0]
    p2, u2 = t2.image_ptr()
    assert u2 == used and p2 != ptr

    def view(p, n):
        class _Raw:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (p, False), "version": 2}
        return torch.as_tensor(_Raw(), device="cuda:0")
    a, b = view(ptr, used), view(p2, used)
    for lo in range(0, used, 1 << 28):
        assert torch.equal(a[lo:lo + (1 << 28)], b[lo:lo + (1 << 28)]), "piece at %d differs" % lo
    reads = synth.make_reads(sdb, 100_000, 150, seed=synth.READ_SEED + 5)
    want = sdb.tree.classify(reads.bases, reads.off, reads.length, rc=True).cpu()
    assert torch.equal(t2.classify(reads.bases, reads.off, reads.length, rc=True).cpu(), want)
    assert int((want[:, 2] > 0).sum()) > 90_000
    print("broadcast of %.2f GiB on a one-rank communicator: %.4f s" % (used / 2**30, DeviceTree.replicate_seconds()))
    t2.close()
    sdb.tree.close()


def test_cli_falls_back_to_uploads_when_the_broadcast_fails(torch_cuda, tmp_path):
    """The command line with a broadcast that fails (UTREE_TEST_REPLICATE_FAIL under the one-rank rehearsal): a warning, every
    GPU reads the tree from the host instead (SURVEY 8(e)'s fallback), same output, exit code 0 -- and without the failure the
    same invocation reports the broadcast."""
    import subprocess
    fa, out = tmp_path / "r.fa", tmp_path / "o.txt"
    fa.write_bytes(util.fixture_bytes("toy_reads.fa.gz"))
    for fail in (0, 1):
        env = dict(os.environ, UTREE_RCCL_FORCE="1", UTREE_GPUS="1")
        if fail:
            env["UTREE_TEST_REPLICATE_FAIL"] = "1"
        p = subprocess.run([lib.CLI_PATH, util.fixture_ctr("toy"), str(fa), str(out), "4", "RC"], env=env, capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        assert out.read_bytes() == util.fixture_bytes("toy_out_rc.txt.gz")
        err = p.stderr.decode()
        assert ("RCCL broadcast of the tree failed" in err) == bool(fail), err
        assert ("replicated to 1 GPU(s) by RCCL broadcast" in err) == (not fail), err


def test_output_that_cannot_seek_is_written_in_order(torch_cuda, tmp_path, monkeypatch):
    """`xtree-searchGG db reads.fa /dev/stdout | gzip`, a FIFO, process substitution: the reference writes with fprintf and works
    there; the device pipeline places chunks with pwrite, so on such an output it writes them in chunk order instead."""
    import threading
    monkeypatch.setenv("UTREE_CHUNK_BYTES", "20000")
    db, tree = tree_for("toy")
    data = util.fixture_bytes("toy_reads.fa.gz")
    fa, fifo = tmp_path / "in.fa", tmp_path / "out.fifo"
    fa.write_bytes(data)
    os.mkfifo(fifo)
    got = {}

    def reader():
        with open(fifo, "rb") as f:
            got["bytes"] = f.read()
    th = threading.Thread(target=reader)
    th.start()
    code, st = search_gg(db, [tree], str(fa), str(fifo), rc=True, threads=4)
    th.join(60)
    assert code == lib.OK and st.pipeline == 1 and st.n_lanes >= 2
    assert got["bytes"] == util.fixture_bytes("toy_out_rc.txt.gz")


@pytest.mark.parametrize("where", [0.02, 0.55, 0.97])
def test_fifo_output_and_a_record_the_device_pipeline_hands_over(torch_cuda, where, tmp_path, monkeypatch):
    """Output that cannot seek AND a late chunk the device pipeline does not take (a NUL byte in a sequence line: fgets / strlen
    semantics, the host framing's business): the chunks in front of it are already in the pipe, so the host pipeline continues from
    that chunk on the same descriptor -- the reader sees every line once, in order: the oracle's file.  (Starting over, as for a
    regular file, would send the first chunks twice: O_TRUNC does nothing to a pipe.)"""
    import threading
    monkeypatch.setenv("UTREE_CHUNK_BYTES", "20000")
    db, tree = tree_for("toy")
    o = orc.OracleDB.load(util.fixture_ctr("toy"))
    data = bytearray(util.fixture_bytes("toy_reads.fa.gz")[:600_000])
    data = data[: data.rfind(b"\n>") + 1]
    at = data.index(b"\n", data.index(b"\n>", int(where * len(data))) + 1) + 30          # inside a sequence line
    assert data[at] in b"ACGTacgtN"
    data[at] = 0
    fa, fifo, want = tmp_path / "in.fa", tmp_path / "out.fifo", tmp_path / "want.txt"
    fa.write_bytes(bytes(data))
    ocode, nr, good, err = o.search_file(str(fa), str(want), threads=4, rc=True)
    assert ocode == 0
    os.mkfifo(fifo)
    got = {}

    def reader():
        with open(fifo, "rb") as f:
            got["bytes"] = f.read()
    th = threading.Thread(target=reader)
    th.start()
    code, st = search_gg(db, [tree], str(fa), str(fifo), rc=True, threads=4)
    th.join(60)
    assert code == lib.OK and st.pipeline == 0                      # the host pipeline finished the file ...
    assert got["bytes"] == want.read_bytes()                        # ... and nothing came twice
    assert st.n_reads == nr and st.good_finds == good
    # the same file into a regular output: started over by the host pipeline, same bytes
    code, st, out = run(db, [tree], bytes(data), tmp_path, rc=True)
    assert code == lib.OK and out == want.read_bytes() and st.n_reads == nr


def test_three_byte_records_fill_a_chunk(torch_cuda, tmp_path):
    """A record can be ">\\n\\n" (the reference reads it as a sequence of length 0 and prints nothing): a chunk of them holds
    more reads than bytes / 4.  Every read must be framed, and the real reads among them classified."""
    db, tree = tree_for("toy")
    o = orc.OracleDB.load(util.fixture_ctr("toy"))
    real = util.fixture_bytes("toy_reads.fa.gz")[:60_000]
    real = real[: real.rfind(b"\n>") + 1]
    data = b">\n\n" * 150_000 + real + b">\n\n" * 70_001 + real + b">x\n\n" * 10
    fa, want = tmp_path / "w.fa", tmp_path / "want.txt"
    fa.write_bytes(data)
    ocode, nr, good, err = o.search_file(str(fa), str(want), threads=4, rc=False)
    code, st, got = run(db, [tree], data, tmp_path)
    assert code == lib.OK and ocode == 0
    assert got == want.read_bytes() and len(got) > 0
    assert st.n_reads == nr == data.count(b"\n") // 2 and st.good_finds == good


@pytest.mark.parametrize("parts", [3, 7])
def test_output_in_parts_concatenates_to_the_one_file_output(torch_cuda, parts, tmp_path, monkeypatch):
    """UTREE_OUTPUT_PARTS=P: part p holds the lines of the chunks that start in the p-th P-th of the input, the parts are filled side by
    side (writers of ONE file serialise on its inode: tools/hostio_probe3.c), and their concatenation is byte for byte the file the
    reference writes -- also when the host pipeline takes the search over (everything in part 000) and with two device handles."""
    monkeypatch.setenv("UTREE_CHUNK_BYTES", "20000")
    monkeypatch.setenv("UTREE_OUTPUT_PARTS", str(parts))
    db, tree = tree_for("toy")
    data = util.fixture_bytes("toy_reads.fa.gz")
    want = util.fixture_bytes("toy_out_rc.txt.gz")
    fa, out = tmp_path / "in.fa", tmp_path / "out.txt"
    fa.write_bytes(data)

    def cat():
        names = sorted(p.name for p in tmp_path.iterdir() if p.name.startswith("out.txt.part"))
        assert names == ["out.txt.part%03d" % i for i in range(parts)] and not out.exists()
        return b"".join((tmp_path / n).read_bytes() for n in names), [(tmp_path / n).stat().st_size for n in names]
    code, st = search_gg(db, [tree], str(fa), str(out), rc=True, threads=4)
    got, sizes = cat()
    assert code == lib.OK and st.pipeline == 1 and got == want and st.bytes_out == len(want)
    assert min(sizes) > 0.5 * len(want) / parts                                   # (the parts really share the output)
    # the host pipeline (a NUL byte in a late chunk hands the search over): all of the output in part 000
    o = orc.OracleDB.load(util.fixture_ctr("toy"))
    bad = bytearray(data[:300_000]); bad = bad[: bad.rfind(b"\n>") + 1]
    at = bad.index(b"\n", bad.index(b"\n>", 200_000) + 1) + 20
    bad[at] = 0
    fa.write_bytes(bytes(bad))
    wantf = tmp_path / "want.txt"
    ocode, nr, good, err = o.search_file(str(fa), str(wantf), threads=4, rc=True)
    code, st = search_gg(db, [tree], str(fa), str(out), rc=True, threads=4)
    got, sizes = cat()
    assert code == lib.OK and st.pipeline == 0 and got == wantf.read_bytes() and sizes[1:] == [0] * (parts - 1)
