"""Host-side mirror of the reference's seams for the SEARCH_GG path, over the C-ABI.

    CtrDB.open(path)                 ~ UTree *XT_read32(char *db, ';')                   itree.c:733
    DeviceTree.upload(db, device)    ~ the UTree's Dump/BinIx made resident in HBM       itree.c:140-141
    tree.get_ix(words)               ~ IXTYPE XT_getIX32(UTree*, WTYPE word)             itree.c:720
    tree.classify(bases, off, len)   ~ the per-read body of XT_doSearch32, GG branch     itree.c:891-1088
    search_gg(db, trees, in, out)    ~ size_t XT_doSearch32(utree, in, out, 8, 0, doRC)  itree.c:833

torch is used only for device memory and streams (plumbing); every computation happens in the HIP kernels
behind libutree_amd.so.  Nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import lib as _lib

RESULT_FIELDS = ("label", "cut", "found", "uix", "sl", "ol")
RESULT_DTYPE = np.dtype([("label", "<u4"), ("cut", "<i4"), ("found", "<u4"), ("uix", "<u4"), ("sl", "<u4"),
                         ("ol", "<u4")])


class CtrDB:
    """Host side of a `.ctr` database: header, bin table, labels (XT_read32, itree.c:733-828)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        info = _lib.CtrInfo()
        _lib.check(_lib.load().utree_ctr_get_info(self._h, C.byref(info)), "utree_ctr_get_info")
        self.info = info

    @classmethod
    def open(cls, path: str) -> "CtrDB":
        h = C.c_void_p()
        _lib.check(_lib.load().utree_ctr_open(path.encode(), C.byref(h)), "utree_ctr_open(%s)" % path)
        return cls(h.value)

    @classmethod
    def from_memory(cls, W: int, I: int, n_nodes: int, binix: np.ndarray, records: Optional[np.ndarray],
                    label_text: bytes) -> "CtrDB":
        width = 4 if n_nodes < 0xFFFFFFFF else 8
        b = np.ascontiguousarray(binix.astype("<u4" if width == 4 else "<u8"))
        rec_ptr = None
        if records is not None:
            records = np.ascontiguousarray(records, dtype=np.uint8)
            rec_ptr = records.ctypes.data
        h = C.c_void_p()
        _lib.check(_lib.load().utree_ctr_from_memory(W, I, n_nodes, b.ctypes.data, width, rec_ptr, label_text,
                                                     len(label_text), C.byref(h)), "utree_ctr_from_memory")
        return cls(h.value)

    W = property(lambda s: s.info.W)
    I = property(lambda s: s.info.I)
    k = property(lambda s: s.info.k)
    n_nodes = property(lambda s: s.info.n_nodes)
    n_labels = property(lambda s: s.info.n_labels)

    def label(self, ix: int) -> Optional[bytes]:
        n = C.c_uint32()
        p = _lib.load().utree_ctr_label(self._h, ix, C.byref(n))
        return None if not p else C.string_at(p, n.value)

    def format(self, buf: np.ndarray, name_off: np.ndarray, name_len: np.ndarray, results: np.ndarray,
               rank: bool = False) -> bytes:
        """Output lines of itree.c:1032/1040/1096 for framed reads and their results (rank=True: the
        rank-specific search's lines, itree.c:1002)."""
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        name_len = np.ascontiguousarray(name_len, dtype=np.uint32)
        results = np.ascontiguousarray(results)
        assert results.dtype.itemsize == 24 or results.dtype == np.int32
        n = len(name_off)
        cap = int(name_len.sum()) + n * 256 + 4096
        bad = C.c_size_t(-1).value
        while True:
            out = np.empty(cap, dtype=np.uint8)
            good = C.c_uint64(0)
            fn = _lib.load().utree_format_rank_records if rank else _lib.load().utree_format_records
            L = fn(self._h, buf.ctypes.data, name_off.ctypes.data, name_len.ctypes.data,
                   results.ctypes.data, n, out.ctypes.data, cap, C.byref(good))
            if L != bad:
                break
            if cap > (1 << 34):
                raise _lib.UtreeError(_lib.E_NOMEM, "utree_format_records")
            cap *= 4
        return out[:L].tobytes()

    def close(self):
        if self._h:
            _lib.load().utree_ctr_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def frame_fasta(data: bytes, final: bool = True):
    """a2: frame reads as the reference's two fgets per read do (itree.c:866-890)."""
    buf = np.frombuffer(data, dtype=np.uint8)
    cap = max(1, data.count(b"\n") // 2 + 2)
    seq_off = np.zeros(cap, dtype=np.uint64)
    seq_len = np.zeros(cap, dtype=np.uint32)
    name_off = np.zeros(cap, dtype=np.uint64)
    name_len = np.zeros(cap, dtype=np.uint32)
    n = C.c_size_t(0)
    used = C.c_size_t(0)
    err = _lib.FastaError()
    rc = _lib.load().utree_fasta_frame(buf.ctypes.data if len(buf) else None, len(buf), int(final), cap,
                                       seq_off.ctypes.data, seq_len.ctypes.data, name_off.ctypes.data,
                                       name_len.ctypes.data, C.byref(n), C.byref(used), C.byref(err))
    if rc not in (_lib.OK, _lib.E_FASTA):
        _lib.check(rc, "utree_fasta_frame")
    k = n.value
    return dict(seq_off=seq_off[:k], seq_len=seq_len[:k], name_off=name_off[:k], name_len=name_len[:k],
                consumed=used.value, error_code=err.code if rc else 0, error_read=err.read_index)


def frame_reads(data: bytes, fmt: int, final: bool = True):
    """Opt-in input formats (FASTQ, multi-line FASTA).  Returns the framing AND the buffer: multi-line
    sequences are compacted in place, so offsets refer to the returned array, not to `data`."""
    buf = np.frombuffer(data, dtype=np.uint8).copy()
    cap = max(1, data.count(b"\n") + 2)
    seq_off = np.zeros(cap, dtype=np.uint64)
    seq_len = np.zeros(cap, dtype=np.uint32)
    name_off = np.zeros(cap, dtype=np.uint64)
    name_len = np.zeros(cap, dtype=np.uint32)
    n = C.c_size_t(0)
    used = C.c_size_t(0)
    err = _lib.FastaError()
    rc = _lib.load().utree_reads_frame(buf.ctypes.data if len(buf) else None, len(buf), int(final), fmt, cap,
                                       seq_off.ctypes.data, seq_len.ctypes.data, name_off.ctypes.data,
                                       name_len.ctypes.data, C.byref(n), C.byref(used), C.byref(err))
    if rc not in (_lib.OK, _lib.E_FASTA):
        _lib.check(rc, "utree_reads_frame")
    k = n.value
    return dict(buf=buf, seq_off=seq_off[:k], seq_len=seq_len[:k], name_off=name_off[:k], name_len=name_len[:k],
                consumed=used.value, error_code=err.code if rc else 0, error_read=err.read_index)


class DeviceTree:
    """The database resident in one GPU's HBM (device image)."""

    def __init__(self, handle, db: CtrDB, keepalive=None):
        self._h = C.c_void_p(handle)
        self.db = db
        self._keep = keepalive
        info = _lib.DevInfo()
        _lib.check(_lib.load().utree_dev_get_info(self._h, C.byref(info)), "utree_dev_get_info")
        self.info = info
        self._ws = None

    @classmethod
    def upload(cls, db: CtrDB, device: int = 0, fine_bits: int = _lib.FINE_AUTO) -> "DeviceTree":
        h = C.c_void_p()
        _lib.check(_lib.load().utree_dev_upload(db._h, device, fine_bits, C.byref(h)), "utree_dev_upload")
        return cls(h.value, db)

    @classmethod
    def build_from_device(cls, db: CtrDB, d_binix, d_records, device: int = 0, fine_bits: int = _lib.FINE_AUTO,
                          image=None) -> "DeviceTree":
        """d_binix / d_records: torch uint8 CUDA tensors holding the on-disk bin table and node dump."""
        import torch
        L = _lib.load()
        need = L.utree_dev_image_bytes(db._h, fine_bits)
        if image is None:
            image = torch.empty(need, dtype=torch.uint8, device="cuda:%d" % device)
        assert image.numel() >= need and image.is_cuda
        h = C.c_void_p()
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(L.utree_dev_build(db._h, device, fine_bits, d_binix.data_ptr(), d_records.data_ptr(), image.data_ptr(),
                                     image.numel(), stream, C.byref(h)), "utree_dev_build")
        return cls(h.value, db, keepalive=image)

    @classmethod
    def attach(cls, db: CtrDB, image, device: int) -> "DeviceTree":
        """Adopt an image received by torch.distributed.broadcast (RCCL) on this rank's GPU."""
        h = C.c_void_p()
        _lib.check(_lib.load().utree_dev_attach(db._h, device, image.data_ptr(), image.numel(), C.byref(h)),
                   "utree_dev_attach")
        return cls(h.value, db, keepalive=image)

    @staticmethod
    def rccl_unique_id() -> bytes:
        """Root side of the one-process-per-GPU replication: the id every rank's communicator is built from."""
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().utree_rccl_unique_id(buf, 128), "utree_rccl_unique_id")
        return buf.raw

    @classmethod
    def replicate_rank(cls, db: Optional[CtrDB], tree: Optional["DeviceTree"], device: int, rank: int, world: int, root: int,
                       uid: bytes) -> "DeviceTree":
        """utree_dev_replicate_rank: ONE ncclBroadcast (RCCL over xGMI) of the root's flat image issued from C; the root
        passes its tree and gets it back, the others pass tree=None and get a handle that owns the received copy."""
        h = C.c_void_p()
        _lib.check(_lib.load().utree_dev_replicate_rank(db._h if db is not None else None, tree._h if tree is not None else None,
                                                        device, rank, world, root, uid, len(uid), C.byref(h)),
                   "utree_dev_replicate_rank")
        if tree is not None and h.value == tree._h.value:
            return tree
        return cls(h.value, db if db is not None else (tree.db if tree is not None else None))   # a received copy (root: only under UTREE_RCCL_FORCE)

    @classmethod
    def replicate(cls, db: CtrDB, tree: "DeviceTree", devices) -> list:
        """utree_dev_replicate: one process, the image broadcast to `devices` (devices[0] = the tree's own).  Returns the handles;
        [0] is `tree` itself unless UTREE_RCCL_FORCE made it a replica on the same card."""
        n = len(devices)
        arr = (C.c_int * n)(*devices)
        out = (C.c_void_p * n)()
        _lib.check(_lib.load().utree_dev_replicate(db._h, tree._h, arr, n, out), "utree_dev_replicate")
        return [tree if out[i] == tree._h.value else cls(out[i], db) for i in range(n)]

    @staticmethod
    def replicate_seconds() -> float:
        return float(_lib.load().utree_dev_replicate_seconds())

    def image_tensor(self):
        """The flat image as a torch uint8 tensor view (for broadcast); only when torch owns the memory."""
        return self._keep

    def image_ptr(self):
        p = C.c_void_p()
        n = C.c_size_t()
        _lib.check(_lib.load().utree_dev_image(self._h, C.byref(p), C.byref(n)), "utree_dev_image")
        return p.value, n.value

    def get_ix(self, hi, lo):
        """XT_getIX32 (itree.c:720) for a batch of words. hi/lo: torch int64 CUDA tensors (hi may be None for k=32)."""
        import torch
        n = lo.numel()
        out = torch.empty(n, dtype=torch.int32, device=lo.device)
        stream = torch.cuda.current_stream(lo.device).cuda_stream
        _lib.check(_lib.load().utree_lookup_words(self._h, hi.data_ptr() if hi is not None else None, lo.data_ptr(), n,
                                                  out.data_ptr(), stream), "utree_lookup_words")
        return out

    def workspace_bytes(self, n_reads: int, total_bases: int, max_len: int, rc: bool) -> int:
        return _lib.load().utree_classify_workspace_bytes(self._h, n_reads, total_bases, max_len, int(rc))

    def classify(self, bases, off, length, rc: bool = False, total_bases: Optional[int] = None,
                 max_len: Optional[int] = None, out=None, workspace=None):
        """The hot path for one batch (a3-a9). bases: uint8 CUDA tensor; off: int64; length: int32.
        Returns an int32 [n, 6] CUDA tensor (label, cut, found, uix, sl, ol). Asynchronous on torch's
        current stream."""
        import torch
        n = off.numel()
        dev = bases.device
        if total_bases is None:
            total_bases = int(length.sum().item())
        if max_len is None:
            max_len = int(length.max().item()) if n else 0
        if out is None:
            out = torch.empty((n, 6), dtype=torch.int32, device=dev)
        need = self.workspace_bytes(n, total_bases, max_len, rc)
        if workspace is None:
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            workspace = self._ws
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.load().utree_classify_batch(self._h, bases.data_ptr(), off.data_ptr(), length.data_ptr(), n,
                                                    total_bases, max_len, int(rc), out.data_ptr(), workspace.data_ptr(),
                                                    workspace.numel(), stream), "utree_classify_batch")
        return out

    def poll(self):
        """utree_classify_poll: raises UtreeError(E_DEVICE) if a batch that has finished since the last call found its workspace
        too small (call after the stream has drained; its results are not to be used)."""
        _lib.check(_lib.load().utree_classify_poll(self._h), "utree_classify_poll")

    def rank_search(self, bases, off, length, rc: bool = False, slack: int = 2, sparsity: int = 4, tolerance: int = 2,
                    total_bases: Optional[int] = None, max_len: Optional[int] = None, out=None):
        """One batch of the rank-specific search (`xtree-search`, itree.c:969-1007).  Batches must come in file
        order: each read's vote also counts an entry left by an earlier read (itree.c:982); rank_reset() starts a
        new file.  Returns int32 [n, 6]: (mostIX, -2 printed / -4 not, hits kept, 0, most, secondMost)."""
        import torch
        n = off.numel()
        dev = bases.device
        if total_bases is None:
            total_bases = int(length.sum().item())
        if max_len is None:
            max_len = int(length.max().item()) if n else 0
        if out is None:
            out = torch.empty((n, 6), dtype=torch.int32, device=dev)
        prm = _lib.RankParams(slack, sparsity, tolerance)
        need = _lib.load().utree_rank_workspace_bytes(self._h, n, total_bases, max_len, int(rc), C.byref(prm))
        if n and not need:
            raise _lib.UtreeError(_lib.E_ARG, "utree_rank_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.load().utree_rank_batch(self._h, bases.data_ptr(), off.data_ptr(), length.data_ptr(), n,
                                                total_bases, max_len, int(rc), C.byref(prm), out.data_ptr(),
                                                self._ws.data_ptr(), self._ws.numel(), stream), "utree_rank_batch")
        return out

    def rank_reset(self):
        _lib.check(_lib.load().utree_rank_reset(self._h), "utree_rank_reset")

    def model_counts(self, bases, off, length, rc: bool = False) -> dict:
        """Measurement aid (bench.py's byte model): windows and distinct buckets / HBM lines per read of a batch."""
        import torch
        out = (C.c_uint64 * 5)()
        stream = torch.cuda.current_stream(bases.device).cuda_stream
        _lib.check(_lib.load().utree_model_counts(self._h, bases.data_ptr(), off.data_ptr(), length.data_ptr(), off.numel(), int(rc),
                                                  out, stream), "utree_model_counts")
        return dict(zip(("reads", "windows", "buckets", "lines128", "overflow_buckets"), [int(x) for x in out]))

    def kernel_name(self) -> str:
        return _lib.load().utree_classify_kernel_name(self._h).decode()

    def kernel_time(self, reset: bool = False):
        ms = C.c_double(0)
        n = C.c_uint64(0)
        _lib.check(_lib.load().utree_classify_kernel_time(self._h, int(reset), C.byref(ms), C.byref(n)),
                   "utree_classify_kernel_time")
        return ms.value, n.value

    def close(self):
        if self._h:
            _lib.load().utree_dev_free(self._h)
            self._h = None
            self._keep = None
            self._ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def search_gg(db: CtrDB, trees: Sequence[DeviceTree], fasta: str, out: str, rc: bool = False, threads: int = 0,
              input_format: int = _lib.INPUT_REFERENCE):
    """XT_doSearch32(utree, in, out, 8, speed, doRC) (itree.c:833): returns (code, stats); stats.fasta_error says which of
    the reference's exit(2) conditions a malformed read hit.  input_format != INPUT_REFERENCE opts into FASTQ / multi-line
    FASTA / gzip input."""
    arr = (C.c_void_p * len(trees))(*[t._h for t in trees])
    st = _lib.SearchStats()
    code = _lib.load().utree_search_file_opts(db._h, arr, len(trees), fasta.encode(), out.encode(), int(rc), threads,
                                              input_format, C.byref(st))
    return code, st


def search_rank(db: CtrDB, tree: DeviceTree, fasta: str, out: str, rc: bool = False, slack: int = 2, sparsity: int = 4,
                tolerance: int = 2, threads: int = 0, input_format: int = _lib.INPUT_REFERENCE):
    """XT_doSearch32(utree, in, out, 0, speed, doRC): the `xtree-search` binary (itree.c:1376 without DO_GG)."""
    st = _lib.SearchStats()
    prm = _lib.RankParams(slack, sparsity, tolerance)
    code = _lib.load().utree_rank_search_file_opts(db._h, tree._h, fasta.encode(), out.encode(), int(rc), C.byref(prm),
                                                   threads, input_format, C.byref(st))
    return code, st


def build(fasta: str, mapfile: str, ubt: str, W: int = 8, I: int = 2, complevel: int = 1, gg: bool = True, device: int = 0):
    """`utree-build[GG] in.fa labels.map out.ubt threads complevel` (itree.c:1379-1407): returns (code, stats)."""
    st = _lib.BuildStats()
    code = _lib.load().utree_build_file(fasta.encode(), mapfile.encode(), ubt.encode(), W, I, complevel, int(gg), device,
                                        C.byref(st))
    return code, st


def compress(ubt: str, ctr: str, device: int = 0):
    """XT_cmp32(preTree.ubt, compTree.ctr) (itree.c:1234): returns (code, stats)."""
    st = _lib.CompressStats()
    code = _lib.load().utree_compress_file(ubt.encode(), ctr.encode(), device, C.byref(st))
    return code, st


def classify_fasta_bytes(db: CtrDB, tree: DeviceTree, data: bytes, rc: bool = False) -> bytes:
    """Convenience for tests: frame -> upload -> classify -> format, through the C-ABI pieces."""
    import torch
    fr = frame_fasta(data, final=True)
    n = len(fr["seq_off"])
    dev = "cuda:%d" % tree.info.device
    if n == 0:
        return b""
    buf = np.frombuffer(data, dtype=np.uint8)
    d_buf = torch.from_numpy(buf.copy()).to(dev)
    d_off = torch.from_numpy(fr["seq_off"].astype(np.int64)).to(dev)
    d_len = torch.from_numpy(fr["seq_len"].astype(np.int32)).to(dev)
    res = tree.classify(d_buf, d_off, d_len, rc=rc, total_bases=int(fr["seq_len"].sum()),
                        max_len=int(fr["seq_len"].max()))
    torch.cuda.synchronize()
    tree.poll()
    h = res.cpu().numpy()
    return db.format(buf, fr["name_off"], fr["name_len"], h)
