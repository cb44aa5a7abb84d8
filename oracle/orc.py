"""ctypes wrapper over oracle/liboracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product package utree_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BAD_IX = 0xFFFFFFFF


class Result(C.Structure):
    _fields_ = [("label", C.c_uint32), ("cut", C.c_int32), ("found", C.c_uint32), ("uix", C.c_uint32),
                ("sl", C.c_uint32), ("ol", C.c_uint32)]


RESULT_DTYPE = np.dtype([("label", "<u4"), ("cut", "<i4"), ("found", "<u4"), ("uix", "<u4"),
                         ("sl", "<u4"), ("ol", "<u4")])


class RankParams(C.Structure):
    """SLACK / SPARSITY / TOLERANCE_THRESHOLD (itree.c:952-960)."""
    _fields_ = [("slack", C.c_uint32), ("sparsity", C.c_uint32), ("tolerance", C.c_uint32)]


class RankResult(C.Structure):
    _fields_ = [("label", C.c_uint32), ("printed", C.c_uint32), ("found", C.c_uint32), ("most", C.c_uint32),
                ("second", C.c_uint32)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "utree_oracle.c")
    src2 = os.path.join(_HERE, "utree_build_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(src2)):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_db_load.restype = C.c_void_p
        L.orc_db_load.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.orc_db_from_memory.restype = C.c_void_p
        L.orc_db_from_memory.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_char_p,
                                         C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_db_free.argtypes = [C.c_void_p]
        for f in ("orc_db_W", "orc_db_I", "orc_db_labels"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_db_nodes.restype = C.c_uint64
        L.orc_db_nodes.argtypes = [C.c_void_p]
        L.orc_db_label.restype = C.c_char_p
        L.orc_db_label.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_windows.restype = C.c_size_t
        L.orc_windows.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_lookup.restype = C.c_uint32
        L.orc_lookup.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.orc_vote.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Result)]
        L.orc_format.restype = C.c_size_t
        L.orc_format.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(Result), C.c_char_p, C.c_size_t]
        L.orc_classify_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(Result)]
        L.orc_classify_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                         C.c_void_p]
        L.orc_search_file.restype = C.c_int
        L.orc_search_file.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        L.orc_rank_state_new.restype = C.c_void_p
        L.orc_rank_state_new.argtypes = [C.c_void_p]
        L.orc_rank_state_free.argtypes = [C.c_void_p]
        L.orc_rank_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(RankParams),
                                    C.POINTER(RankResult)]
        L.orc_rank_format.restype = C.c_size_t
        L.orc_rank_format.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(RankResult), C.c_char_p, C.c_size_t]
        L.orc_rank_search_file.restype = C.c_int
        L.orc_rank_search_file.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(RankParams),
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        L.orc_build_file.restype = C.c_int
        L.orc_build_file.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        _LIB = L
    return _LIB


class OracleDB:
    """XT_read32 restated (itree.c:733)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load(cls, path: str) -> "OracleDB":
        err = C.create_string_buffer(256)
        h = lib().orc_db_load(path.encode(), err, 256)
        if not h:
            raise RuntimeError("oracle load failed: " + err.value.decode())
        return cls(h)

    @classmethod
    def from_memory(cls, W, I, binix_u64: np.ndarray, records_u8: np.ndarray, label_text: bytes) -> "OracleDB":
        err = C.create_string_buffer(256)
        binix_u64 = np.ascontiguousarray(binix_u64, dtype=np.uint64)
        records_u8 = np.ascontiguousarray(records_u8, dtype=np.uint8)
        n = records_u8.size // (W + I - 3)
        h = lib().orc_db_from_memory(W, I, n, binix_u64.ctypes.data, records_u8.ctypes.data, label_text,
                                     len(label_text), err, 256)
        if not h:
            raise RuntimeError("oracle from_memory failed: " + err.value.decode())
        return cls(h)

    def close(self):
        if self._h:
            lib().orc_db_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def W(self): return lib().orc_db_W(self._h)
    @property
    def I(self): return lib().orc_db_I(self._h)
    @property
    def k(self): return 4 * self.W
    @property
    def n_nodes(self): return lib().orc_db_nodes(self._h)
    @property
    def n_labels(self): return lib().orc_db_labels(self._h)

    def label(self, ix: int) -> bytes:
        return lib().orc_db_label(self._h, ix)

    def lookup(self, hi: int, lo: int) -> int:
        return lib().orc_lookup(self._h, hi, lo)

    def vote(self, hits) -> Result:
        a = np.ascontiguousarray(hits, dtype=np.uint32)
        r = Result()
        lib().orc_vote(self._h, a.ctypes.data, len(a), C.byref(r))
        return r

    def classify_read(self, seq: bytes, rc: bool = False) -> Result:
        r = Result()
        b = np.frombuffer(seq, dtype=np.uint8)
        lib().orc_classify_read(self._h, b.ctypes.data if len(b) else None, len(b), int(rc), C.byref(r))
        return r

    def classify_batch(self, buf: np.ndarray, off: np.ndarray, length: np.ndarray, rc: bool = False,
                       threads: int = 0) -> np.ndarray:
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        out = np.zeros(len(off), dtype=RESULT_DTYPE)
        lib().orc_classify_batch(self._h, buf.ctypes.data, off.ctypes.data, length.ctypes.data, len(off), int(rc),
                                 threads, out.ctypes.data)
        return out

    def format(self, name: bytes, r: Result) -> bytes:
        cap = len(name) + 70000
        out = C.create_string_buffer(cap)
        n = lib().orc_format(self._h, name, len(name), C.byref(r), out, cap)
        return out.raw[:n]

    def search_file(self, fasta: str, out: str, threads: int = 1, rc: bool = False):
        """XT_doSearch32 GG branch restated (itree.c:833). Returns (exit_code, n_reads, good_finds, err)."""
        nr = C.c_uint64(0)
        gf = C.c_uint64(0)
        err = C.create_string_buffer(512)
        code = lib().orc_search_file(self._h, fasta.encode(), out.encode(), threads, int(rc), C.byref(nr), C.byref(gf),
                                     err, 512)
        return code, nr.value, gf.value, err.value.decode("latin-1")


class RankSearch:
    """`xtree-search` restated (itree.c:969-1007): reads go through in file order, one state object."""

    def __init__(self, db: OracleDB, slack: int = 2, sparsity: int = 4, tolerance: int = 2):
        self.db = db
        self.prm = RankParams(slack, sparsity, tolerance)
        self._st = lib().orc_rank_state_new(db._h)

    def read(self, seq: bytes, rc: bool = False) -> RankResult:
        r = RankResult()
        b = np.frombuffer(seq, dtype=np.uint8)
        lib().orc_rank_read(self.db._h, self._st, b.ctypes.data if len(b) else None, len(b), int(rc),
                            C.byref(self.prm), C.byref(r))
        return r

    def format(self, name: bytes, r: RankResult) -> bytes:
        cap = len(name) + 70000
        out = C.create_string_buffer(cap)
        n = lib().orc_rank_format(self.db._h, name, len(name), C.byref(r), out, cap)
        return out.raw[:n]

    def close(self):
        if self._st:
            lib().orc_rank_state_free(self._st)
            self._st = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rank_search_file(db: OracleDB, fasta: str, out: str, rc: bool = False, slack: int = 2, sparsity: int = 4,
                     tolerance: int = 2):
    """Returns (exit_code, n_reads, good_finds, err)."""
    nr = C.c_uint64(0)
    gf = C.c_uint64(0)
    err = C.create_string_buffer(512)
    prm = RankParams(slack, sparsity, tolerance)
    code = lib().orc_rank_search_file(db._h, fasta.encode(), out.encode(), int(rc), C.byref(prm), C.byref(nr),
                                      C.byref(gf), err, 512)
    return code, nr.value, gf.value, err.value.decode("latin-1")


def build_file(fasta: str, mapfile: str, out_ubt: str, W: int = 8, I: int = 2, complevel: int = 1, gg: bool = True):
    """`utree-build[GG] in.fa labels.map out.ubt threads complevel` restated.  Returns (exit_code, seqs, nodes, labels, err)."""
    ns, nn, nl = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    err = C.create_string_buffer(512)
    code = lib().orc_build_file(fasta.encode(), mapfile.encode(), out_ubt.encode(), W, I, complevel, int(gg), C.byref(ns),
                                C.byref(nn), C.byref(nl), err, 512)
    return code, ns.value, nn.value, nl.value, err.value.decode("latin-1")


def windows(seq: bytes, k: int):
    b = np.frombuffer(seq, dtype=np.uint8)
    cap = max(len(b), 1)
    pos = np.zeros(cap, dtype=np.uint32)
    hi = np.zeros(cap, dtype=np.uint64)
    lo = np.zeros(cap, dtype=np.uint64)
    n = lib().orc_windows(b.ctypes.data if len(b) else None, len(b), k, pos.ctypes.data, hi.ctypes.data, lo.ctypes.data,
                          cap)
    return pos[:n], hi[:n], lo[:n]
