"""Read / write UTree `.ctr` (compressed tree) and `.ubt` files with numpy.

The `.ctr` layout is the unchanged contract of the reference (writer: itree.c:1301-1313, reader:
itree.c:736-775):

    u64[4]  {W, 0, I, N}            W = bytes per packed k-mer word, I = bytes per label index
    (2^24+1) bin starts             4 bytes each iff N < UINT32_MAX, else 8
    N records of SZ = W+I-3 bytes   low W-3 bytes of the little-endian word, then I bytes label index
    text                            "label\\tcount\\n" per label, to EOF

This module is a utility for tests, synthetic-database generation and `bench.py`; the search product
loads `.ctr` files through the C-ABI loader (utree_amd/csrc/ctr_host.c), not through this file.
"""
from __future__ import annotations

import hashlib
import io
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

NUMBINS = (1 << 24) + 1  # itree.c:693
UINT32_MAX = 0xFFFFFFFF


@dataclass
class CtrData:
    W: int                      # bytes per word (8 for k=32, 16 for k=64)
    I: int                      # bytes per label index (2 or 4)
    n_nodes: int
    binix: np.ndarray           # uint64[NUMBINS]
    records: np.ndarray         # uint8[n_nodes, SZ]
    label_text: bytes           # the file tail, verbatim

    @property
    def SZ(self) -> int:
        return self.W + self.I - 3

    @property
    def k(self) -> int:
        return 4 * self.W

    def labels(self) -> List[str]:
        """First-seen-order unique labels (itree.c:1154-1223)."""
        seen = {}
        out = []
        lines = self.label_text.split(b"\n")
        if lines and lines[-1] == b"":
            lines.pop()          # the piece after the final newline is not a line
        for line in lines:
            lab = line.split(b"\t", 1)[0].decode("latin-1")
            if lab not in seen:
                seen[lab] = len(out)
                out.append(lab)
        return out

    def suffixes(self) -> Tuple[np.ndarray, np.ndarray]:
        """(hi, lo) uint64 arrays of the stored suffix (low 8*(W-3) bits of each word)."""
        sb = self.W - 3
        pad = np.zeros((self.n_nodes, 16), dtype=np.uint8)
        pad[:, :sb] = self.records[:, :sb]
        v = pad.view("<u8")
        return v[:, 1].copy(), v[:, 0].copy()

    def ix(self) -> np.ndarray:
        sb = self.W - 3
        pad = np.zeros((self.n_nodes, 4), dtype=np.uint8)
        pad[:, : self.I] = self.records[:, sb : sb + self.I]
        return pad.view("<u4")[:, 0].copy()

    def words(self) -> Tuple[np.ndarray, np.ndarray]:
        """Full words (hi, lo) reconstructed from bin membership + suffix."""
        counts = np.diff(self.binix.astype(np.int64))
        counts = np.maximum(counts, 0)
        prefix = np.repeat(np.arange(NUMBINS - 1, dtype=np.uint64), counts)
        # records before the first bin start or after the last are not addressable; assume none
        hi, lo = self.suffixes()
        if self.W == 8:
            return np.zeros_like(lo), lo | (prefix << np.uint64(40))
        if self.W == 16:
            return hi | (prefix << np.uint64(40)), lo
        if self.W == 4:
            return np.zeros_like(lo), lo | (prefix << np.uint64(8))
        raise ValueError("unsupported W")


def read_ctr(path: str) -> CtrData:
    with open(path, "rb") as f:
        meta = np.frombuffer(f.read(32), dtype="<u8")
        W, cnt, I, N = (int(x) for x in meta)
        if cnt != 0:
            raise ValueError("count field not supported")
        ixsz = 4 if N < UINT32_MAX else 8
        b = np.frombuffer(f.read(NUMBINS * ixsz), dtype="<u4" if ixsz == 4 else "<u8").astype(np.uint64)
        SZ = W + I - 3
        rec = np.frombuffer(f.read(N * SZ), dtype=np.uint8).reshape(N, SZ).copy()
        text = f.read()
    return CtrData(W, I, N, b, rec, text)


def pack_records(W: int, I: int, hi: np.ndarray, lo: np.ndarray, ix: np.ndarray) -> np.ndarray:
    """SZ-byte records from full words (hi:lo) and label indices."""
    n = len(lo)
    sb = W - 3
    full = np.zeros((n, 16), dtype=np.uint8)
    full[:, :8] = np.ascontiguousarray(lo.astype("<u8")).view(np.uint8).reshape(n, 8)
    full[:, 8:] = np.ascontiguousarray(hi.astype("<u8")).view(np.uint8).reshape(n, 8)
    rec = np.empty((n, sb + I), dtype=np.uint8)
    rec[:, :sb] = full[:, :sb]
    rec[:, sb:] = np.ascontiguousarray(ix.astype("<u4")).view(np.uint8).reshape(n, 4)[:, :I]
    return rec


def word_prefix(W: int, hi: np.ndarray, lo: np.ndarray) -> np.ndarray:
    """Top 24 bits of the 2k-bit word (itree.c:684 PREFIX_L)."""
    if W == 8:
        return (lo >> np.uint64(40)).astype(np.int64)
    if W == 16:
        return (hi >> np.uint64(40)).astype(np.int64)
    if W == 4:
        return (lo >> np.uint64(8)).astype(np.int64)
    raise ValueError("unsupported W")


def binix_exact(prefix: np.ndarray, n: int) -> np.ndarray:
    """Bin starts for ascending words: bin p = [binix[p], binix[p+1])."""
    counts = np.bincount(prefix, minlength=NUMBINS - 1).astype(np.uint64)
    b = np.zeros(NUMBINS, dtype=np.uint64)
    np.cumsum(counts, out=b[1:])
    assert int(b[-1]) == n
    return b


def binix_like_compress(prefix: np.ndarray, n: int) -> np.ndarray:
    """Bin starts exactly as the reference's COMPRESS computes them (itree.c:1281-1289), including its
    first-bin quirk: `if(!BinIx[v]) BinIx[v]=i` cannot tell "unset" from
    "starts at 0"; the first non-zero entry is then zeroed and only entries above it are back-filled."""
    b = np.zeros(NUMBINS, dtype=np.uint64)
    # first index i at which each prefix occurs, but an entry can only be set to a non-zero i
    idx = np.arange(n, dtype=np.uint64)
    nz = idx != 0
    p_nz = prefix[nz]
    i_nz = idx[nz]
    # minimum non-zero i per prefix
    first = np.full(NUMBINS - 1, np.iinfo(np.uint64).max, dtype=np.uint64)
    np.minimum.at(first, p_nz, i_nz)
    has = first != np.iinfo(np.uint64).max
    b[:-1][has] = first[has]
    b[NUMBINS - 1] = n
    u = int(np.flatnonzero(b)[0])
    b[u] = 0
    # back-fill zeros above u from the right
    for_fill = b.copy()
    # vectorised back-fill: positions with zero take the next non-zero to the right
    arr = for_fill[u + 1 :]
    zero = arr == 0
    if zero.any():
        # index of next non-zero at or after each position
        pos = np.where(~zero, np.arange(len(arr)), len(arr) + 10)
        nxt = np.minimum.accumulate(pos[::-1])[::-1]
        arr = arr[np.minimum(nxt, len(arr) - 1)]
        for_fill[u + 1 :] = arr
    return for_fill


def write_ctr(path: str, W: int, I: int, hi: np.ndarray, lo: np.ndarray, ix: np.ndarray,
              labels: Sequence[str], label_counts: Optional[Sequence[int]] = None,
              binix: Optional[np.ndarray] = None, like_compress: bool = False) -> None:
    """Write a `.ctr`. Words must already be in ascending order (as the reference's `.ubt` is)."""
    n = len(lo)
    if binix is None:
        pref = word_prefix(W, hi, lo)
        binix = binix_like_compress(pref, n) if like_compress else binix_exact(pref, n)
    rec = pack_records(W, I, hi, lo, ix)
    if label_counts is None:
        label_counts = np.bincount(ix.astype(np.int64), minlength=len(labels))[: len(labels)]
    with open(path, "wb") as f:
        f.write(np.array([W, 0, I, n], dtype="<u8").tobytes())
        if n < UINT32_MAX:
            f.write(binix.astype("<u4").tobytes())
        else:
            f.write(binix.astype("<u8").tobytes())
        f.write(rec.tobytes())
        out = io.BytesIO()
        for lab, c in zip(labels, label_counts):
            out.write(lab.encode("latin-1") + b"\t" + str(int(c)).encode() + b"\n")
        f.write(out.getvalue())


def read_ubt(path: str):
    """`.ubt`: u64[4]{W,0,I,N} + N x (W-byte word, I-byte ix) ascending + label lines (itree.c:1317-1343)."""
    with open(path, "rb") as f:
        meta = np.frombuffer(f.read(32), dtype="<u8")
        W, cnt, I, N = (int(x) for x in meta)
        rec = np.frombuffer(f.read(N * (W + I)), dtype=np.uint8).reshape(N, W + I)
        text = f.read()
    full = np.zeros((N, 16), dtype=np.uint8)
    full[:, :W] = rec[:, :W]
    v = full.view("<u8")
    ixb = np.zeros((N, 4), dtype=np.uint8)
    ixb[:, :I] = rec[:, W:]
    return W, I, v[:, 1].copy(), v[:, 0].copy(), ixb.view("<u4")[:, 0].copy(), text


def write_ubt(path: str, W: int, I: int, hi: np.ndarray, lo: np.ndarray, ix: np.ndarray, label_text: bytes) -> None:
    """`.ubt` as UT_writeTreeBinary leaves it (itree.c:1317-1343): header, N x (W-byte LE word, I-byte ix), label lines."""
    n = len(lo)
    full = np.zeros((n, 16), dtype=np.uint8)
    full[:, :8] = np.ascontiguousarray(lo.astype("<u8")).view(np.uint8).reshape(n, 8)
    full[:, 8:] = np.ascontiguousarray(hi.astype("<u8")).view(np.uint8).reshape(n, 8)
    rec = np.empty((n, W + I), dtype=np.uint8)
    rec[:, :W] = full[:, :W]
    rec[:, W:] = np.ascontiguousarray(ix.astype("<u4")).view(np.uint8).reshape(n, 4)[:, :I]
    with open(path, "wb") as f:
        f.write(np.array([W, 0, I, n], dtype="<u8").tobytes())
        f.write(rec.tobytes())
        f.write(label_text)


def sha256_file(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def encode_kmers(seqs: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """Pack equal-length ACGT strings (k<=64) first-base-most-significant into (hi, lo) uint64."""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    hi = np.zeros(len(seqs), dtype=np.uint64)
    lo = np.zeros(len(seqs), dtype=np.uint64)
    for n, s in enumerate(seqs):
        v = 0
        for ch in s:
            v = (v << 2) | code[ch.upper()]
        hi[n] = (v >> 64) & 0xFFFFFFFFFFFFFFFF
        lo[n] = v & 0xFFFFFFFFFFFFFFFF
    return hi, lo


def decode_kmer(hi: int, lo: int, k: int) -> str:
    v = (int(hi) << 64) | int(lo)
    return "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))
