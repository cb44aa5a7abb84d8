#!/usr/bin/env python3
"""Runs of windows that share their sampled position, per 150 bp read (k = 32: 119 windows) -- what decides how many table lines a read
fetches -- for the window schemes VERDICT r02 item 3 names, on random sequence (numpy, CPU; no product code involved).

  minimizer(m)      the m-mer of the window with the smallest hash (leftmost on ties): w = 33 - m positions per window
  oc(m, s)          open-closed syncmer order: m-mers whose smallest s-mer sits in the middle first, then those with it at an end, then
                    the rest; hash order inside a class (Groot Koerkamp, Liu, Pibiri 2025)
  mod(m, t)         mod-minimizer: position of the smallest t-mer of the window, mod w (Groot Koerkamp & Pibiri 2024; pays for m > w)
  floor             1 / w: no scheme samples fewer positions than one per w windows

A bucket is addressed by the sampled m-mer, so m also fixes how many database k-mers share an address: with N = 1.2e9 nodes there are
N w / 4^m nodes per m-mer VALUE at the dense end of the hash order (the sampled m-mer is a minimum) -- the last column."""
import numpy as np

rng = np.random.default_rng(1)
K, L, R = 32, 150, 3000
seq = rng.integers(0, 4, (R, L), dtype=np.uint64)


def mers(m):
    v = np.zeros((R, L - m + 1), dtype=np.uint64)
    for j in range(m):
        v = (v << np.uint64(2)) | seq[:, j:L - m + 1 + j]
    return v


def mix(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9); x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB); x ^= x >> np.uint64(31)
    return x


def runs_from_positions(pos):
    # pos[r, j]: absolute sampled position of window j of read r
    return 1 + (pos[:, 1:] != pos[:, :-1]).sum(1)


def sliding_argmin(key, w):
    n = key.shape[1] - w + 1
    idx = np.arange(n)[:, None] + np.arange(w)[None, :]
    win = key[:, idx]                                      # R x n x w
    return win.argmin(2) + np.arange(n)[None, :]


def minimizer(m):
    w = K - m + 1
    return runs_from_positions(sliding_argmin(mix(mers(m)), w))


def oc(m, s):
    w = K - m + 1
    sm = mix(mers(s))
    inner = m - s + 1
    p = sliding_argmin(sm, inner) - np.arange(L - m + 1)[None, :]      # position of the smallest s-mer inside each m-mer
    cls = np.where(p == (inner - 1) // 2, 0, np.where((p == 0) | (p == inner - 1), 1, 2)).astype(np.uint64)
    key = (cls << np.uint64(62)) | (mix(mers(m)) >> np.uint64(2))
    return runs_from_positions(sliding_argmin(key, w))


def mod(m, t):
    w = K - m + 1
    tm = mix(mers(t))
    n = L - K + 1
    x = sliding_argmin(tm, K - t + 1)[:, :n] - np.arange(n)[None, :]   # position of the smallest t-mer inside each window
    return runs_from_positions(np.arange(n)[None, :] + (x % w))


N = 1.217e9
rows = [("minimizer(16)  [shipped]", minimizer(16), 16), ("minimizer(14)", minimizer(14), 14), ("minimizer(12)", minimizer(12), 12),
        ("minimizer(20)", minimizer(20), 20), ("oc(16, s=5)", oc(16, 5), 16), ("oc(16, s=8)", oc(16, 8), 16), ("oc(14, s=5)", oc(14, 5), 14),
        ("mod(20, t=7)", mod(20, 7), 20), ("mod(24, t=6)", mod(24, 6), 24)]
print("%-26s %5s %12s %14s %22s" % ("scheme", "w", "runs / read", "floor 1 + 118/w", "nodes per m-mer value"))
for name, r, m in rows:
    w = K - m + 1
    print("%-26s %5d %12.2f %14.2f %22.3g" % (name, w, r.mean(), 1 + 118.0 / w, N * w / 4.0 ** m))
