#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the GENUINE reference.

Run once in the build container (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Inputs are produced by this script's own seeded generators; expected outputs come from the reference
binaries in oracle/_ref/ (itree.c compiled with -D SEARCH_GG / BUILD_GG / COMPRESS, 1 thread so that the
output order is the input order).  Only data is committed: inputs, the reference's outputs, and compact
node lists parsed out of the reference-built databases.  No reference source is stored here.

Fixture sets:
  toy     F1  reference BUILD_GG(complevel 2)+COMPRESS on 1000 seeded refs; 10 000 x 100 bp reads; +RC
  edge    F1  parser edge cases on the toy DB (CRLF, tabs, lowercase, N, short, no trailing newline)
  vote    F2  hand-built label sets with planted hit multisets (every vote branch)
  kat     F3  direct-written DB: bin sizes 1,2,3.., min/max suffix, first-bin quirk; one probe per read
  k64     F4  PACKSIZE=64 build of a smaller toy + reads (+RC)
  ix32    F4  IXTYPE=uint32_t build of a smaller toy + reads
  k64ix32 F4  PACKSIZE=64 and IXTYPE=uint32_t together (+RC)
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from utree_amd import ctrfile  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def run(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout, r.stderr


def rand_seq(rng, n):
    return BASES[rng.integers(0, 4, n)]


def mutate(rng, seq, rate):
    s = seq.copy()
    m = rng.random(len(s)) < rate
    s[m] = BASES[rng.integers(0, 4, int(m.sum()))]
    return s


RANKS = ["k", "p", "c", "o", "f", "g", "s", "t"]


def make_taxonomy(rng, n_leaves, fan=(2, 3, 2, 3, 2, 3, 3, 3), empty_tail_frac=0.15):
    """Random 8-rank GG-style strings; some leaves have empty trailing ranks (s__;t__)."""
    names = {}

    def nm(level, path):
        key = (level, path)
        if key not in names:
            names[key] = "%s%d" % ("BPCOFGST"[level], len([k for k in names if k[0] == level]))
        return names[key]

    leaves = []
    for _ in range(n_leaves):
        path = tuple(int(rng.integers(0, f)) for f in fan)
        toks = []
        depth = 8
        r = rng.random()
        if r < empty_tail_frac:
            depth = int(rng.integers(5, 8))      # ranks >= depth are empty
        for lv in range(8):
            if lv < depth:
                toks.append("%s__%s" % (RANKS[lv], nm(lv, path[: lv + 1])))
            else:
                toks.append("%s__" % RANKS[lv])
        leaves.append((path, ";".join(toks)))
    return leaves


def make_refs(rng, leaves, length, per_level_rate=0.012):
    """Hierarchically mutated sequences so that related taxa share k-mers."""
    cache = {(): rand_seq(rng, length)}

    def seq_for(path):
        if path not in cache:
            cache[path] = mutate(rng, seq_for(path[:-1]), per_level_rate)
        return cache[path]

    out = []
    for n, (path, tax) in enumerate(leaves):
        s = mutate(rng, seq_for(path), 0.004)
        out.append(("ref%05d sample" % n, tax, s))
    return out


def write_fasta(path, items, crlf=False):
    nl = b"\r\n" if crlf else b"\n"
    with open(path, "wb") as f:
        for name, seq in items:
            f.write(b">" + name.encode("latin-1") + nl + bytes(seq) + nl)


def sample_reads(rng, refs, n, length, prefix="q"):
    reads = []
    L = len(refs[0][2])
    for i in range(n):
        r = rng.random()
        ref = refs[int(rng.integers(0, len(refs)))][2]
        if r < 0.08:
            s = rand_seq(rng, length)                                   # no hit expected
        elif r < 0.18:                                                  # chimera of two refs
            ref2 = refs[int(rng.integers(0, len(refs)))][2]
            a = int(rng.integers(0, L - length))
            b = int(rng.integers(0, L - length))
            h = length // 2
            s = np.concatenate([ref[a : a + h], ref2[b : b + length - h]])
        else:
            a = int(rng.integers(0, L - length))
            s = mutate(rng, ref[a : a + length], 0.01 if r < 0.7 else 0.0)
        s = s.copy()
        q = rng.random()
        if q < 0.03:
            s[int(rng.integers(0, len(s)))] = ord("N")
        elif q < 0.05:
            s = np.frombuffer(bytes(s).lower(), dtype=np.uint8).copy()
        elif q < 0.06:
            s = s[: int(rng.integers(1, 40))]                           # shorter than / around k
        elif q < 0.08:                                                  # reverse strand
            comp = {65: 84, 67: 71, 71: 67, 84: 65}
            s = np.array([comp.get(int(c), 78) for c in s[::-1]], dtype=np.uint8)
        reads.append(("%s%d" % (prefix, i), s))
    return reads


def ref_search(binname, ctr, fasta, rc):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "out.txt")
        cmd = [os.path.join(REF, binname), ctr, fasta, out, "1"] + (["RC"] if rc else [])
        code, so, se = run(cmd)
        data = open(out, "rb").read() if os.path.exists(out) else b""
        return code, data, so, se


def gz_write(path, data):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(data)


def build_with_reference(td, refs, suffix, complevel):
    fa = os.path.join(td, "refs.fa")
    mp = os.path.join(td, "refs.map")
    write_fasta(fa, [(n, s) for n, _, s in refs])
    with open(mp, "wb") as f:
        for n, tax, _ in refs:
            f.write(n.encode() + b"\t" + tax.encode() + b"\n")
    ubt = os.path.join(td, "db.ubt")
    ctr = os.path.join(td, "db.ctr")
    code, so, se = run([os.path.join(REF, "utree-buildGG" + suffix), fa, mp, ubt, "1", str(complevel)])
    assert code == 0, (code, so[-500:], se[-500:])
    code, so, se = run([os.path.join(REF, "xtree-compress" + suffix), ubt, ctr])
    assert code == 0, (code, so[-500:], se[-500:])
    return ctr


def save_db_fixture(name, ctr_path, manifest):
    d = ctrfile.read_ctr(ctr_path)
    hi, lo = d.suffixes()
    # records in file order + the bin table run-length coded (it is piecewise constant): exact for any
    # table, including the first-bin-quirk one
    b = d.binix
    chg = (np.flatnonzero(b[1:] != b[:-1]) + 1).astype(np.uint32)
    np.savez_compressed(
        os.path.join(HERE, name + "_db.npz"),
        W=np.int64(d.W), I=np.int64(d.I),
        suf_hi=hi if d.W == 16 else np.zeros(0, dtype=np.uint64), suf_lo=lo, ix=d.ix().astype(np.uint32),
        bin0=np.uint64(b[0]), bin_chg_idx=chg, bin_chg_val=b[chg].astype(np.uint64),
        label_text=np.frombuffer(d.label_text, dtype=np.uint8),
    )
    manifest[name + "_ctr_sha256"] = ctrfile.sha256_file(ctr_path)
    manifest[name + "_nodes"] = int(d.n_nodes)
    manifest[name + "_labels"] = len(d.labels())
    return d


def gen_toy(manifest, name, suffix, searchbin, n_refs, ref_len, n_reads, read_len, seed, complevel=2):
    rng = np.random.default_rng(seed)
    leaves = make_taxonomy(rng, n_refs)
    refs = make_refs(rng, leaves, ref_len)
    with tempfile.TemporaryDirectory() as td:
        ctr = build_with_reference(td, refs, suffix, complevel)
        d = save_db_fixture(name, ctr, manifest)
        reads = sample_reads(rng, refs, n_reads, read_len)
        fa = os.path.join(td, "reads.fa")
        write_fasta(fa, reads)
        gz_write(os.path.join(HERE, name + "_reads.fa.gz"), open(fa, "rb").read())
        for rc in (0, 1):
            code, out, so, se = ref_search(searchbin, ctr, fa, rc)
            assert code == 0, (code, se)
            gz_write(os.path.join(HERE, "%s_out%s.txt.gz" % (name, "_rc" if rc else "")), out)
            manifest["%s_out%s_lines" % (name, "_rc" if rc else "")] = out.count(b"\n")
        if name == "toy":
            gen_edge(manifest, td, ctr, refs, rng)
    return d


def gen_edge(manifest, td, ctr, refs, rng):
    """Parser edge cases, each file run separately (some make the reference exit non-zero)."""
    r0 = bytes(refs[3][2][100:200])
    r1 = bytes(refs[7][2][300:420])
    cases = {}
    cases["crlf"] = b">readA extra words\r\n" + r0 + b"\r\n>readB\r\n" + r1 + b"\r\n"
    cases["tabs_and_space"] = b">read\twith\ttabs and space\n" + r0 + b"\n>x y\n" + r1 + b"\n"
    cases["no_trailing_newline"] = b">r1\n" + r0 + b"\n>r2\n" + r1
    cases["lower_and_N"] = b">low\n" + r0.lower() + b"\n>nn\n" + r0[:50] + b"N" + r0[51:] + b"\n>n2\n" + r1[:33] + b"NN" + r1[35:70] + b"n" + r1[71:] + b"\n"
    cases["short"] = b">s1\n" + r0[:31] + b"\n>s2\n" + r0[:32] + b"\n>s3\nA\n>s4\n" + r0[:33] + b"\n"
    cases["blank_seq"] = b">b1\n\n>b2\n" + r0 + b"\n"
    cases["iupac_and_misc"] = b">iu\n" + r0[:40] + b"RYKM" + r0[44:] + b"\n>gap\n" + r0[:60] + b"-" + r0[61:] + b"\n>dot\n" + r1[:64] + b"." + r1[65:] + b"\n"
    cases["hdr_only_gt"] = b">\n" + r0 + b"\n> lead space\n" + r1 + b"\n"
    cases["err_no_header"] = b">ok\n" + r0 + b"\nnot a header\n" + r1 + b"\n"
    cases["err_seq_is_header"] = b">ok\n" + r0 + b"\n>h1\n>h2\n" + r1 + b"\n"
    cases["err_missing_seq"] = b">ok\n" + r0 + b"\n>last\n"
    cases["err_blank_line"] = b">ok\n" + r0 + b"\n\n>h\n" + r1 + b"\n"
    out = {}
    for nm, data in cases.items():
        fa = os.path.join(td, "edge_%s.fa" % nm)
        open(fa, "wb").write(data)
        for rc in (0, 1):
            code, res, so, se = ref_search("xtree-searchGG", ctr, fa, rc)
            out["%s%s" % (nm, "_rc" if rc else "")] = {
                "input_hex": data.hex(), "rc": rc, "exit": code, "output_hex": res.hex(),
            }
    json.dump(out, open(os.path.join(HERE, "edge_cases.json"), "w"), indent=0, sort_keys=True)
    manifest["edge_cases"] = len(out)


def unique_kmers(rng, n, k):
    seen = set()
    out = []
    while len(out) < n:
        s = bytes(rand_seq(rng, k)).decode()
        if s not in seen:
            seen.add(s)
            out.append(s)
    return out


def gen_vote(manifest, seed=77):
    """F2: label sets exercising every branch of the vote with planted hit multisets."""
    rng = np.random.default_rng(seed)
    k = 32
    labels = []

    def full(kd, p, c, o, f, g, s, t):
        return "k__%s;p__%s;c__%s;o__%s;f__%s;g__%s;s__%s;t__%s" % (kd, p, c, o, f, g, s, t)

    # three kingdoms, a bushy clade, empty ranks at various depths, truncated (interpolated) ancestors
    for kd in ("Bacteria", "Archaea", "Viruses"):
        for g in range(3):
            for s in range(3):
                for t in range(2):
                    labels.append(full(kd, "P1", "C1", "O1", "F1", "G%d" % g, "G%d_s%d" % (g, s), "G%d_s%d_t%d" % (g, s, t)))
                labels.append(full(kd, "P1", "C1", "O1", "F1", "G%d" % g, "G%d_s%d" % (g, s), ""))          # empty t__
            labels.append(full(kd, "P1", "C1", "O1", "F1", "G%d" % g, "", ""))                               # empty s__;t__
        labels.append("k__%s;p__P1;c__C1;o__;f__;g__;s__;t__" % kd)                                          # empty from o__ down
        labels.append("k__%s;p__P2;c__C9;o__O9;f__F9;g__G9;s__G9_s;t__G9_s_t" % kd)
        # interpolated ancestors (what BUILD_GG appends on k-mer collisions: truncated at a rank)
        labels.append("k__%s;p__P1" % kd)
        labels.append("k__%s;p__P1;c__C1" % kd)
        labels.append("k__%s;p__P1;c__C1;o__O1" % kd)
        labels.append("k__%s;p__P1;c__C1;o__O1;f__F1" % kd)
        for g in range(3):
            labels.append("k__%s;p__P1;c__C1;o__O1;f__F1;g__G%d" % (kd, g))
            labels.append("k__%s;p__P1;c__C1;o__O1;f__F1;g__G%d;s__G%d_s0" % (kd, g, g))
    # prefix-of-token neighbours (td stops inside a token)
    labels += ["k__Bacteria;p__P1;c__C1;o__O1;f__F1;g__G1x;s__a;t__b",
               "k__Bacteria;p__P1;c__C1;o__O1;f__F1;g__G10;s__a;t__b",
               "k__Bact;p__P1", "k__Bacteria_X;p__P1;c__C1;o__O1;f__F1;g__G1;s__q;t__r"]
    labels = list(dict.fromkeys(labels))
    order = rng.permutation(len(labels))
    labels = [labels[i] for i in order]                     # file order != lexical order
    per = 6                                                 # k-mers per label
    kms = unique_kmers(rng, per * len(labels), k)
    hi, lo = ctrfile.encode_kmers(kms)
    ix = np.repeat(np.arange(len(labels), dtype=np.uint32), per)
    srt = np.argsort(lo, kind="stable")
    kms_by_label = [[kms[l * per + j] for j in range(per)] for l in range(len(labels))]
    lab_index = {s: i for i, s in enumerate(labels)}
    reads = []

    def plant(name, picks):
        # picks: list of (label_index, copies); non-overlapping k-mers joined by a 'N' so that no
        # spanning window is ever valid: the hit multiset is exactly what we plant
        parts = []
        for li, copies in picks:
            for c in range(copies):
                parts.append(kms_by_label[li][c % per])
        reads.append((name, np.frombuffer("N".join(parts).encode(), dtype=np.uint8)))

    L = lab_index
    B = "Bacteria"
    t000 = L[full(B, "P1", "C1", "O1", "F1", "G0", "G0_s0", "G0_s0_t0")]
    t001 = L[full(B, "P1", "C1", "O1", "F1", "G0", "G0_s0", "G0_s0_t1")]
    t00e = L[full(B, "P1", "C1", "O1", "F1", "G0", "G0_s0", "")]
    t010 = L[full(B, "P1", "C1", "O1", "F1", "G0", "G0_s1", "G0_s1_t0")]
    t100 = L[full(B, "P1", "C1", "O1", "F1", "G1", "G1_s0", "G1_s0_t0")]
    g0e = L[full(B, "P1", "C1", "O1", "F1", "G0", "", "")]
    anc_p = L["k__Bacteria;p__P1"]
    anc_f = L["k__Bacteria;p__P1;c__C1;o__O1;f__F1"]
    anc_g0 = L["k__Bacteria;p__P1;c__C1;o__O1;f__F1;g__G0"]
    anc_s = L["k__Bacteria;p__P1;c__C1;o__O1;f__F1;g__G0;s__G0_s0"]
    arch = L[full("Archaea", "P1", "C1", "O1", "F1", "G0", "G0_s0", "G0_s0_t0")]
    vir = L[full("Viruses", "P1", "C1", "O1", "F1", "G2", "G2_s2", "G2_s2_t1")]
    oe = L["k__Bacteria;p__P1;c__C1;o__;f__;g__;s__;t__"]
    p2 = L["k__Bacteria;p__P2;c__C9;o__O9;f__F9;g__G9;s__G9_s;t__G9_s_t"]
    hand = [
        ("one_hit", [(t000, 1)]),
        ("one_label_many", [(t000, 5)]),
        ("full_label_wins", [(t000, 9), (t001, 1)]),
        ("tie_strains", [(t000, 3), (t001, 3)]),
        ("species_level", [(t000, 3), (t001, 3), (t00e, 2)]),
        ("genus_level", [(t000, 2), (t010, 2), (g0e, 1)]),
        ("family_level", [(t000, 2), (t100, 2)]),
        ("with_ancestors", [(t000, 4), (anc_p, 2), (anc_f, 1), (anc_g0, 1), (anc_s, 1)]),
        ("ancestors_only", [(anc_p, 2), (anc_f, 2)]),
        ("ancestor_majority", [(anc_p, 6), (t000, 1), (t100, 1)]),
        ("cross_kingdom_even", [(t000, 3), (arch, 3)]),
        ("cross_kingdom_major", [(t000, 7), (arch, 1)]),
        ("three_kingdoms", [(t000, 2), (arch, 2), (vir, 2)]),
        ("empty_order_down", [(oe, 3), (t000, 3)]),
        ("empty_order_major", [(oe, 6), (p2, 1)]),
        ("p1_vs_p2", [(t000, 3), (p2, 1)]),
        ("p1_vs_p2_even", [(t000, 2), (p2, 2)]),
        ("late_group_wins", [(arch, 1), (t000, 1), (vir, 6)]),
        ("token_prefix_a", [(L["k__Bacteria;p__P1;c__C1;o__O1;f__F1;g__G1x;s__a;t__b"], 2),
                            (L["k__Bacteria;p__P1;c__C1;o__O1;f__F1;g__G10;s__a;t__b"], 2), (t100, 2)]),
        ("token_prefix_b", [(L["k__Bact;p__P1"], 2), (t000, 2)]),
        ("token_prefix_c", [(L["k__Bacteria_X;p__P1;c__C1;o__O1;f__F1;g__G1;s__q;t__r"], 3), (t000, 3), (anc_p, 1)]),
    ]
    for nm, picks in hand:
        plant(nm, picks)
    # random multisets: 2..14 hits over related / unrelated labels
    nlab = len(labels)
    srt_labels = sorted(range(nlab), key=lambda i: labels[i])
    for i in range(4000):
        npk = int(rng.integers(1, 7))
        if rng.random() < 0.7:                                  # a lexical neighbourhood
            c = int(rng.integers(0, nlab))
            cand = [srt_labels[(c + d) % nlab] for d in range(-4, 5)]
        else:
            cand = list(range(nlab))
        picks = [(cand[int(rng.integers(0, len(cand)))], int(rng.integers(1, 5))) for _ in range(npk)]
        plant("rnd%d" % i, picks)
    with tempfile.TemporaryDirectory() as td:
        ctr = os.path.join(td, "vote.ctr")
        ctrfile.write_ctr(ctr, 8, 2, hi[srt], lo[srt], ix[srt], labels)
        save_db_fixture("vote", ctr, manifest)
        fa = os.path.join(td, "reads.fa")
        write_fasta(fa, reads)
        gz_write(os.path.join(HERE, "vote_reads.fa.gz"), open(fa, "rb").read())
        code, out, so, se = ref_search("xtree-searchGG", ctr, fa, 0)
        assert code == 0
        gz_write(os.path.join(HERE, "vote_out.txt.gz"), out)
        manifest["vote_out_lines"] = out.count(b"\n")


def gen_kat(manifest, seed=99):
    """F3: lookup known-answer DB written directly, incl. the first-bin quirk layout."""
    rng = np.random.default_rng(seed)
    k = 32
    labels = ["k__A;p__L%d" % i for i in range(40)]
    words = set()
    # bins with 1,2,3,...,40 records at scattered prefixes; min and max suffix in some bins
    prefixes = sorted(int(x) for x in rng.choice(1 << 24, size=60, replace=False))
    prefixes[0] = max(prefixes[0], 5)
    sizes = list(range(1, 41)) + [1, 1, 2, 64, 100, 257] + [3] * 14
    for pfx, sz in zip(prefixes, sizes):
        sufs = set(int(x) for x in rng.integers(0, 1 << 40, size=sz))
        if sz >= 2:
            sufs.add(0)
            sufs.add((1 << 40) - 1)
        for s in sufs:
            words.add((pfx << 40) | s)
    # dense neighbours: consecutive suffixes and consecutive prefixes
    base = (prefixes[10] << 40) | 0x1234567800
    for d in range(6):
        words.add(base + d)
    words.add((0xFFFFFF << 40) | 0xFFFFFFFFFF)              # very last possible word
    words = np.array(sorted(words), dtype=np.uint64)
    ixs = rng.integers(0, len(labels), size=len(words)).astype(np.uint32)

    def mk_reads(wlist):
        reads = []
        n = 0
        for w in wlist:
            s = ctrfile.decode_kmer(0, int(w), k)
            reads.append(("hit%d" % n, np.frombuffer(s.encode(), dtype=np.uint8)))
            # neighbours: +1 / -1 in the suffix (mostly misses), and a flipped prefix bit
            for delta in (1, -1, 1 << 40, 1 << 39):
                v = (int(w) + delta) & 0xFFFFFFFFFFFFFFFF
                reads.append(("nbr%d_%d" % (n, delta), np.frombuffer(ctrfile.decode_kmer(0, v, k).encode(), dtype=np.uint8)))
            n += 1
        return reads

    with tempfile.TemporaryDirectory() as td:
        # (a) exact bin table
        ctr = os.path.join(td, "kat.ctr")
        ctrfile.write_ctr(ctr, 8, 2, np.zeros_like(words), words, ixs, labels)
        save_db_fixture("kat", ctr, manifest)
        reads = mk_reads(words)
        fa = os.path.join(td, "reads.fa")
        write_fasta(fa, reads)
        gz_write(os.path.join(HERE, "kat_reads.fa.gz"), open(fa, "rb").read())
        code, out, so, se = ref_search("xtree-searchGG", ctr, fa, 0)
        assert code == 0
        gz_write(os.path.join(HERE, "kat_out.txt.gz"), out)
        manifest["kat_out_lines"] = out.count(b"\n")
        # (b) first-bin quirk: the first non-empty bin holds exactly one record -> COMPRESS-style table
        qw = np.concatenate([[np.uint64((3 << 40) | 0x7777777777)], words]).astype(np.uint64)
        qw = np.array(sorted(set(int(x) for x in qw)), dtype=np.uint64)
        qix = rng.integers(0, len(labels), size=len(qw)).astype(np.uint32)
        ctrq = os.path.join(td, "katq.ctr")
        ctrfile.write_ctr(ctrq, 8, 2, np.zeros_like(qw), qw, qix, labels, like_compress=True)
        save_db_fixture("katq", ctrq, manifest)
        # probes: every word, plus suffix-equal probes placed in the merged bin
        extra = [np.uint64((int(qw[1]) >> 40 << 40) | 0x7777777777)]
        readsq = mk_reads(list(qw) + extra)
        faq = os.path.join(td, "readsq.fa")
        write_fasta(faq, readsq)
        gz_write(os.path.join(HERE, "katq_reads.fa.gz"), open(faq, "rb").read())
        code, out, so, se = ref_search("xtree-searchGG", ctrq, faq, 0)
        assert code == 0
        gz_write(os.path.join(HERE, "katq_out.txt.gz"), out)
        manifest["katq_out_lines"] = out.count(b"\n")


def gen_irregular(manifest, seed=123):
    """F3 extras: (a) `katq2` -- COMPRESS' first-bin quirk where the stray record is LARGER than the records it is
    merged in front of (a bin that is not ascending: only the reference's exact probe order gives its answers);
    (b) `generic` -- a bin table that is not monotone (never written by COMPRESS, but the reference trusts the
    table verbatim, itree.c:724-728)."""
    rng = np.random.default_rng(seed)
    k = 32
    labels = ["k__A;p__L%d" % i for i in range(25)]

    def probes(wlist, extra=()):
        reads = []
        for n, w in enumerate(list(wlist) + list(extra)):
            for tag, delta in (("w", 0), ("p", 1), ("m", -1)):
                v = (int(w) + delta) & 0xFFFFFFFFFFFFFFFF
                reads.append(("%s%d" % (tag, n), np.frombuffer(ctrfile.decode_kmer(0, v, k).encode(), dtype=np.uint8)))
        return reads

    with tempfile.TemporaryDirectory() as td:
        # (a) first bin = one record with a huge suffix; next bin = 9 records with small suffixes
        P = 0x123456
        words = [(3 << 40) | 0xFFFFFFFFFF] + [(P << 40) | (0x10 * (j + 1)) for j in range(9)]
        words += [((P + 5) << 40) | int(x) for x in sorted(set(int(v) for v in rng.integers(0, 1 << 40, 40)))]
        words = np.array(sorted(words), dtype=np.uint64)
        ixs = rng.integers(0, len(labels), size=len(words)).astype(np.uint32)
        ctr = os.path.join(td, "katq2.ctr")
        ctrfile.write_ctr(ctr, 8, 2, np.zeros_like(words), words, ixs, labels, like_compress=True)
        d = save_db_fixture("katq2", ctr, manifest)
        assert int(d.binix[P]) == 0 and int(d.binix[P + 1]) == 10       # the stray record sits in front of bin P
        extra = [(P << 40) | 0xFFFFFFFFFF, (P << 40) | 0x5, (P << 40) | 0x95, (3 << 40) | 0x10]
        reads = probes(words, extra)
        fa = os.path.join(td, "r.fa")
        write_fasta(fa, reads)
        gz_write(os.path.join(HERE, "katq2_reads.fa.gz"), open(fa, "rb").read())
        code, out, so, se = ref_search("xtree-searchGG", ctr, fa, 0)
        assert code == 0
        gz_write(os.path.join(HERE, "katq2_out.txt.gz"), out)
        manifest["katq2_out_lines"] = out.count(b"\n")
        # (b) non-monotone table: take the kat DB and pull some interior bin starts back / push them forward
        dk = ctrfile.read_ctr(os.path.join(td, "katq2.ctr"))
        kat = np.load(os.path.join(HERE, "kat_db.npz"))
        lo = kat["suf_lo"]
        # rebuild kat exactly, then damage its table
        idx = kat["bin_chg_idx"].astype(np.int64); val = kat["bin_chg_val"].astype(np.uint64)
        starts = np.concatenate([[0], idx]); vals = np.concatenate([[np.uint64(kat["bin0"])], val]).astype(np.uint64)
        lens = np.diff(np.concatenate([starts, [ctrfile.NUMBINS]]))
        b = np.repeat(vals, lens).copy()
        nz = np.flatnonzero(np.diff(b.astype(np.int64)) > 0)
        for j in (5, 11, 17, 23, 31):                 # bin nz[j]: start pulled back by 3 records (overlaps its predecessors)
            b[nz[j]] = np.uint64(max(0, int(b[nz[j]]) - 3))
        for j in (8, 20):                             # bin nz[j]: start pushed past its end (s > e: empty)
            b[nz[j]] = b[nz[j] + 1] + np.uint64(2)
        rec = ctrfile.pack_records(8, 2, np.zeros_like(lo), lo, kat["ix"])
        g = os.path.join(td, "generic.ctr")
        with open(g, "wb") as f:
            f.write(np.array([8, 0, 2, len(lo)], dtype="<u8").tobytes())
            f.write(b.astype("<u4").tobytes())
            f.write(rec.tobytes())
            f.write(kat["label_text"].tobytes())
        save_db_fixture("generic", g, manifest)
        gz = gzip.open(os.path.join(HERE, "kat_reads.fa.gz"), "rb").read()
        fa2 = os.path.join(td, "r2.fa")
        open(fa2, "wb").write(gz)
        code, out, so, se = ref_search("xtree-searchGG", g, fa2, 0)
        assert code == 0, (code, se[-300:])
        gz_write(os.path.join(HERE, "generic_out.txt.gz"), out)
        manifest["generic_out_lines"] = out.count(b"\n")


def gen_compress(manifest, seed=4242):
    """`.ubt` -> `.ctr` by the reference's xtree-compress on hand-made `.ubt` files that hit its
    corners: first bin with ONE record (bin-table quirk), first bin with several, a repeated label line (its count lands
    on the newest label), unsorted words (the table records first NON-ZERO occurrences)."""
    rng = np.random.default_rng(seed)
    cases = {}

    def words_in(prefixes, per):
        out = []
        for p, n in zip(prefixes, per):
            out += [(p << 40) | int(x) for x in rng.integers(0, 1 << 40, n)]
        return np.array(sorted(set(out)), dtype=np.uint64)

    labels = [b"k__A;p__L%d\t%d\n" % (i, 100 + i) for i in range(9)]
    cases["cq_single_first"] = (words_in([7, 9, 300, 70000, (1 << 24) - 1], [1, 5, 3, 40, 2]), b"".join(labels))
    cases["cq_multi_first"] = (words_in([0, 1, 2, 5000], [4, 1, 1, 30]), b"".join(labels))
    cases["cq_dup_labels"] = (words_in([11, 12, 9000], [2, 2, 20]), b"".join(labels[:4]) + labels[1] + labels[5] + b"k__A;p__L0\t77\n" + labels[6])
    w = words_in([21, 22, 23, 40000], [3, 3, 3, 25])
    cases["cq_unsorted"] = (w[rng.permutation(len(w))], b"".join(labels))
    with tempfile.TemporaryDirectory() as td:
        for nm, (w, tail) in cases.items():
            ix = rng.integers(0, 9, size=len(w)).astype(np.uint32)
            ubt = os.path.join(td, nm + ".ubt")
            ctrfile.write_ubt(ubt, 8, 2, np.zeros_like(w), w, ix, tail)
            ctr = os.path.join(td, nm + ".ctr")
            code, so, se = run([os.path.join(REF, "xtree-compress"), ubt, ctr])
            assert code == 0, (nm, code, so[-300:])
            np.savez_compressed(os.path.join(HERE, nm + "_ubt.npz"), lo=w, ix=ix, tail=np.frombuffer(tail, dtype=np.uint8))
            manifest[nm + "_ctr_sha256"] = ctrfile.sha256_file(ctr)
            manifest[nm + "_stdout_tail"] = so.decode().strip().splitlines()[-1]


RANK_BINS = {            # binary -> (slack, sparsity, tolerance): the compile-time knobs of itree.c:952-960
    "xtree-search": (2, 4, 2),
    "xtree-search-s3p8": (3, 8, 2),
    "xtree-search-p1": (2, 1, 2),
    "xtree-search-p2": (2, 2, 2),
    "xtree-search-p32s1t1": (1, 32, 1),
}


def gen_rank(manifest, seed=777):
    """the rank-specific `xtree-search` (itree.c -D SEARCH).  Outputs of the genuine binaries
    (several SLACK / SPARSITY / TOLERANCE_THRESHOLD builds) on
      * the existing toy / k64 / ix32 / vote fixtures, and
      * `rk`: a DENSE direct-written DB (every window of 48 related references) plus planted "register" words --
        what the reference really looks up after a hit (see oracle/utree_oracle.c rank_hits) -- and homopolymer /
        repeat k-mers; reads of 30 bp .. 120 kb in one file, so that the entry a read's vote picks up from an
        earlier read's hit list (itree.c:982) comes from far and near."""
    rng = np.random.default_rng(seed)
    k = 32
    out_names = {}
    # ---- (a) existing fixtures through the rank-specific binaries
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util  # noqa: E402
    for name, binn in (("toy", "xtree-search"), ("toy", "xtree-search-s3p8"), ("k64", "xtree-search-k64"),
                       ("ix32", "xtree-search-ix32"), ("vote", "xtree-search")):
        ctr = util.fixture_ctr(name)
        fa = util.fixture_reads_path(name)
        for rc in (0, 1):
            code, out, so, se = ref_search(binn, ctr, fa, rc)
            assert code == 0, (binn, code, se[-300:])
            tag = "%s_rank%s%s" % (name, binn[len("xtree-search"):].replace("-k64", "").replace("-ix32", ""), "_rc" if rc else "")
            gz_write(os.path.join(HERE, tag + ".txt.gz"), out)
            out_names[tag] = {"db": name, "reads": name, "bin": binn, "rc": rc, "lines": out.count(b"\n")}
    # ---- (b) the dense DB
    labels = ["lab_%02d_%s" % (i, "".join(chr(97 + int(c)) for c in rng.integers(0, 26, 5))) for i in range(48)]
    roots = [rand_seq(rng, 1500) for _ in range(6)]
    refs = [mutate(rng, roots[i % 6], 0.03) for i in range(48)]
    kmers = {}
    for li, s in enumerate(refs):
        b = bytes(s).decode()
        for j in range(len(b) - k + 1):
            kmers.setdefault(b[j:j + k], li)

    def garbage(word, newbases, S):
        """what the reference's register holds d = len(newbases) bases after a hit on `word` (S = k/SPARSITY)"""
        d, E = len(newbases), S - 1
        return (word[d + E:] + "A" * E + newbases) if d + E <= k else ("A" * (k - d) + newbases)

    planted = []
    for n in range(60):                                   # reads whose post-hit register words are in the DB
        li = int(rng.integers(0, 48))
        a = int(rng.integers(0, 1500 - 200))
        r = bytes(refs[li][a:a + 160]).decode()
        w0 = r[:k]
        S = 8
        d = int(rng.integers(S, k))                        # a register word d bases after the first hit
        g1 = garbage(w0, r[k:k + d], S)
        kmers.setdefault(g1, int(rng.integers(0, 48)))
        if n % 2:                                          # and a second one chained from it
            d2 = int(rng.integers(S, k))
            g2 = garbage(g1, r[k + d:k + d + d2], S)
            kmers.setdefault(g2, int(rng.integers(0, 48)))
        planted.append(("plant%d" % n, np.frombuffer(r.encode(), dtype=np.uint8)))
    for hp in ("A" * k, "C" * k, "ACGT" * 8, "AC" * 16, "T" * k):
        kmers.setdefault(hp, int(rng.integers(0, 48)))
    ks = list(kmers)
    hi, lo = ctrfile.encode_kmers(ks)
    ix = np.array([kmers[s] for s in ks], dtype=np.uint32)
    srt = np.argsort(lo, kind="stable")
    reads = []
    comp = {65: 84, 67: 71, 71: 67, 84: 65}

    def draw(n, lo_len, hi_len, tag):
        for i in range(n):
            L = int(rng.integers(lo_len, hi_len))
            parts = []
            while sum(len(p) for p in parts) < L:          # stitched from several references (chimeras for long reads)
                ref = refs[int(rng.integers(0, 48))]
                seg = int(rng.integers(20, 1200))
                a = int(rng.integers(0, 1500 - 20))
                parts.append(mutate(rng, ref[a:a + seg], float(rng.choice([0.0, 0.005, 0.02, 0.3]))))
            s = np.concatenate(parts)[:L].copy()
            q = rng.random()
            if q < 0.1 and L > 4:
                s[rng.integers(0, L, size=max(1, L // 300))] = ord("N")
            elif q < 0.15:
                s = np.frombuffer(bytes(s).lower(), dtype=np.uint8).copy()
            elif q < 0.25:
                s = np.array([comp.get(int(c), 78) for c in s[::-1]], dtype=np.uint8)
            reads.append(("%s%d" % (tag, i), s))

    draw(1500, 30, 400, "s")
    draw(25, 1000, 6000, "m")
    draw(2, 30000, 60000, "l")
    reads += planted[:30]
    draw(1, 119000, 120000, "x")
    draw(1200, 30, 700, "t")
    reads += planted[30:]
    for nm, s in (("polyA", "A" * 300), ("polyC", "C" * 90 + "A" * 90), ("acgt", "ACGT" * 60), ("ac", "AC" * 100 + "N" + "T" * 70)):
        reads.append((nm, np.frombuffer(s.encode(), dtype=np.uint8)))
    draw(300, 30, 300, "u")
    with tempfile.TemporaryDirectory() as td:
        ctr = os.path.join(td, "rk.ctr")
        ctrfile.write_ctr(ctr, 8, 2, hi[srt], lo[srt], ix[srt], labels)
        save_db_fixture("rk", ctr, manifest)
        fa = os.path.join(td, "reads.fa")
        write_fasta(fa, reads)
        gz_write(os.path.join(HERE, "rk_reads.fa.gz"), open(fa, "rb").read())
        for binn in RANK_BINS:
            for rc in ((0, 1) if binn in ("xtree-search", "xtree-search-p2") else (0,)):
                code, out, so, se = ref_search(binn, ctr, fa, rc)
                assert code == 0, (binn, code, se[-300:])
                tag = "rk_rank%s%s" % (binn[len("xtree-search"):], "_rc" if rc else "")
                gz_write(os.path.join(HERE, tag + ".txt.gz"), out)
                out_names[tag] = {"db": "rk", "reads": "rk", "bin": binn, "rc": rc, "lines": out.count(b"\n")}
    for v in out_names.values():
        v["params"] = list(RANK_BINS[v["bin"].replace("-k64", "").replace("-ix32", "")])
    manifest["rank_outputs"] = out_names


def gen_build(manifest, seed=2025):
    """database BUILD.  Inputs (FASTA + map) are committed; expected = SHA-256 of the `.ubt` and of the
    `[.gg].log` the genuine `utree-build` / `utree-buildGG` (and the PACKSIZE=64 / IXTYPE=uint32_t builds) write."""
    rng = np.random.default_rng(seed)
    sets = {}
    # (a) related references: k-mer collisions at every rank, several references per leaf label
    leaves = make_taxonomy(rng, 60)
    leaves = leaves + [leaves[int(i)] for i in rng.integers(0, 60, 40)]            # repeated labels
    refs = make_refs(rng, leaves, 900, per_level_rate=0.004)
    fa = b"".join(b">" + n.encode() + b"\n" + bytes(sq) + b"\n" for n, _, sq in refs)
    mp = b"".join(n.encode() + b"\t" + t.encode() + b"\n" for n, t, _ in refs[::-1])   # map order != FASTA order
    sets["rel"] = (fa, mp)
    # (b) corners
    def rs(n):
        return bytes(rand_seq(rng, n))
    A = "k__K;p__P;c__C;o__O;f__F;g__G;s__S1;t__T1"
    B = "k__K;p__P;c__C;o__O;f__F;g__G;s__S2;t__T2"
    Cc = "k__K;p__P2;c__C;o__O;f__F;g__G;s__S9;t__T9"
    S, S2, S3, S4 = rs(300), rs(120), rs(150), rs(200)
    items = [
        ("r1 first", A, S), ("r2 same seq other species", B, S), ("r3 first label again", A, S), ("r4 other phylum", Cc, S),
        ("r5", "k__K;p__P", S2), ("r6", "k__K;p__P;c__X", S2),
        ("r7 lower", A, rs(100).lower()), ("r8 n", B, S4[:70] + b"N" + S4[71:140] + b"nn" + S4[142:]),
        ("r9 short", A, rs(31)), ("r10 exact", B, rs(32)), ("r11 crlf", Cc, rs(90) + b"\r"),
        ("r12", A, rs(80)),
        ("r13", "a;b;c;d", S3), ("r14", "a;b;c;e", S3), ("r15", "a;b;c", S3), ("r16", "a;b;c;d", S3), ("r17", "a;b;x", S3),
        ("r18 iupac", B, S4[:50] + b"RYK" + S4[53:120]),
        ("r19 polyA", Cc, b"A" * 80), ("r20 polyA2", "k__K;p__P2;c__C;o__O2", b"A" * 70 + b"C" * 40),
    ]
    fa = b"".join(b">" + n.encode() + b"\n" + sq + b"\n" for n, _, sq in items)
    mp = b"".join(n.encode() + b"\t" + t.encode() + b"\n" for n, t, _ in sorted(items, key=lambda x: x[0][::-1]))
    sets["corner"] = (fa, mp)
    sets["err_missing"] = (b">zz not in map\n" + rs(50) + b"\n", b"other\tk__K;p__P\n")
    sets["err_map_no_newline"] = (b">a\n" + rs(50) + b"\n", b"a\tk__K;p__P")
    sets["err_no_kmers"] = (b">a\n" + rs(20) + b"\n", b"a\tk__K;p__P\n")
    sets["err_missing_seq"] = (b">a\n" + rs(50) + b"\n>b\n", b"a\tk__K;p__P\nb\tk__K;p__Q\n")
    runs = [("rel", "utree-buildGG", 8, 2, c, 1) for c in (0, 1, 2, 4)] + [
        ("rel", "utree-build", 8, 2, 1, 0), ("rel", "utree-buildGG-k64", 16, 2, 1, 1), ("rel", "utree-buildGG-ix32", 8, 4, 2, 1),
        ("corner", "utree-buildGG", 8, 2, 0, 1), ("corner", "utree-buildGG", 8, 2, 1, 1), ("corner", "utree-build", 8, 2, 0, 0),
        ("corner", "utree-buildGG-k64", 16, 2, 0, 1),
        ("err_missing", "utree-buildGG", 8, 2, 0, 1), ("err_map_no_newline", "utree-buildGG", 8, 2, 0, 1),
        ("err_no_kmers", "utree-buildGG", 8, 2, 0, 1), ("err_missing_seq", "utree-buildGG", 8, 2, 0, 1)]
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for nm, (fa, mp) in sets.items():
            gz_write(os.path.join(HERE, "build_%s.fa.gz" % nm), fa)
            gz_write(os.path.join(HERE, "build_%s.map.gz" % nm), mp)
            open(os.path.join(td, nm + ".fa"), "wb").write(fa)
            open(os.path.join(td, nm + ".map"), "wb").write(mp)
        for nm, binn, W, I, cl, gg in runs:
            ubt = os.path.join(td, "o.ubt")
            log = ubt + (".gg.log" if gg else ".log")
            for f in (ubt, log):
                if os.path.exists(f):
                    os.remove(f)
            code, so, se = run([os.path.join(REF, binn), os.path.join(td, nm + ".fa"), os.path.join(td, nm + ".map"), ubt, "1", str(cl)])
            tag = "%s_%s_c%d" % (nm, binn.replace("utree-", ""), cl)
            rec = {"set": nm, "bin": binn, "W": W, "I": I, "complevel": cl, "gg": gg, "exit": code}
            if code == 0:
                rec["ubt_sha256"] = ctrfile.sha256_file(ubt)
                rec["log_sha256"] = ctrfile.sha256_file(log)
                rec["ubt_bytes"] = os.path.getsize(ubt)
                rec["stdout_tail"] = so.decode(errors="replace").strip().splitlines()[-3:]
            out[tag] = rec
    manifest["build_outputs"] = out


def main():
    if not os.path.exists(os.path.join(REF, "xtree-searchGG")):
        sys.exit("build the reference first: make -C oracle ref")
    if len(sys.argv) > 1 and sys.argv[1] == "compress":       # add the COMPRESS fixtures to an existing golden set
        manifest = json.load(open(os.path.join(HERE, "manifest.json")))
        gen_compress(manifest)
        json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "rank":           # add the rank-specific (`xtree-search`) fixtures
        manifest = json.load(open(os.path.join(HERE, "manifest.json")))
        gen_rank(manifest)
        json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "build":          # add the BUILD / BUILD_GG fixtures
        manifest = json.load(open(os.path.join(HERE, "manifest.json")))
        gen_build(manifest)
        json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "k16":            # add the PACKSIZE=16 fixture (reference built with -D PACKSIZE=16: oracle/_ref/*-k16)
        manifest = json.load(open(os.path.join(HERE, "manifest.json")))
        gen_toy(manifest, "k16", "-k16", "xtree-searchGG-k16", 120, 900, 3000, 100, seed=16, complevel=0)
        json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "k64ix32":        # add the PACKSIZE=64 + IXTYPE=uint32_t fixture (oracle/_ref/*-k64-ix32)
        manifest = json.load(open(os.path.join(HERE, "manifest.json")))
        gen_toy(manifest, "k64ix32", "-k64-ix32", "xtree-searchGG-k64-ix32", 150, 1200, 3000, 150, seed=6432)
        json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "irregular":      # add the irregular-bin fixtures to an existing golden set
        manifest = json.load(open(os.path.join(HERE, "manifest.json")))
        gen_irregular(manifest)
        json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
        return
    manifest = {}
    gen_toy(manifest, "toy", "", "xtree-searchGG", 1000, 1000, 10000, 100, seed=20240807)
    gen_toy(manifest, "k64", "-k64", "xtree-searchGG-k64", 150, 1200, 3000, 150, seed=64)
    gen_toy(manifest, "ix32", "-ix32", "xtree-searchGG-ix32", 150, 1200, 3000, 120, seed=32)
    gen_toy(manifest, "k16", "-k16", "xtree-searchGG-k16", 120, 900, 3000, 100, seed=16, complevel=0)
    gen_toy(manifest, "k64ix32", "-k64-ix32", "xtree-searchGG-k64-ix32", 150, 1200, 3000, 150, seed=6432)
    gen_vote(manifest)
    gen_kat(manifest)
    gen_irregular(manifest)
    gen_compress(manifest)
    gen_rank(manifest)
    gen_build(manifest)
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(manifest, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
