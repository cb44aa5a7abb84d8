#!/usr/bin/env python3
"""The whole chain on one box, ours against the genuine reference, every link compared:
     references.fa + labels.map --utree-buildGG--> .ubt --xtree-compress--> .ctr --xtree-searchGG / xtree-search--> classifications
References: mutated copies of a few root sequences with GG-style labels (k-mers collide at every rank); reads: 150 bp
slices of the references with 1 % substitutions, some reverse-complemented.  usage: chain_check.py [n_refs] [ref_len] [complevel] [n_reads]
VARIANT=k64|ix32 selects the reference's -D PACKSIZE=64 / -D IXTYPE=uint32_t builds."""
import hashlib, json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from utree_amd import ctrfile, lib
n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ref_len = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
cl = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n_reads = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
variant = os.environ.get("VARIANT", "")
sfx = "-" + variant if variant else ""
env = dict(os.environ)
if variant == "k64": env["UTREE_PACKSIZE"] = "64"
if variant == "ix32": env["UTREE_IXTYPE"] = "32"
d = "/dev/shm/utree_chain"; os.makedirs(d, exist_ok=True)
REF = os.path.join(os.path.dirname(lib.SO_PATH), "..", "oracle", "_ref")
rng = np.random.default_rng(11)
n_roots = max(4, n_refs // 25)
roots = [rng.integers(0, 4, ref_len, dtype=np.uint8) for _ in range(n_roots)]
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
ranks = "kpcofgst"
refs = []
with open(d + "/refs.fa", "wb") as f, open(d + "/refs.map", "wb") as g:
    for i in range(n_refs):
        r = i % n_roots
        s = roots[r].copy()
        m = rng.random(ref_len) < 0.02
        s[m] = rng.integers(0, 4, int(m.sum()), dtype=np.uint8)
        if i < 64: refs.append(s)
        f.write(b">ref%06d\n" % i); f.write(ACGT[s].tobytes()); f.write(b"\n")
        path = [r % 2, r % 3, r % 5, r % 7, r % 11, r, i % 9, i]
        g.write(b"ref%06d\t" % i + ";".join("%s__%d" % (ranks[k], path[k]) for k in range(8)).encode() + b"\n")
with open(d + "/reads.fa", "wb") as f:
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    for a in range(0, n_reads, 100000):
        n = min(100000, n_reads - a)
        which = rng.integers(0, len(refs), n); pos = rng.integers(0, ref_len - 150, n)
        for j in range(n):
            s = refs[which[j]][pos[j]:pos[j] + 150].copy()
            mm = rng.random(150) < 0.01
            s[mm] = rng.integers(0, 4, int(mm.sum()), dtype=np.uint8)
            if j & 3 == 0: s = comp[s[::-1]]
            f.write(b">q%d\n" % (a + j)); f.write(ACGT[s].tobytes()); f.write(b"\n")
out = {"n_refs": n_refs, "ref_len": ref_len, "complevel": cl, "n_reads": n_reads, "variant": variant or "default"}
def run(cmd, e=None):
    t = time.time(); r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e); return time.time() - t, r
def sorted_sha(p):
    l = open(p, "rb").read().split(b"\n"); l.sort(); return hashlib.sha256(b"\n".join(l)).hexdigest()
T = {}
T["build_ours"], r = run([lib.BUILD_GG_CLI_PATH, d + "/refs.fa", d + "/refs.map", d + "/ours.ubt", "0", str(cl)], env); assert r.returncode == 0, r.stderr
T["build_ref"], r = run([REF + "/utree-buildGG" + sfx, d + "/refs.fa", d + "/refs.map", d + "/ref.ubt", "0", str(cl)]); assert r.returncode == 0
out["ubt_identical"] = ctrfile.sha256_file(d + "/ours.ubt") == ctrfile.sha256_file(d + "/ref.ubt")
out["build_stdout"] = r.stdout.decode().strip().splitlines()[-2:-1]
T["compress_ours"], r = run([lib.COMPRESS_CLI_PATH, d + "/ours.ubt", d + "/ours.ctr"]); assert r.returncode == 0, r.stderr
T["compress_ref"], r = run([REF + "/xtree-compress" + sfx, d + "/ref.ubt", d + "/ref.ctr"]); assert r.returncode == 0
out["ctr_identical"] = ctrfile.sha256_file(d + "/ours.ctr") == ctrfile.sha256_file(d + "/ref.ctr")
for rc in ([], ["RC"]):
    tag = "gg" + ("_rc" if rc else "")
    T["search_ours_" + tag], r = run([lib.CLI_PATH, d + "/ours.ctr", d + "/reads.fa", d + "/ours.txt", "16"] + rc); assert r.returncode == 0, r.stderr
    T["search_ref_" + tag], r = run([REF + "/xtree-searchGG" + sfx, d + "/ref.ctr", d + "/reads.fa", d + "/ref.txt", "16"] + rc); assert r.returncode == 0
    out["search_%s_identical" % tag] = sorted_sha(d + "/ours.txt") == sorted_sha(d + "/ref.txt")
    out["search_%s_lines" % tag] = open(d + "/ours.txt", "rb").read().count(b"\n")
T["rank_ours"], r = run([lib.RANK_CLI_PATH, d + "/ours.ctr", d + "/reads.fa", d + "/ours.txt", "16", "RC"]); assert r.returncode == 0, r.stderr
T["rank_ref"], r = run([REF + "/xtree-search" + sfx, d + "/ref.ctr", d + "/reads.fa", d + "/ref.txt", "1", "RC"]); assert r.returncode == 0
out["rank_rc_files_identical"] = ctrfile.sha256_file(d + "/ours.txt") == ctrfile.sha256_file(d + "/ref.txt")
out["rank_rc_lines"] = open(d + "/ours.txt", "rb").read().count(b"\n")
out["seconds"] = {k: round(v, 3) for k, v in T.items()}
print(json.dumps(out, indent=1))
for f in os.listdir(d): os.remove(os.path.join(d, f))
