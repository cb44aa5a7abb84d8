#!/usr/bin/env python3
"""Measure the database BUILD: our utree-buildGG (GPU) vs the genuine reference's on the same FASTA + map; the two `.ubt`
files and the two `.gg.log` files must be identical.  Synthetic references: mutated copies of a few root sequences with
GG-style labels, so k-mers collide at every rank.  usage: build_bench.py [n_refs] [ref_len] [complevel]"""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from utree_amd import ctrfile, lib
n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ref_len = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
cl = int(sys.argv[3]) if len(sys.argv) > 3 else 2
d = "/dev/shm/utree_bld"; os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(7)
n_roots = max(4, n_refs // 25)
roots = [rng.integers(0, 4, ref_len, dtype=np.uint8) for _ in range(n_roots)]
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
ranks = "kpcofgst"
fa = d + "/refs.fa"; mp = d + "/refs.map"
with open(fa, "wb") as f, open(mp, "wb") as g:
    for i in range(n_refs):
        r = i % n_roots
        s = roots[r].copy()
        m = rng.random(ref_len) < 0.02
        s[m] = rng.integers(0, 4, int(m.sum()), dtype=np.uint8)
        f.write(b">ref%06d\n" % i); f.write(ACGT[s].tobytes()); f.write(b"\n")
        # root r = a genus; references of one root = species / strains below it
        path = [r % 2, r % 3, r % 5, r % 7, r % 11, r, i % 9, i]
        g.write(b"ref%06d\t" % i + ";".join("%s__%d" % (ranks[k], path[k]) for k in range(8)).encode() + b"\n")
out = {"n_refs": n_refs, "ref_len": ref_len, "complevel": cl, "fasta_bytes": os.path.getsize(fa)}
def run(cmd):
    t = time.time(); r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE); return time.time() - t, r
variant = os.environ.get("VARIANT", "")                   # "", "k64" or "ix32": the reference's -D PACKSIZE=64 / -D IXTYPE=uint32_t builds
env = dict(os.environ)
if variant == "k64": env["UTREE_PACKSIZE"] = "64"
if variant == "ix32": env["UTREE_IXTYPE"] = "32"
out["variant"] = variant or "default"
t0_ = time.time(); r = subprocess.run([lib.BUILD_GG_CLI_PATH, fa, mp, d + "/ours.ubt", "0", str(cl)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env); t = time.time() - t0_
out["ours_seconds"] = t; out["ours_exit"] = r.returncode
out["ours_stderr"] = r.stderr.decode().strip().splitlines()[-1:]; out["ours_stdout"] = r.stdout.decode().strip().splitlines()[-2:]
ref = os.path.join(os.path.dirname(lib.SO_PATH), "..", "oracle", "_ref", "utree-buildGG" + ("-" + variant if variant else ""))
if os.path.exists(ref) and not os.environ.get("SKIP_REFERENCE"):
    t, r = run([ref, fa, mp, d + "/ref.ubt", "0", str(cl)]); out["reference_seconds"] = t; out["reference_exit"] = r.returncode
    out["reference_stdout"] = r.stdout.decode().strip().splitlines()[-2:]
    out["ubt_identical"] = ctrfile.sha256_file(d + "/ours.ubt") == ctrfile.sha256_file(d + "/ref.ubt")
    out["log_identical"] = ctrfile.sha256_file(d + "/ours.ubt.gg.log") == ctrfile.sha256_file(d + "/ref.ubt.gg.log")
    out["speedup"] = out["reference_seconds"] / out["ours_seconds"]
print(json.dumps(out, indent=1))
if not os.environ.get("KEEP_FILES"):
    for f in os.listdir(d): os.remove(os.path.join(d, f))
