#!/usr/bin/env python3
"""Multi-chunk check of the opt-in input formats: the same 3 M reads as two-line FASTA, FASTQ, multi-line FASTA (61 columns)
and gzip FASTQ through the command line; all four outputs must be identical (records cross the 96 MiB chunk boundaries)."""
import gzip, hashlib, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from utree_amd import lib
d = "/dev/shm/utree_e2e"
os.environ["KEEP_FILES"] = "1"
subprocess.check_call([sys.executable, os.path.join(os.path.dirname(__file__), "e2e_scale.py"), "--nodes", "100000000", "--reads", "3000000",
                       "--skip-reference"], stdout=subprocess.DEVNULL)
fa = open(d + "/reads.fa", "rb").read().split(b"\n")
names, seqs = fa[0:-1:2], fa[1::2]
with open(d + "/reads.fq", "wb") as f:
    for n, s in zip(names, seqs):
        f.write(b"@" + n[1:] + b" extra\n" + s + b"\n+\n" + b"F" * len(s) + b"\n")
with open(d + "/reads_ml.fa", "wb") as f:
    for n, s in zip(names, seqs):
        f.write(n + b" extra\n" + b"".join(s[a:a + 61] + b"\n" for a in range(0, len(s), 61)))
with open(d + "/reads.fq", "rb") as f, gzip.open(d + "/reads.fq.gz", "wb", compresslevel=1) as g:
    while True:
        b = f.read(1 << 24)
        if not b: break
        g.write(b)
out = {"reads": len(names), "bytes": {k: os.path.getsize(d + "/" + k) for k in ("reads.fa", "reads.fq", "reads_ml.fa", "reads.fq.gz")}}
sha = {}
for k, env in (("reads.fa", None), ("reads.fq", "auto"), ("reads_ml.fa", "auto"), ("reads.fq.gz", "fastq"), ("reads.fa", "fasta")):
    e = dict(os.environ)
    if env: e["UTREE_INPUT"] = env
    r = subprocess.run([lib.CLI_PATH, d + "/synth.ctr", d + "/" + k, d + "/o.txt", "16"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    tag = k + ("" if not env else " UTREE_INPUT=" + env)
    sha[tag] = (r.returncode, hashlib.sha256(open(d + "/o.txt", "rb").read()).hexdigest(), [l for l in r.stderr.decode().splitlines() if "search" in l][-1:])
out["runs"] = sha
out["all_identical"] = len({v[1] for v in sha.values()}) == 1 and all(v[0] == 0 for v in sha.values())
print(json.dumps(out, indent=1))
for f in os.listdir(d): os.remove(os.path.join(d, f))
