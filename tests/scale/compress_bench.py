#!/usr/bin/env python3
"""Measure `.ubt` -> `.ctr`: our xtree-compress (GPU) vs the genuine reference's, same file, outputs compared."""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from utree_amd import ctrfile, lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
d = "/dev/shm/utree_cmp"; os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(1)
lo = np.unique(rng.integers(0, 1 << 63, size=int(n * 1.02), dtype=np.uint64))[:n]
ix = rng.integers(0, 1000, size=len(lo)).astype(np.uint32)
cnt = np.bincount(ix, minlength=1000)
tail = b"".join(b"k__R;p__%d\t%d\n" % (i, c) for i, c in enumerate(cnt))
ubt = d + "/x.ubt"; ctrfile.write_ubt(ubt, 8, 2, np.zeros_like(lo), lo, ix, tail)
out = {"nodes": len(lo), "ubt_bytes": os.path.getsize(ubt)}
def run(cmd):
    t = time.time(); r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE); return time.time() - t, r
t, r = run([lib.COMPRESS_CLI_PATH, ubt, d + "/ours.ctr"]); out["ours_seconds"] = t; out["ours_stderr"] = r.stderr.decode().strip().splitlines()[-1:]
ref = os.path.join(os.path.dirname(lib.SO_PATH), "..", "oracle", "_ref", "xtree-compress")
if os.path.exists(ref):
    t, r = run([ref, ubt, d + "/ref.ctr"]); out["reference_seconds"] = t
    out["identical"] = ctrfile.sha256_file(d + "/ours.ctr") == ctrfile.sha256_file(d + "/ref.ctr")
print(json.dumps(out))
for f in os.listdir(d): os.remove(os.path.join(d, f))
