#!/usr/bin/env python3
"""Fold one profiles/run_prof.sh result (gpurun_out/prof_<tag>/counters.json, written by summarize.py on the GPU box) into
profiles/traffic.json, the file bench.py reads `roofline.traffic` and the measured fractions from.

    python profiles/make_traffic.py gpurun_out/prof_<tag> profiles/r02/prof_<tag>.txt

The entry records the kernel's full template signature and the hash of the kernel sources it was built from; bench.py uses
the entry only when both match the library it is running (otherwise `traffic` is null and `profile.why_not` says why).
HBM bytes: FETCH_SIZE (KB) x 2 -- MI355X_MICROARCH.md's gfx950 correction: the counter tallies a 128-byte line as 64 bytes --
plus WRITE_SIZE (KB), each from its own --pmc pass, averaged over the launches of the dominant kernel."""
import json
import os
import sys

d, summary = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
c = json.load(open(os.path.join(d, "counters.json")))
line = c["bench_line"]
sig = line["roofline"]["kernel"]
name = [k for k in c["counters"] if sig in k or (sig.endswith(">") and k.startswith(sig[:-1] + ","))]   # (a signature recorded with fewer template arguments)
assert len(name) == 1, (sig, list(c["counters"]))
cn = c["counters"][name[0]]
st = [v for k, v in c["kernel_stats"].items() if sig in k or (sig.endswith(">") and (sig[:-1] + ",") in k)]
assert len(st) == 1
a = dict(x.split("=") for x in [])
ap = line["config"]["workload"]
import re
m = re.search(r"\(k=(\d+),", ap)
kmer = int(m.group(1))
rc = int(re.search(r"RC=(\d)", ap).group(1))
nodes = int(float(re.search(r": ([0-9.e+]+)-node", ap).group(1)))
# exact figures from the args (the workload string rounds the node count)
args = c.get("bench_args", "").split()
def arg(flag, default):
    return type(default)(args[args.index(flag) + 1]) if flag in args else default
nodes = arg("--nodes", 1_217_000_000)
reads = arg("--batch-reads", int(line["roofline"].get("reads_per_launch", 4_000_000)))      # (bench.py's default differs by workload: take it from the line)
length = arg("--read-len", 150)
key = "nodes=%d,reads=%d,len=%d,k=%d,rc=%d" % (nodes, reads, length, kmer, rc)
if arg("--len-dist", "fixed") != "fixed":
    key += ",dist=%s" % arg("--len-dist", "fixed")
if arg("--workload", "config") == "hit_dense":
    nodes = int(float(re.search(r": ([0-9.e+]+)-node", ap).group(1)))          # (the builder decides the node count: take it from the line)
    m_nodes = re.search(r"nodes_exact=(\d+)", ap)
    nodes = int(m_nodes.group(1)) if m_nodes else nodes
    key = "nodes=%d,reads=%d,len=%d,k=%d,rc=%d,workload=hit_dense" % (nodes, reads, length, kmer, rc)
bb = int(line["roofline"].get("model", {}).get("bucket_bytes", 64))
if bb != 64:
    key += ",bucket=%d" % bb
entry = {
    "kernel": name[0],
    "kernel_source_sha256": c["kernel_source_sha256"],
    "avg_launch_ms": st[0]["avg_ns"] / 1e6,
    "hbm_bytes_per_launch": cn["FETCH_SIZE"] * 1024 * 2 + cn["WRITE_SIZE"] * 1024,
    "FETCH_SIZE_KB_per_launch": cn["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": cn["WRITE_SIZE"],
    "TCC_MISS_sum_per_launch": cn.get("TCC_MISS_sum"), "TCC_HIT_sum_per_launch": cn.get("TCC_HIT_sum"),
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/run_prof.sh), averaged over the launches of the "
              "kernel; read side x2 per MI355X_MICROARCH.md (FETCH_SIZE tallies a 128-B line as 64 B on gfx950; confirmed for random-line "
              "reads by profiles/r01/membench_random_lines.txt: TCC_MISS x 128 B); WRITE_SIZE as is; avg_launch_ms from the --kernel-trace --stats pass",
    "source": summary,
}
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY",
          "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_WR"):
    if k in cn:
        entry[k + "_per_launch"] = cn[k]
tp = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tp)) if os.path.exists(tp) else {}
tj = {k: v for k, v in tj.items() if "k=" in k}            # entries of the older key format are stale by construction
tj[key] = entry
json.dump(tj, open(tp, "w"), indent=1)
print(key, json.dumps(entry, indent=1))
