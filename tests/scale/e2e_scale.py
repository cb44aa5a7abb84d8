#!/usr/bin/env python3
"""End-to-end check at scale on the GPU box (run through gpurun):

  1. write the seeded synthetic database (utree_amd.synth) as a real `.ctr` FILE and a FASTA file of reads;
  2. run OUR command line `utree_amd/xtree-searchGG` on them (file in -> file out: disk read, framing, PCIe,
     kernels, formatting, write);
  3. run the GENUINE reference `oracle/_ref/xtree-searchGG` (built by `make -C oracle ref` in the build
     container; the binary travels, its source does not) on the same files with all host threads, and
     once more on an empty FASTA to take its database load time out;
  4. compare: our output == the reference's output as multisets of lines (the reference writes in thread
     completion order), and byte-for-byte against a 1-thread reference run on a prefix of the reads.

Prints one JSON object.  Nothing here is used by the product; it is measurement + parity infrastructure.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(cmd, **kw):
    t0 = time.time()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)
    return r.returncode, time.time() - t0, r.stdout.decode(errors="replace"), r.stderr.decode(errors="replace")


def sorted_sha(path):
    lines = open(path, "rb").read().split(b"\n")
    lines.sort()
    return hashlib.sha256(b"\n".join(lines)).hexdigest(), len(lines) - 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_217_000_000)
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--kmer", type=int, default=32)
    ap.add_argument("--rc", type=int, default=0)
    ap.add_argument("--dir", default="/dev/shm/utree_e2e")
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--prefix-reads", type=int, default=100_000, help="reads of the ordered (1-thread) reference comparison")
    ap.add_argument("--skip-reference", action="store_true")
    ap.add_argument("--rank", action="store_true",
                    help="the rank-specific `xtree-search` pair instead of xtree-searchGG: the reference is sequential there, "
                         "so the two output FILES must be identical byte for byte")
    ap.add_argument("--sweep-threads", default="", help="comma list: extra reference runs at these thread counts")
    args = ap.parse_args()
    import numpy as np
    import torch
    from utree_amd import lib, synth
    os.makedirs(args.dir, exist_ok=True)
    ctr_path = os.path.join(args.dir, "synth.ctr")
    fa_path = os.path.join(args.dir, "reads.fa")
    W = args.kmer // 4
    out = {"nodes": args.nodes, "reads": args.reads, "read_len": args.read_len, "kmer": args.kmer, "rc": args.rc}
    dev = torch.device("cuda:0")
    t0 = time.time()
    sdb = synth.make_db(dev, args.nodes, W=W, keep_raw=True)
    with open(ctr_path, "wb") as f:
        f.write(np.array([W, 0, 2, args.nodes], dtype="<u8").tobytes())
        f.write(sdb.binix.cpu().numpy().tobytes())
        rec = sdb.records.cpu().numpy()
        step = 1 << 30
        for lo in range(0, rec.size, step):
            f.write(rec[lo:lo + step].tobytes())
        f.write(sdb.label_text)
    out["ctr_bytes"] = os.path.getsize(ctr_path)
    with open(fa_path, "wb") as f:
        done = 0
        b = 0
        while done < args.reads:
            n = min(1_000_000, args.reads - done)
            r = synth.make_reads(sdb, n, args.read_len, seed=synth.READ_SEED + b)
            f.write(synth.reads_to_fasta(r, first_index=done))
            done += n
            b += 1
    out["fasta_bytes"] = os.path.getsize(fa_path)
    out["generate_seconds"] = time.time() - t0
    del sdb
    torch.cuda.empty_cache()
    rcarg = ["RC"] if args.rc else []

    ours = os.path.join(args.dir, "ours.txt")
    our_cli = lib.RANK_CLI_PATH if args.rank else lib.CLI_PATH
    out["cli"] = os.path.basename(our_cli)
    code, secs, so, se = run([our_cli, ctr_path, fa_path, ours, str(args.threads)] + rcarg)
    out["ours"] = {"exit": code, "wall_seconds": secs, "stderr_tail": [l for l in se.splitlines() if l.startswith("[utree_amd]")]}
    for ln in se.splitlines():
        if "search" in ln and "reads/s" in ln:
            out["ours"]["search_line"] = ln.strip()
    o_sha, o_lines = sorted_sha(ours)
    out["ours"]["lines"] = o_lines
    out["ours"]["sorted_sha256"] = o_sha

    ref_bin = os.path.join(ROOT, "oracle", "_ref", ("xtree-search" if args.rank else "xtree-searchGG") + ("-k64" if args.kmer == 64 else ""))
    if not args.skip_reference and os.path.exists(ref_bin):
        empty = os.path.join(args.dir, "empty.fa")
        open(empty, "wb").close()
        code0, load_secs, _, _ = run([ref_bin, ctr_path, empty, os.path.join(args.dir, "ref_empty.txt"), str(args.threads)] + rcarg)
        ref = os.path.join(args.dir, "ref.txt")
        code, secs, so, se = run([ref_bin, ctr_path, fa_path, ref, str(args.threads)] + rcarg)
        r_sha, r_lines = sorted_sha(ref)
        out["reference"] = {"exit": code, "threads": args.threads, "wall_seconds": secs, "load_only_seconds": load_secs,
                            "search_seconds": secs - load_secs, "reads_per_second": args.reads / max(1e-9, secs - load_secs),
                            "lines": r_lines, "sorted_sha256": r_sha}
        out["parity_sorted_lines_identical"] = (r_sha == o_sha)
        if args.rank:
            sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
            out["parity_files_identical"] = (sha(ref) == sha(ours))
            print(json.dumps(out, indent=1))
            if not os.environ.get("KEEP_FILES"):
                for f in os.listdir(args.dir):
                    os.remove(os.path.join(args.dir, f))
            return
        # ordered comparison on a prefix with ONE reference thread
        n = min(args.prefix_reads, args.reads)
        pfa = os.path.join(args.dir, "prefix.fa")
        with open(fa_path, "rb") as f, open(pfa, "wb") as g:
            for _ in range(2 * n):
                g.write(f.readline())
        code, secs1, _, _ = run([ref_bin, ctr_path, pfa, os.path.join(args.dir, "ref_prefix.txt"), "1"] + rcarg)
        code2, _, _, _ = run([lib.CLI_PATH, ctr_path, pfa, os.path.join(args.dir, "ours_prefix.txt"), "8"] + rcarg)
        a = open(os.path.join(args.dir, "ref_prefix.txt"), "rb").read()
        b = open(os.path.join(args.dir, "ours_prefix.txt"), "rb").read()
        out["parity_prefix_bytes_identical"] = (a == b and len(a) > 0)
        out["prefix_reads"] = n
        out["reference_1thread_reads_per_second"] = n / max(1e-9, secs1 - load_secs)
        if args.sweep_threads:
            sweep = {}
            for t in [int(x) for x in args.sweep_threads.split(",") if x]:
                code, secs, _, _ = run([ref_bin, ctr_path, fa_path, os.path.join(args.dir, "ref_sweep.txt"), str(t)] + rcarg)
                sweep[str(t)] = args.reads / max(1e-9, secs - load_secs)
            out["reference_thread_sweep_reads_per_second"] = sweep
    print(json.dumps(out, indent=1))
    if not os.environ.get("KEEP_FILES"):
        for f in os.listdir(args.dir):
            os.remove(os.path.join(args.dir, f))


if __name__ == "__main__":
    main()
