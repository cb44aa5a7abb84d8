#!/usr/bin/env python3
"""bench.py -- reads classified / second on BASELINE.json's config 2: 8 GB L2 CTR (1 217 000 000 synthetic
32-mer nodes, 21 844 labels), 150 bp synthetic reads, 40 M reads per GPU (10 steps x 4 M-read batches).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" is one pass of the hot path (reverse-complement staging off, k-mer roller,
node lookup, tally, vote -> one 24-byte result per read) over one batch of reads that is already resident in
HBM; the database image is resident too.  Reads shard across ranks (each rank has its own batches: weak
scaling, no data-path collective); the only collective is the one-off RCCL broadcast of the database image
from rank 0 before the timed region.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel (classify_short_k): algorithmic bytes per launch / average launch duration
                measured with HIP events on the launch stream, against 8 TB/s HBM peak.
  cpu_baseline  (N=1 only) the CPU oracle (OpenMP port of the reference path) timed on this box's host cores
                on a bounded sample of the same batches, and a parity check of the GPU results on that sample.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_read(n_nodes: int, W: int, I: int, read_len: int):
    """B_win = 8 + SZ*(ceil(log2(nbar))+1);  B_read = windows*B_win + L + 24."""
    SZ = W + I - 3
    k = 4 * W
    nbar = n_nodes / float(1 << 24)
    steps = max(0, math.ceil(math.log2(nbar))) if nbar > 1 else 0
    b_win = 8 + SZ * (steps + 1)
    windows = max(0, read_len - k + 1)
    return windows * b_win + read_len + 24, b_win, windows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=1_217_000_000, help="synthetic CTR nodes (config 2: 1.217e9 = 8 GB)")
    ap.add_argument("--batch-reads", type=int, default=4_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--kmer", type=int, default=32, choices=(32, 64))
    ap.add_argument("--distinct-batches", type=int, default=3)
    ap.add_argument("--fine-bits", type=int, default=-1)
    ap.add_argument("--rc", type=int, default=0)
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the batches alternate over (batch i+1's lookups overlap batch i's vote)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-baseline", action="store_true", help="cpu_baseline from the oracle port only")
    ap.add_argument("--reference-threads", type=int, default=16, help="threads for the genuine reference (its best on a 256-core box)")
    ap.add_argument("--reference-reads", type=int, default=1_000_000)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only to rehearse on one GPU")
    ap.add_argument("--share-gpu0", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    # stdout carries ONE JSON line: libraries that chat on fd 1 (RCCL prints a version banner there when its
    # communicator is created) are sent to stderr until that line is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from utree_amd import dist as udist
    from utree_amd import synth
    from utree_amd.search import CtrDB, DeviceTree

    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = world > 1 or bool(os.environ.get("UTREE_BENCH_FORCE_DIST"))   # the env var rehearses the N>1 code path with one rank
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.share_gpu0:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist_on:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    W = args.kmer // 4
    want_cpu = (world == 1 and not args.no_cpu_baseline)

    # ---- database: rank 0 builds the image in HBM, the others receive it by ONE broadcast (RCCL / xGMI) ----
    t0 = time.time()
    bcast_s = 0.0
    if rank == 0:
        sdb = synth.make_db(dev, args.nodes, W=W, fine_bits=args.fine_bits, keep_raw=want_cpu)
        tree, ctr = sdb.tree, sdb.ctr
    if dist_on:
        meta = None
        if rank == 0:
            used = tree.image_ptr()[1]                      # the built image may be smaller than its allocation
            meta = dict(label_text=sdb.label_text, image_bytes=used, W=W, n_nodes=args.nodes)
        image, m, bcast_s = udist.broadcast_image(tree.image_tensor()[:used] if rank == 0 else None, meta, 0, dev)
        if rank != 0:
            dummy_bins = np.zeros((1 << 24) + 1, dtype=np.uint64)
            dummy_bins[-1] = m["n_nodes"]
            ctr = CtrDB.from_memory(m["W"], 2, m["n_nodes"], dummy_bins, None, m["label_text"])
            tree = DeviceTree.attach(ctr, image, local)
            sdb = synth.SynthDB(ctr=ctr, tree=tree, n_nodes=m["n_nodes"], W=m["W"], block=max(1, m["n_nodes"] // synth.N_LABELS),
                                seed=synth.DB_SEED, tree2file=torch.zeros(1, device=dev), label_text=m["label_text"])
    db_s = time.time() - t0

    # ---- reads: each rank's own batches, resident in HBM before the timed region ----
    nb = max(1, min(args.distinct_batches, args.steps + args.warmup))
    batches = [synth.make_reads(sdb, args.batch_reads, args.read_len, seed=synth.READ_SEED + 1000 * rank + b, device=dev)
               for b in range(nb)]
    total_bases = args.batch_reads * args.read_len
    outs = [torch.empty((args.batch_reads, 6), dtype=torch.int32, device=dev) for _ in range(nb)]
    ns = max(1, min(args.streams, nb))
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    wss = [torch.empty(tree.workspace_bytes(args.batch_reads, total_bases, args.read_len, bool(args.rc)), dtype=torch.uint8, device=dev)
           for _ in range(ns)]

    def step(i):
        # consecutive batches alternate over the streams (each with its own workspace and result buffer), the way a
        # double-buffered host pipeline submits them; every step is still one complete pass of the hot path
        b = batches[i % nb]
        with torch.cuda.stream(streams[i % ns]):
            tree.classify(b.bases, b.off, b.length, rc=bool(args.rc), total_bases=total_bases, max_len=args.read_len,
                          out=outs[i % nb], workspace=wss[i % ns])

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    tree.kernel_time(reset=True)                 # switches the HIP-event bracket of the dominant kernel on
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.time()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.time() - t1
    if dist_on:
        elapsed = udist.max_over_ranks(elapsed, dev)
    k_ms, k_launches = tree.kernel_time(reset=True)

    if rank == 0:
        reads_total = world * args.batch_reads * args.steps
        value = reads_total / elapsed
        b_read, b_win, windows = algorithmic_bytes_per_read(args.nodes, W, 2, args.read_len)
        if args.rc:
            b_read = 2 * windows * b_win + args.read_len + 24
        avg_launch_s = (k_ms / 1e3) / max(1, k_launches)
        achieved = b_read * args.batch_reads / avg_launch_s / 1e9 if k_launches else None
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                key = "nodes=%d,reads=%d,len=%d" % (args.nodes, args.batch_reads, args.read_len)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        nfound = int((outs[(args.warmup + args.steps - 1) % nb][:, 2] > 0).sum().item())
        line = {
            "metric": "reads classified/sec, 8 GB L2 CTR, 150 bp reads",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64" if W == 8 else "u128", "data": "synthetic",
            "config": {"workload": "configs[1]: %.3g-node synthetic L2 CTR (k=%d, %d labels, image %.1f GiB, fine_bits=%d), "
                                   "%d x %d bp reads per GPU (%d steps x %d-read batches), RC=%d"
                                   % (args.nodes, args.kmer, synth.N_LABELS, tree.info.image_bytes / 2**30, tree.info.fine_bits,
                                      args.batch_reads * args.steps, args.read_len, args.steps, args.batch_reads, args.rc),
                       "parallelism": "reads sharded over %d GPU(s), CTR image replicated%s; batches alternate over %d HIP stream(s) per GPU" %
                                      (world, " by one RCCL broadcast (%.2f s)" % bcast_s if dist_on else "", ns)},
            "roofline": {"bound": "hbm", "kernel": tree.kernel_name(), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "algorithmic_bytes_per_read": b_read, "reads_per_launch": args.batch_reads,
                         "avg_launch_ms": 1e3 * avg_launch_s, "launches": int(k_launches)},
            "db_build_seconds": db_s, "classified_fraction_last_batch": nfound / args.batch_reads,
        }
        if want_cpu:
            line["cpu_baseline"] = cpu_baseline(args, sdb, batches[0], outs, tree, total_bases)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, sdb, batch, outs, tree, total_bases):
    """CPU baselines on this box's host cores, rank 0, N=1 only, on a bounded sample of batch 0:

    kind "reference": the GENUINE reference binary (oracle/_ref/xtree-searchGG, compiled from /root/reference by
        `make -C oracle ref` in the build container; only the binary travels) run as a subprocess on the same
        database written out as a real `.ctr` file and the sample written as FASTA, at the thread count where
        it is fastest on this class of box (its `omp critical` input section makes it SLOWER beyond ~16
        threads: profiles/r01/reference_thread_sweep.txt); database load time taken out with an empty-FASTA run.
    kind "port": the CPU oracle (oracle/, OpenMP parallel-for over reads, no input critical section) on all cores.
    Both also serve as parity checks of the GPU results on the sample."""
    import hashlib
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    import torch
    from oracle import orc
    cores = os.cpu_count() or 1
    L = batch.read_len
    cap = min(batch.n, 2_000_000)
    res = tree.classify(batch.bases, batch.off, batch.length, rc=bool(args.rc), total_bases=total_bases, max_len=L).cpu().numpy()
    host = batch.bases[: cap * L].cpu().numpy()
    off = np.arange(cap, dtype=np.uint64) * L
    ln = np.full(cap, L, dtype=np.uint32)
    out = {"value": None, "unit": "reads/s", "cores": cores, "kind": "port", "sample": ""}
    try:
        binix_u32 = sdb.binix.cpu().numpy().view(np.uint32)
        records = sdb.records.cpu().numpy()
        o = orc.OracleDB.from_memory(sdb.W, 2, binix_u32.astype(np.uint64), records, sdb.label_text)
    except (MemoryError, RuntimeError) as e:
        out["sample"] = "skipped: %s" % e
        return out
    probe = 20000
    t0 = time.time()
    o.classify_batch(host, off[:probe], ln[:probe], rc=bool(args.rc), threads=cores)
    rate0 = probe / max(1e-6, time.time() - t0)
    n = int(min(cap, max(probe, rate0 * args.cpu_seconds)))
    t0 = time.time()
    want = o.classify_batch(host, off[:n], ln[:n], rc=bool(args.rc), threads=cores)
    dt = time.time() - t0
    got = res[:n].view(np.uint32)
    hit = want["found"] > 0
    multi = hit & (want["uix"] > 1)
    ok = (np.array_equal(got[:, 2], want["found"]) and np.array_equal(got[hit, 3], want["uix"][hit]) and
          np.array_equal(got[hit, 0], want["label"][hit]) and np.array_equal(res[:n][hit, 1], want["cut"][hit]) and
          np.array_equal(got[multi, 4], want["sl"][multi]) and np.array_equal(got[multi, 5], want["ol"][multi]))
    port = {"value": n / dt, "unit": "reads/s", "cores": cores, "kind": "port", "parity_ok": bool(ok),
            "sample": "first %d reads of batch 0 (same DB, same reads), %.1f s on %d OpenMP threads; GPU results on the "
                      "sample bit-identical to the CPU oracle: %s" % (n, dt, cores, bool(ok))}
    out = dict(port)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "xtree-searchGG" + ("-k64" if sdb.W == 16 else ""))
    if not os.path.exists(ref_bin) or args.no_reference_baseline:
        return out
    tmp = None
    try:
        base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 3 * records.nbytes else None
        tmp = tempfile.mkdtemp(prefix="utree_bench_", dir=base)
        ctr_path, fa, empty = os.path.join(tmp, "db.ctr"), os.path.join(tmp, "sample.fa"), os.path.join(tmp, "empty.fa")
        with open(ctr_path, "wb") as f:
            f.write(np.array([sdb.W, 0, 2, sdb.n_nodes], dtype="<u8").tobytes())
            f.write(binix_u32.tobytes())
            for lo in range(0, records.size, 1 << 30):
                f.write(records[lo:lo + (1 << 30)].tobytes())
            f.write(sdb.label_text)
        T = args.reference_threads
        nref = int(min(cap, max(50_000, args.reference_reads)))
        seq = host[: nref * L].reshape(nref, L)
        with open(fa, "wb") as f:
            for i in range(nref):
                f.write(b">r%d\n" % i)
                f.write(seq[i].tobytes())
                f.write(b"\n")
        open(empty, "wb").close()
        rcarg = ["RC"] if args.rc else []

        def run(fasta, outp):
            t = time.time()
            subprocess.run([ref_bin, ctr_path, fasta, outp, str(T)] + rcarg, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           check=True, timeout=900)
            return time.time() - t
        t_load = run(empty, os.path.join(tmp, "e.txt"))
        t_all = run(fa, os.path.join(tmp, "ref.txt"))
        search = max(1e-6, t_all - t_load)
        # parity: GPU results formatted by the product's formatter == the reference's lines (as a multiset:
        # the reference writes in thread-completion order)
        names_off = np.zeros(nref, dtype=np.uint64)
        names = b"".join(b"r%d" % i for i in range(nref))
        lens = np.array([len(b"r%d" % i) for i in range(nref)], dtype=np.uint32)
        names_off[1:] = np.cumsum(lens[:-1])
        ours = sdb.ctr.format(np.frombuffer(names, dtype=np.uint8), names_off, lens, res[:nref])
        ref_lines = sorted(open(os.path.join(tmp, "ref.txt"), "rb").read().split(b"\n"))
        same = sorted(ours.split(b"\n")) == ref_lines
        out = {"value": nref / search, "unit": "reads/s", "cores": T, "kind": "reference", "parity_ok": bool(same and ok),
               "sample": "genuine reference binary (itree.c -D SEARCH_GG), %d threads, first %d reads of batch 0 on the same "
                         "database written as a .ctr file: %.1f s total - %.1f s load-only run = %.2f s search; its output "
                         "lines == GPU results formatted by the product (multiset): %s" % (T, nref, t_all, t_load, search, same),
               "port": port}
    except Exception as e:  # the baseline must never take the benchmark line down
        out["reference_error"] = repr(e)
    finally:
        if tmp:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


if __name__ == "__main__":
    main()
