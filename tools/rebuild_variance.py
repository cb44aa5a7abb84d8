#!/usr/bin/env python3
"""Where does the run-to-run spread of the lane pass (+-4 %, DESIGN.md sections 0.10 / 0.12) come from?  ONE process builds the config-2 image several
times (same seed: the same bytes, a new allocation every time) and times the same 16 M-read batch on each; a spread between the builds of one
process is a property of where the image landed in HBM, not of the box or the clock.
usage (GPU box): python3 tools/rebuild_variance.py [--builds 4] [--reads 16000000]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Sampler:
    """Reads the card's DPM tables and sensors from sysfs (read-only, an ordinary user may) every few ms while a timing runs: which clock levels
    were current (the line with the '*'), junction / memory temperature, power."""
    def __init__(self):
        import glob
        self.files = {}
        for c in sorted(glob.glob("/sys/class/drm/card*/device")):
            for n in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
                if os.path.exists(os.path.join(c, n)) and n not in self.files:
                    self.files[n] = os.path.join(c, n)
            for h in glob.glob(os.path.join(c, "hwmon/hwmon*")):
                for n in ("temp1_input", "temp2_input", "temp3_input", "power1_average", "power1_input", "freq1_input", "freq2_input"):
                    if os.path.exists(os.path.join(h, n)) and n not in self.files:
                        self.files[n] = os.path.join(h, n)

    def run(self, stop, acc):
        while not stop.is_set():
            for n, f in self.files.items():
                try:
                    t = open(f).read()
                except OSError:
                    continue
                if n.startswith("pp_dpm"):
                    cur = [l.strip() for l in t.splitlines() if "*" in l]
                    v = cur[0] if cur else "?"
                else:
                    v = t.strip()
                acc.setdefault(n, {}).setdefault(v, 0)
                acc[n][v] += 1
            time.sleep(0.002)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sample-clocks", action="store_true", help="sample the card's clock levels / sensors from sysfs during every timing")
    ap.add_argument("--builds", type=int, default=4)
    ap.add_argument("--reads", type=int, default=16_000_000)
    ap.add_argument("--nodes", type=int, default=1_217_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--image-first-gib", type=int, default=0, help="allocate this many GiB for the image as the process's FIRST device allocation (before torch has "
                                                                   "allocated and freed anything) and build every image in it")
    ap.add_argument("--same-workspace", action="store_true", help="with --same-image: the workspace and the result buffer are allocated once too")
    ap.add_argument("--same-image", action="store_true", help="build ONCE and time the batch --builds times, with the work of a build (a 1.2 G-element sort) between the timings: "
                                                                "a spread here is the chip's state, not the image's pages")
    args = ap.parse_args()
    import torch
    from utree_amd import synth
    dev = torch.device("cuda:0")
    first = torch.empty(args.image_first_gib << 30, dtype=torch.uint8, device=dev) if args.image_first_gib else None
    out = {"image_first_gib": args.image_first_gib, "same_workspace": bool(args.same_workspace), "same_image": bool(args.same_image), "nodes": args.nodes, "reads_per_launch": args.reads, "builds": []}
    reads = None
    sdb = None
    for b in range(args.builds):
        if args.same_image and sdb is not None:
            junk = torch.sort(torch.randint(0, 1 << 62, (args.nodes,), dtype=torch.int64, device=dev)).values
            torch.cuda.synchronize()
            del junk
            torch.cuda.empty_cache()
        else:
            sdb = synth.make_db(dev, args.nodes, W=8, image=first)
        tree = sdb.tree
        if reads is None:
            reads = synth.make_reads(sdb, args.reads, 150, seed=synth.READ_SEED)
            tot = args.reads * 150
            res = torch.empty((args.reads, 6), dtype=torch.int32, device=dev)
        if not (args.same_workspace and b):
            ws = torch.empty(tree.workspace_bytes(args.reads, tot, 150, False), dtype=torch.uint8, device=dev)
        tree.classify(reads.bases, reads.off, reads.length, rc=False, total_bases=tot, max_len=150, out=res, workspace=ws)
        torch.cuda.synchronize()
        tree.kernel_time(reset=True)
        acc, stop, th = {}, None, None
        if args.sample_clocks:
            import threading
            stop = threading.Event()
            th = threading.Thread(target=Sampler().run, args=(stop, acc), daemon=True)
            th.start()
        t0 = time.time()
        for _ in range(args.reps):
            tree.classify(reads.bases, reads.off, reads.length, rc=False, total_bases=tot, max_len=150, out=res, workspace=ws)
        torch.cuda.synchronize()
        step_ms = 1e3 * (time.time() - t0) / args.reps
        if th is not None:
            stop.set()
            th.join()
        k_ms, k_n = tree.kernel_time(reset=True)
        tree.poll()
        ptr = tree.image_ptr()[0]
        out["builds"].append({"build": b, "step_ms": step_ms, "kernel_ms": k_ms / max(1, k_n), "us_per_M_reads": 1e3 * k_ms / max(1, k_n) / (args.reads / 1e6),
                              "image_ptr": hex(ptr), "workspace_ptr": hex(ws.data_ptr()), "image_ptr_mod_1GiB_MiB": (ptr % (1 << 30)) >> 20, "classified": int((res[:, 2] > 0).sum().item()),
                              "sensors": {n: dict(sorted(v.items(), key=lambda kv: -kv[1])[:3]) for n, v in acc.items()}})
        print(json.dumps(out["builds"][-1]), file=sys.stderr, flush=True)
        if not args.same_image:
            tree.close()
            del sdb, tree
            sdb = None
        if not args.same_workspace:
            del ws
        torch.cuda.empty_cache()
    ks = [x["kernel_ms"] for x in out["builds"]]
    out["spread"] = (max(ks) - min(ks)) / min(ks)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
