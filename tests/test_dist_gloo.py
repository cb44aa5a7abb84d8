"""N>1 path on CPU: world_size-2 `gloo` processes exercise the sharding / image-replication / ordered-concat
plumbing of utree_amd.dist (the part of the multi-GPU path that is not a kernel).  The per-shard "classify"
stand-in here is the CPU oracle, so the test also shows that contiguous sharding + rank-order concatenation
reproduces the single-process output byte for byte."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from utree_amd import dist as udist
import util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ctr_path, fasta_path, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import orc
        from utree_amd.search import CtrDB, frame_fasta
        # (1) image replication: rank 0 owns a fake flat image, everybody must end up with the same bytes
        img = None
        meta = None
        if rank == 0:
            g = torch.Generator().manual_seed(7)
            img = torch.randint(0, 256, (3_000_001,), dtype=torch.uint8, generator=g)
            meta = {"image_bytes": img.numel(), "label_text": b"a\t1\n", "n_nodes": 5}
        img, meta, secs = udist.broadcast_image(img, meta, 0, torch.device("cpu"))
        g = torch.Generator().manual_seed(7)
        want = torch.randint(0, 256, (3_000_001,), dtype=torch.uint8, generator=g)
        assert torch.equal(img, want) and meta["n_nodes"] == 5 and secs >= 0
        # (2) contiguous read shards, classified independently, concatenated in rank order
        data = open(fasta_path, "rb").read()
        fr = frame_fasta(data)
        n = len(fr["seq_off"])
        lo, hi = udist.shard_range(n, rank, world)
        o = orc.OracleDB.load(ctr_path)
        db = CtrDB.open(ctr_path)
        buf = np.frombuffer(data, dtype=np.uint8)
        res = o.classify_batch(buf, fr["seq_off"][lo:hi], fr["seq_len"][lo:hi], rc=True, threads=2)
        text = db.format(buf, fr["name_off"][lo:hi], fr["name_len"][lo:hi], res)
        full = udist.gather_outputs_in_order(text, dst=0)
        # (3) the bench's timing reduction
        m = udist.max_over_ranks(float(rank + 1), torch.device("cpu"))
        assert m == float(world)
        if rank == 0:
            open(os.path.join(out_dir, "out.txt"), "wb").write(full)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_ranges_partition_exactly():
    for n in (0, 1, 7, 10_000, 40_000_001):
        for world in (1, 2, 3, 8):
            r = [udist.shard_range(n, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_two_rank_gloo_shard_and_concat(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, util.fixture_ctr("toy"), util.fixture_reads_path("toy"), str(tmp_path)), nprocs=2,
             join=True)
    assert (tmp_path / "out.txt").read_bytes() == util.fixture_bytes("toy_out_rc.txt.gz")


def test_bench_gpus_flag_launches_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` outside a launcher starts N ranks itself (before anything touches the GPU) with the driver's
    own command line; inside a launcher the flag must agree with WORLD_SIZE."""
    import argparse
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert bench.self_launch(argparse.Namespace(gpus=4)) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_launch_size_defaults():
    """What bench.py runs when the command line leaves the launch size open (DESIGN_APPENDIX.md section 0.12): 16 M reads per launch for the 150-bp
    configurations, about as many bases per launch for long reads, the kept profile's 4 M for the hit-dense workload; the file -> file leg
    takes config 2's 40 M reads, or the whole run when that is less.  An explicit --batch-reads / --e2e-reads stands."""
    import argparse
    import bench

    def r(**kw):
        a = dict(batch_reads=0, workload="config", read_len=150, e2e_reads=0, steps=10)
        a.update(kw)
        return bench.resolve_defaults(argparse.Namespace(**a))

    a = r()
    assert (a.batch_reads, a.e2e_reads) == (16_000_000, 40_000_000)
    assert r(steps=2).e2e_reads == 32_000_000
    assert r(workload="hit_dense").batch_reads == 4_000_000
    assert r(read_len=10_000).batch_reads == 200_000
    assert r(read_len=250).batch_reads == 16_000_000
    a = r(batch_reads=4_000_000, e2e_reads=1_000_000)
    assert (a.batch_reads, a.e2e_reads) == (4_000_000, 1_000_000)


def test_bench_gpus_flag_cold_start_reaches_every_rank(tmp_path):
    """The real thing from a cold start, on this GPU-less box: two ranks come up under torch.distributed.run and each one refuses
    to run without an MI355X (there is no CPU fallback); the launcher's exit code comes back."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the N-rank launch is exercised by bench.py itself")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "launching" in p.stderr and p.stderr.count("bench.py needs an MI355X") >= 2
    assert p.stdout.strip() == ""
