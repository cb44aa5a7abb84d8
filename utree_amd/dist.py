"""Multi-GPU plumbing for the SEARCH_GG path: one process per GPU, torch.distributed ("nccl" = RCCL on ROCm).

The path shards by reads and has NO data-path collective: the database image is replicated
by ONE broadcast before any read is classified, every rank then classifies a contiguous range of reads, and
the per-rank outputs are concatenated on the host in rank order (= input order).  The only other traffic is
bench.py's barrier and its MAX-reduction of the elapsed time.

Everything here is backend-agnostic so that tests can run it with `gloo` on CPU tensors (world_size 2).
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [g*R/G, (g+1)*R/G) of rank g: contiguity makes the ordered output a
    plain concatenation."""
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


def broadcast_image(image, meta: Optional[dict], src: int, device):
    """Replicate the flat device image (a uint8 tensor) and a small metadata dict from `src` to every rank.
    Non-source ranks pass image=None, meta=None.  Returns (image, meta, seconds spent in the image broadcast)."""
    import time
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    box = [meta if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    meta = box[0]
    if rank != src:
        image = torch.empty(int(meta["image_bytes"]), dtype=torch.uint8, device=device)
    if image.is_cuda:
        torch.cuda.synchronize(device)
    dist.barrier()
    t0 = time.time()
    # one collective for the whole image; chunked only because a single NCCL/RCCL count is limited to < 2^31
    # elements on some torch builds
    step = 1 << 30
    for lo in range(0, image.numel(), step):
        dist.broadcast(image[lo:lo + step], src=src)
    if image.is_cuda:
        torch.cuda.synchronize(device)
    return image, meta, time.time() - t0


def replicate_image_c(tree, ctr_of_rank, meta: Optional[dict], src: int, device_index: int):
    """The same replication through the C-ABI (utree_dev_replicate_rank: ncclCommInitRank + ONE ncclBroadcast issued from C,
    as north_star asks of the host orchestration).  torch.distributed only carries the control plane: the RCCL unique id and the
    small metadata dict.  `ctr_of_rank(meta)` builds the non-source ranks' host-side database object from the metadata.
    Returns (tree, ctr, meta, seconds)."""
    import time
    import torch.distributed as dist
    from .search import DeviceTree
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [(meta, DeviceTree.rccl_unique_id()) if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    meta, uid = box[0]
    ctr = None if rank == src else ctr_of_rank(meta)
    dist.barrier()
    t0 = time.time()
    tree = DeviceTree.replicate_rank(ctr, tree if rank == src else None, device_index, rank, world, src, uid)
    return tree, ctr, meta, time.time() - t0


def max_over_ranks(value: float, device) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_outputs_in_order(local: bytes, dst: int = 0) -> Optional[bytes]:
    """Host-side concatenation of the per-rank output text in rank order (only `dst` gets the result)."""
    import torch.distributed as dist
    world = dist.get_world_size()
    parts: List[Optional[bytes]] = [None] * world
    dist.gather_object(local, parts if dist.get_rank() == dst else None, dst=dst)
    if dist.get_rank() != dst:
        return None
    return b"".join(parts)
