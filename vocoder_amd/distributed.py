"""Multi-GPU plumbing: one process per GPU, streams sharded across ranks.

Channels of the vocoder are independent folds that share only the read-only parameters
(vocoder-conduit/src/Vocoder/Conduit.hs:55-72), so sharding by stream needs no collective on the
data path: each rank owns a contiguous block of streams and runs the same kernels on it.
torch.distributed is used for rendezvous, the barrier around timed regions and the max-over-ranks
of the elapsed time only.
"""
from __future__ import annotations

import os


def world():
    """(rank, local_rank, world_size) from the launcher's environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total: int, world_size: int, rank: int):
    """Contiguous block [lo, hi) of `total` streams owned by `rank`: sizes differ by at most one,
    lower ranks take the extra streams."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init(backend: str, device=None):
    """Joins the process group when launched with more than one rank; returns world size."""
    import torch.distributed as dist
    _, _, ws = world()
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return ws


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    """Largest `value` over all ranks (the step time of the slowest rank)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def finish():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
