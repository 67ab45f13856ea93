"""Multi-GPU sharding of the front-end: frames / sequences are independent, so stream `r` goes to
rank `r` (one process per GPU) and nothing is exchanged while frames are processed.  The only
collectives are the barrier + max-over-ranks of the elapsed time and ONE final gather of the
per-rank result digests (RCCL on GPUs, gloo in the CPU tests)."""
import os

import torch
import torch.distributed as dist


def rank_info():
    """(rank, world_size, local_rank) from the torch.distributed.run environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def launched():
    """True under a launcher (torch.distributed.run exports RANK, WORLD_SIZE and the rendezvous address, also for one rank)."""
    return all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))


def init(backend, device=None, force=False):
    """Joins the process group when WORLD_SIZE > 1, and also at world size 1 when a launcher started this process (or
    force=True): a one-rank RCCL group runs the same barrier / all_reduce / all_gather code as the N-rank job.
    Returns (rank, world)."""
    rank, world, _ = rank_info()
    if (world > 1 or force or launched()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kwargs = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def stream_for_rank(rank, base_stream=0):
    """Synthetic stream / sequence index owned by `rank` (weak scaling: one stream per GPU)."""
    return base_stream + rank


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds, device="cpu"):
    """Whole-job time = slowest rank."""
    if not dist.is_initialized():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_digests(digest, device="cpu"):
    """Final gather: every rank's digest (a short float64 vector: keypoints, matches, checksum ...)
    in rank order, as a list of lists on every rank."""
    d = torch.as_tensor(digest, dtype=torch.float64, device=device)
    if not dist.is_initialized():
        return [d.tolist()]
    out = [torch.zeros_like(d) for _ in range(dist.get_world_size())]
    dist.all_gather(out, d)
    return [o.tolist() for o in out]


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
