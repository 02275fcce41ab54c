"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU tests).

The front-end shards by frame with NO data-path collective: frames are independent for ORB, and
consecutive-frame matching needs a one-frame halo — rank g also processes the last frame of rank g-1 as the train side
of its first pair.  The only collective in bench.py is the MAX-reduce of the elapsed time."""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    kw = {}
    if device is not None:
        kw["device_id"] = device
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, **kw)
    return dist


def frame_shard(n_frames, rank, world):
    """Contiguous shard of a frame sequence: returns (start, stop, halo) with halo = 1 if the rank must also load frame
    start-1 (the train side of its first consecutive-frame pair), else 0."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    stop = start + base + (1 if rank < rem else 0)
    halo = 1 if (start > 0 and stop > start) else 0
    return start, stop, halo


def pair_shard(n_frames, rank, world):
    """Consecutive-frame pairs (i, i-1), i = 1..n-1, owned by this rank (pair i belongs to the owner of frame i)."""
    start, stop, _ = frame_shard(n_frames, rank, world)
    return max(start, 1), stop


def reduce_max(value, dist=None, device="cpu"):
    """MAX over ranks of a python float (bench.py's elapsed time)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
