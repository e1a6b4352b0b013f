"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for CPU rehearsal).

The encode path shards: every segment's K-step loop depends only on that segment and the
replicated dictionary (/root/reference/modules/matchingpursuit.py:298-328 operate row-wise on
the batch), so a batch is split contiguously over ranks and NO collective touches the data
path.  The only cross-segment quantity on the whole surface is the per-atom window sum of
dictionary_learning_step (:400-401) -- one [L] fp64 all-reduce per used atom -- and, for the
gradient-trained dictionary of config 5, one all-reduce of the [A, L] gradient per step.
"""
import os

import torch
import torch.distributed as dist


def is_distributed(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def init_from_env(backend=None):
    """Initialise the default process group from torchrun-style environment variables
    (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK).  Returns (rank, world, local_rank).
    A single process (WORLD_SIZE unset or 1) needs no group and gets (0, 1, 0)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, world, local_rank


def shard_range(n_items, rank, world):
    """Contiguous split of range(n_items); the first n_items % world ranks get one extra."""
    base, extra = divmod(int(n_items), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _backend_needs_cpu(group=None):
    return dist.get_backend(group) == "gloo"


def all_reduce_sum(t, group=None):
    """In-place-style sum over ranks; identity without a process group."""
    if not is_distributed(group):
        return t
    if _backend_needs_cpu(group) and t.is_cuda:
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        return c.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_reduce_max(t, group=None):
    if not is_distributed(group):
        return t
    if _backend_needs_cpu(group) and t.is_cuda:
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
        return c.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def gather_batch(t, group=None):
    """Concatenate per-rank shards [B_r, ...] along dim 0 in rank order (shards may differ in
    size).  Returns (global tensor, this rank's row offset)."""
    if not is_distributed(group):
        return t, 0
    rank, world = rank_world(group)
    on_cpu = _backend_needs_cpu(group)
    src = t.cpu() if (on_cpu and t.is_cuda) else t
    sizes = torch.zeros(world, dtype=torch.int64, device=src.device)
    sizes[rank] = src.shape[0]
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
    sizes_l = sizes.tolist()
    mx = max(sizes_l)
    pad = torch.zeros((mx,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[: src.shape[0]] = src
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.cat([p[:n] for p, n in zip(parts, sizes_l)], dim=0)
    return out.to(t.device), int(sum(sizes_l[:rank]))


def barrier(group=None):
    if is_distributed(group):
        dist.barrier(group=group)
