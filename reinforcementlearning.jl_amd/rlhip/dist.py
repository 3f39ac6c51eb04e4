"""Multi-GPU sharding of the hot path: one process per GPU, independent env shards, one all-reduce of
the flat gradient per optimiser step (RCCL over xGMI on the GPU box; `gloo` in CPU tests).

The reference has no counterpart (no Distributed / MPI / NCCL anywhere).  Design:
  * rollout is embarrassingly parallel: rank r owns env ids [r * n_per_rank, (r + 1) * n_per_rank) and
    the Philox streams keyed by those GLOBAL ids, so a trajectory does not depend on the number of GPUs;
  * parameters and Adam state are replicated; every rank initialises them from the same seed;
  * per optimiser step: grad kernel -> all_reduce(SUM) of the 13 KB flat gradient -> clip+Adam kernel
    with grad_scale = 1 / world (the clip sees the GLOBAL mean gradient = single-GPU semantics with a
    world-times larger micro-batch); replicas stay bit-identical because the reduced buffer is.
"""
import os

import torch


def env_shard(rank, n_per_rank):
    """(env_id_base, n) of a rank's shard."""
    return rank * n_per_rank, n_per_rank


def init_process_group_from_env(backend=None):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, local_rank, world, group or None)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return rank, local_rank, 1, None
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world, dist.group.WORLD


def allreduce_mean_(flat_grad, group=None):
    """In-place mean of a flat gradient buffer over the group (sum all-reduce, then scale)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
        flat_grad.mul_(1.0 / world)
    return flat_grad


def max_over_ranks(value, group=None, device=None):
    """Timing reduction used by bench.py: the slowest rank defines the step time."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ("cuda" if torch.cuda.is_available() else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def params_checksum_equal(params, group=None):
    """Debug check that replicas are bit-identical: all-reduce MIN and MAX of an integer checksum."""
    import torch.distributed as dist

    c = params.view(torch.int32).to(torch.int64).sum().reshape(1)
    lo, hi = c.clone(), c.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool((lo == hi).all())
