"""Collective plumbing of the multi-GPU step; one process per GPU, torch.distributed.

The shard unit of the dense work is the (plane, channel) slice: the inverse DWT is depthwise, so slices are
independent through IDWT, its adjoint and Adam.  Per step and rank:
    plane-gradient slices [S,R,R]  --reduce_scatter-->  own S/G slices  -> adjoint -> Adam -> IDWT
    own plane slices [S/G,R,R]     --all_gather----->   all S slices
On RCCL (backend "nccl") these are reduce_scatter_tensor / all_gather_into_tensor; on gloo (the CPU tests)
the same results are produced with all_reduce / all_gather, so the N>1 logic is testable without GPUs.
"""
import torch
import torch.distributed as dist


# True: a process group of one rank still issues every collective (TrainStep(single_rank_collectives=True): the RCCL
# calls of the N-GPU step executed on a one-GPU box); False: a lone rank returns its input untouched
FORCE_COLLECTIVES = False


def _alone(world):
    return world == 1 and not (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def world_rank(pg=None):
    if not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(pg), dist.get_rank(pg)


def slice_range(S, world, rank):
    """Contiguous block of the S (plane, channel) slices owned by `rank` (S % world == 0)."""
    if S % world != 0:
        raise ValueError(f"{S} slices cannot be split evenly over {world} ranks")
    per = S // world
    return rank * per, (rank + 1) * per


def shard_rays(n_total, world, rank):
    """Contiguous split of a global ray batch (the last rank takes the remainder)."""
    per = n_total // world
    start = rank * per
    end = n_total if rank == world - 1 else start + per
    return start, end


def _native(pg):
    return dist.get_backend(pg) == "nccl"


def reduce_scatter_slices(full, pg=None):
    """full: [S, ...] on every rank -> sum over ranks of the caller's block [S/G, ...]."""
    world, rank = world_rank(pg)
    if _alone(world):
        return full
    s0, s1 = slice_range(full.shape[0], world, rank)
    if _native(pg):
        out = torch.empty((s1 - s0, *full.shape[1:]), dtype=full.dtype, device=full.device)
        dist.reduce_scatter_tensor(out, full.contiguous(), group=pg)
        return out
    tmp = full.clone()
    dist.all_reduce(tmp, group=pg)
    return tmp[s0:s1].contiguous()


def reduce_scatter_slices_async(full, pg=None):
    """reduce_scatter_slices as (result tensor, wait): the collective is enqueued now -- on RCCL it runs on the process
    group's own stream behind the work already queued on the CURRENT stream -- and wait() makes the stream current at
    that time wait for it.  gloo (the CPU-testable transport) completes inside this call; wait() is then a no-op."""
    world, rank = world_rank(pg)
    if _alone(world):
        return full, (lambda: None)
    if _native(pg):
        s0, s1 = slice_range(full.shape[0], world, rank)
        out = torch.empty((s1 - s0, *full.shape[1:]), dtype=full.dtype, device=full.device)
        work = dist.reduce_scatter_tensor(out, full.contiguous(), group=pg, async_op=True)
        return out, work.wait
    return reduce_scatter_slices(full, pg), (lambda: None)


def all_gather_slices(mine, pg=None):
    """mine: [S/G, ...] -> [S, ...] in rank order."""
    world, _ = world_rank(pg)
    if _alone(world):
        return mine
    mine = mine.contiguous()
    if _native(pg):
        out = torch.empty((mine.shape[0] * world, *mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(out, mine, group=pg)
        return out
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=pg)
    return torch.cat(parts, 0)


def all_reduce_(t, pg=None, op=None):
    world, _ = world_rank(pg)
    if not _alone(world):
        dist.all_reduce(t, op=op or dist.ReduceOp.SUM, group=pg)
    return t
