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


# ... or, scoped to the groups that asked for it (TrainStep(single_rank_collectives=True) registers ITS process group:
# another TrainStep or model in the same process keeps the short cut): see force_collectives()
_FORCED_GROUPS = set()


def _group_key(pg):
    return "default" if pg is None else id(pg)


def force_collectives(pg=None, on=True):
    """A process group of one rank issues every collective of this module (on=True) or takes the identity short cut again."""
    (_FORCED_GROUPS.add if on else _FORCED_GROUPS.discard)(_group_key(pg))


def _alone(world, pg=None):
    forced = FORCE_COLLECTIVES or _group_key(pg) in _FORCED_GROUPS
    return world == 1 and not (forced and dist.is_available() and dist.is_initialized())


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


# ---- the exchange's cost model -----------------------------------------------------------------------
# One xGMI link per pair of GPUs (7 per GPU, ~153 GB/s each way by specification); the direct reduce-scatter / all-gather
# RCCL runs on a fully connected node moves bytes / G over every link at once.  TNL_XGMI_GBS = the rate per link and
# direction the plan assumes (default 0.8 x 153); a measured SCALE curve replaces it.
import os as _os

XGMI_GBS_DEFAULT = 122.0
# single-GPU section times at the base configuration (ms; profiles/r06a_bench_default.json), per unit of what they scale
# with: W = work that does not shard (field forward / backward, compositing, launches) per million samples, TILE = the tile
# reduction per million samples, D = dense work that shards (rebuild, adjoint + optimiser, coefficient pass) per GB of
# fp32 plane-gradient window, FIXED = refresh + replay amortised and the small all-reduces
_W_MS_PER_MSAMPLE = (0.78 + 0.10 + 0.59 + 0.10) / 4.65
_TILE_MS_PER_MSAMPLE = 0.35 / 4.65
_D_MS_PER_GB = (0.51 + 0.85 + 0.23) / 0.51
_FIXED_MS = 0.38
_BAND_MS = 0.03            # every extra band of overlap_exchange costs the tile reduction this much (tools/ab_exchange.sh)


def link_rate_gbs():
    try:
        return max(float(_os.environ.get("TNL_XGMI_GBS", XGMI_GBS_DEFAULT)), 1e-3)
    except ValueError:
        return XGMI_GBS_DEFAULT


def plan_exchange(world, slices, window_texels, samples, plane_bytes=2, link_gbs=None, transports=("fp32",), max_bands=4):
    """Mode, band count and transport of the plane-gradient exchange for `world` ranks from the cost model of DESIGN.md
    section 5 (weak scaling: every rank marches its own `samples`):

        T = W + D / G' + FIXED + exposed reduce-scatter + all-gather         (milliseconds)

    slices = 3 * channels, window_texels = texels of one slice's occupancy window (R^2 without one), plane_bytes = bytes per
    element of the sampler's planes.  "sharded": G' = G, reduce-scatter of the gradient window (4 B per element, 2 with
    transport "bf16") + all-gather of the rebuilt window, each bytes / (G x link rate); K bands hide the reduce-scatter
    behind (K - 1) / K of the tile reduction at _BAND_MS per extra band.  "allreduce": G' = 1 and the gradient travels twice.
    "sharded" needs slices % world == 0.  Returns a dict: mode, overlap_exchange, transport, ms (the chosen plan's
    prediction), ms_one_gpu, table (every candidate).  transports: the candidates the caller allows -- "bf16" only where a
    PSNR run has cleared it (profiles/r06_psnr_ci_bf16_transport.json)."""
    B = link_gbs if link_gbs is not None else link_rate_gbs()
    ms = lambda gbytes: gbytes / B * 1e3           # GB over one link -> ms
    msamp = samples / 1e6
    grad_gb = slices * window_texels * 4.0 / 1e9
    planes_gb = slices * window_texels * float(plane_bytes) / 1e9
    W, tile = _W_MS_PER_MSAMPLE * msamp, _TILE_MS_PER_MSAMPLE * msamp
    D = _D_MS_PER_GB * grad_gb
    one = W + tile + D + _FIXED_MS
    if world <= 1:
        return {"mode": None, "overlap_exchange": 0, "transport": "fp32", "ms": one, "ms_one_gpu": one, "table": []}
    table = []
    G = world
    for tr in transports:
        f = 0.5 if tr == "bf16" else 1.0
        if slices % G == 0:
            rs, ag = ms(f * grad_gb / G), ms(planes_gb / G)
            for K in range(1, max_bands + 1):
                hidden = min(rs * (K - 1) / K, tile * (K - 1) / K) if K > 1 else 0.0
                t = W + tile + _BAND_MS * (K - 1) + D / G + _FIXED_MS + (rs - hidden) + ag
                table.append({"mode": "sharded", "overlap_exchange": K if K > 1 else 0, "transport": tr, "ms": t})
        table.append({"mode": "allreduce", "overlap_exchange": 0, "transport": tr,
                      "ms": W + tile + D + _FIXED_MS + 2.0 * ms(f * grad_gb / G)})
    best = min(table, key=lambda e: (e["ms"], e["overlap_exchange"]))
    return {**best, "ms_one_gpu": one, "table": table, "link_gbs": B}


def _reduce_scatter_bf16(full, pg, world, rank):
    """The same sum with the contributions travelling as bfloat16 (half the bytes of the exchange's larger half, SURVEY.md
    8(e)) and ACCUMULATED IN fp32 ON THE OWNER, in rank order: every rank's block -- the owner's own included -- is rounded
    to bf16 once, so the result does not depend on who owns a slice and the replicas stay bit-identical.  RCCL: one
    all_to_all_single of [G, S/G, ...] bf16; gloo (the CPU-testable transport, no all_to_all): all_gather of the bf16
    arrays."""
    s0, s1 = slice_range(full.shape[0], world, rank)
    half = full.to(torch.bfloat16).contiguous()
    if _native(pg):
        recv = torch.empty_like(half)
        dist.all_to_all_single(recv, half, group=pg)          # block r of `recv` = rank r's rounding of MY slices
        return recv.view(world, s1 - s0, *full.shape[1:]).to(torch.float32).sum(0)
    parts = [torch.empty_like(half) for _ in range(world)]
    dist.all_gather(parts, half, group=pg)
    return torch.stack([p[s0:s1] for p in parts]).to(torch.float32).sum(0)


def reduce_scatter_slices(full, pg=None, transport="fp32"):
    """full: [S, ...] on every rank -> sum over ranks of the caller's block [S/G, ...].  transport "bf16": see
    _reduce_scatter_bf16 (a lone rank rounds its own contribution the same way: what a world of one would exchange)."""
    world, rank = world_rank(pg)
    if transport == "bf16" and (world > 1 or not _alone(world, pg)):
        return _reduce_scatter_bf16(full, pg, world, rank)
    if _alone(world, pg):
        return full
    s0, s1 = slice_range(full.shape[0], world, rank)
    if _native(pg):
        out = torch.empty((s1 - s0, *full.shape[1:]), dtype=full.dtype, device=full.device)
        dist.reduce_scatter_tensor(out, full.contiguous(), group=pg)
        return out
    tmp = full.clone()
    dist.all_reduce(tmp, group=pg)
    return tmp[s0:s1].contiguous()


def reduce_scatter_slices_async(full, pg=None, transport="fp32"):
    """reduce_scatter_slices as (result tensor, wait): the collective is enqueued now -- on RCCL it runs on the process
    group's own stream behind the work already queued on the CURRENT stream -- and wait() makes the stream current at
    that time wait for it.  gloo (the CPU-testable transport) completes inside this call; wait() is then a no-op."""
    world, rank = world_rank(pg)
    if transport == "bf16" and (world > 1 or not _alone(world, pg)):
        if _native(pg):
            s0, s1 = slice_range(full.shape[0], world, rank)
            half = full.to(torch.bfloat16).contiguous()
            recv = torch.empty_like(half)
            work = dist.all_to_all_single(recv, half, group=pg, async_op=True)
            box = {}

            def wait():
                work.wait()
                box["out"].copy_(recv.view(world, s1 - s0, *full.shape[1:]).to(torch.float32).sum(0))
            box["out"] = torch.empty((s1 - s0, *full.shape[1:]), dtype=torch.float32, device=full.device)
            return box["out"], wait
        return _reduce_scatter_bf16(full, pg, world, rank), (lambda: None)
    if _alone(world, pg):
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
    if _alone(world, pg):
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
    if not _alone(world, pg):
        dist.all_reduce(t, op=op or dist.ReduceOp.SUM, group=pg)
    return t
