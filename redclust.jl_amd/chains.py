"""Chain-parallel execution: independent chains, one per rank/GPU; the only exchange is the
element-wise SUM all-reduce of the n×n integer co-clustering counts plus the gathering of the per-chain traces.
The reference has no multi-chain concept (`src/mcmc.jl:560` is per run); merged posterior_coclustering =
Σ_chains counts / Σ_chains numsamples.  Works with any torch.distributed backend (nccl = RCCL on the GPUs,
gloo in the CPU tests)."""
from __future__ import annotations

import numpy as np


def chain_seed(base_seed: int, rank: int) -> int:
    """Chain seeds base+0 .. base+N-1 (bench: 1..N)."""
    return int(base_seed) + int(rank)


def device_counts_tensor(ctx, device_index: int):
    """Zero-copy torch view (int32 bit pattern of the uint32 counts) of the library's device count matrix."""
    import torch
    ptr, ldc = ctx.cocluster_device_buffer()

    class _Buf:
        __cuda_array_interface__ = {"shape": (ctx.n, ldc), "typestr": "<i4", "data": (ptr, False), "version": 3}

    return torch.as_tensor(_Buf(), device=torch.device("cuda", device_index))


def merge_chains(counts, numsamples: int, traces: dict, group=None):
    """counts: torch integer tensor (any device) holding this chain's co-clustering counts — reduced IN PLACE.
    Returns (posterior_coclustering_counts_sum, total_numsamples, list of per-chain trace dicts)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return counts, int(numsamples), [traces]
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)       # exact integer sums: order-independent
    ns = torch.tensor([int(numsamples)], dtype=torch.int64, device=counts.device)
    dist.all_reduce(ns, op=dist.ReduceOp.SUM, group=group)
    gathered = [None] * dist.get_world_size(group)
    dist.all_gather_object(gathered, traces, group=group)
    return counts, int(ns.item()), gathered


def run_chains(data, options, params, init, *, base_seed: int = 1, verbose: bool = False, kcap: int = 0):
    """One chain per rank on its own GPU (LOCAL_RANK), then the RCCL all-reduce of the counts.
    Returns (local MCMCResult, merged posterior_coclustering (n×n float64), per-chain traces)."""
    import os
    import torch
    import torch.distributed as dist
    from ._lib import Context
    from .sampler import runsampler
    rank = dist.get_rank() if dist.is_initialized() else 0
    local = int(os.environ.get("LOCAL_RANK", 0))
    ctx = (Context.from_points(data.points, device=local, kcap=kcap) if data.points is not None
           else Context(data.D, device=local, kcap=kcap))
    try:
        res = runsampler(data, options, params, init, verbose=verbose and rank == 0,
                         seed=chain_seed(base_seed, rank), ctx=ctx)
        counts = device_counts_tensor(ctx, local)
        traces = dict(rank=rank, K=res.K, r=res.r, p=res.p, loglik=res.loglik, logposterior=res.logposterior)
        counts, total, chains = merge_chains(counts, options.numsamples, traces)
        n = data.n
        merged = counts[:, :n].to(torch.float64).cpu().numpy() / max(total, 1)
        return res, merged, chains
    finally:
        ctx.close()
