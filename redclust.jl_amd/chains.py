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


def library_merge(ctx, local_device: int, num_samples: int, group=None):
    """The exchange step through the LIBRARY's RCCL communicator (rc_comm_*): torch.distributed (any backend) only
    carries the 128-byte unique id from rank 0 to the others.  Returns (total_samples, allreduce_ms); ctx then holds the
    merged counts.  Works without torch.distributed too (one chain: a real communicator of size 1)."""
    from ._lib import Comm
    rank, world = 0, 1
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(group), dist.get_world_size(group)
    except ImportError:
        dist = None
    uid = None
    if world > 1:
        box = [Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = box[0]
    comm = Comm([local_device], rank_offset=rank, world_size=world, unique_id=uid)
    try:
        return comm.allreduce_counts([ctx], [num_samples])
    finally:
        comm.close()


def agreed_merge(ctx, local_device: int, num_samples: int, group=None, *, probe=None, lib_merge=None, fallback=None):
    """The exchange step with the path AGREED by all ranks before anyone enters a collective.  Every rank probes the
    library's RCCL (opening librccl, `Comm.unique_id()`) inside try; the ok flags are MIN-all-reduced; if every rank can use
    the library the counts are merged by rc_comm_allreduce_counts (`library_merge`), otherwise every rank falls back to a
    torch.distributed all-reduce of the library's device count buffer (`merge_chains`).  A rank that alone failed to open
    librccl would otherwise raise while the others block in broadcast_object_list / ncclCommInitRank for ever.
    Returns (total_samples, allreduce_ms, path).  probe / lib_merge / fallback are injection points of the CPU tests."""
    import time
    import torch
    import torch.distributed as dist
    from ._lib import Comm
    probe = probe or Comm.unique_id
    ok, why = 1, ""
    try:
        probe()
    except Exception as e:   # noqa: BLE001
        ok, why = 0, str(e)
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if distributed:
        on_gpu = dist.get_backend(group) == "nccl"
        flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", local_device) if on_gpu else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = int(flag.item())
    if ok:
        total, ms = (lib_merge or library_merge)(ctx, local_device, num_samples, group)
        return total, ms, "libredclust_hip.so rc_comm_allreduce_counts (RCCL ncclAllReduce sum uint32, in place)"
    t0 = time.perf_counter()
    if fallback is not None:
        total = fallback(ctx, local_device, num_samples, group)
    else:
        counts = device_counts_tensor(ctx, local_device)     # zero-copy view: the reduction lands in the library's buffer
        _c, total, _tr = merge_chains(counts, num_samples, {}, group)
        torch.cuda.synchronize(local_device)
    ms = (time.perf_counter() - t0) * 1e3
    return total, ms, f"torch.distributed all_reduce (library RCCL unavailable on some rank{': ' + why if why else ''})"


def run_chains(data, options, params, init, *, base_seed: int = 1, verbose: bool = False, kcap: int = 0):
    """One chain per rank on its own GPU (LOCAL_RANK): a thin caller of the library — the chain runs through
    runsampler(ctx=...), the counts are merged by rc_comm_allreduce_counts (RCCL inside libredclust_hip.so; all ranks agree
    on that path first and fall back to a torch.distributed all-reduce together, `agreed_merge`) and the merged matrix is
    rc_cocluster of the merged counts.  Returns (local MCMCResult, merged posterior_coclustering (n×n float64), per-chain
    traces)."""
    import os
    import torch
    import torch.distributed as dist
    from ._lib import Context
    from .sampler import runsampler
    rank = dist.get_rank() if dist.is_initialized() else 0
    local = int(os.environ.get("LOCAL_RANK", 0))
    # one chain per GPU: the resolver's grid barrier needs the whole chip, and RCCL refuses two ranks on one device
    ndev = torch.cuda.device_count()
    if local >= ndev:
        raise RuntimeError(f"run_chains: LOCAL_RANK={local} but only {ndev} GPU(s) are visible — launch one process per GPU")
    torch.cuda.set_device(local)      # the object collectives below stage through the current device under nccl
    ctx = (Context.from_points(data.points, device=local, kcap=kcap) if data.points is not None
           else Context(data.D, device=local, kcap=kcap))
    try:
        res = runsampler(data, options, params, init, verbose=verbose and rank == 0,
                         seed=chain_seed(base_seed, rank), ctx=ctx)
        traces = dict(rank=rank, K=res.K, r=res.r, p=res.p, loglik=res.loglik, logposterior=res.logposterior)
        total, _ms, _path = agreed_merge(ctx, local, options.numsamples)
        merged = ctx.cocluster(max(total, 1))
        chains = [traces]
        if dist.is_initialized() and dist.get_world_size() > 1:
            chains = [None] * dist.get_world_size()
            dist.all_gather_object(chains, traces)
        return res, merged, chains
    finally:
        ctx.close()


def run_chains_single_process(data, options, params, init, device_ids, *, base_seed: int = 1, kcap: int = 0,
                              splitmerge: str = "as_written"):
    """All chains from ONE process (rc_run_chains: one host thread + context per device, ncclCommInitAll, one in-place
    all-reduce).  Returns (per-chain dicts of traces, merged posterior_coclustering, total number of samples)."""
    from ._lib import run_chains as _native
    P = dict(delta1=params.delta1, delta2=params.delta2, alpha=params.alpha, beta=params.beta, zeta=params.zeta,
             gamma=params.gamma, eta=params.eta, sigma=params.sigma, u=params.u, v=params.v,
             repulsion=params.repulsion, maxK=params.maxK)
    kw = dict(points=data.points) if data.points is not None and options.numMH == 0 else dict(D=data.D)
    chains, merged, total, _ms = _native(device_ids, P, init.clusts, options.numiters, options.burnin, options.thin,
                                         options.numGibbs, options.numMH, base_seed, init.r, init.p, params.proposalsd_r,
                                         kcap=kcap, splitmerge=splitmerge, **kw)
    return chains, merged, total
