"""N>1 path on CPU: world_size-2 gloo run of the chain-sharding + count all-reduce logic (merge_chains), with
the oracle standing in for the per-chain sampler (no GPU here).  Checks the merged counts against both chains
computed in one process."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import golden_case, load_golden, rp_schedule

S = 5


def _chain_counts(seed):
    import oracle_lib as O
    g, d = load_golden()
    D, P, init, _ = golden_case(g, d, "d1_random")
    orc = O.Oracle(D, P)
    orc.set_state(init)
    counts = np.zeros((100, 100), np.uint32)
    Ks = []
    for t in range(S):
        r, p = rp_schedule(t)
        orc.sweep_stable(r, p, seed, t)
        orc.L.orc_cocluster_add(100, orc.clusts, counts.reshape(-1))
        Ks.append(orc.K)
    return counts, np.array(Ks)


def _worker(rank, world, initfile, outdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import redclust_amd as rc
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    counts, Ks = _chain_counts(rc.chain_seed(1, rank))
    t = torch.from_numpy(counts.view(np.int32).copy())
    merged, total, chains = rc.merge_chains(t, S, dict(rank=rank, K=Ks))
    np.save(os.path.join(outdir, f"merged_{rank}.npy"), merged.numpy())
    np.save(os.path.join(outdir, f"meta_{rank}.npy"), np.array([total, len(chains)] + [c["rank"] for c in chains]))
    np.save(os.path.join(outdir, f"K_{rank}.npy"), np.stack([c["K"] for c in chains]))
    dist.destroy_process_group()


def test_two_chain_merge_gloo():
    world = 2
    with tempfile.TemporaryDirectory() as td:
        initfile = os.path.join(td, "init")
        mp.spawn(_worker, args=(world, initfile, td), nprocs=world, join=True)
        c0, K0 = _chain_counts(1)
        c1, K1 = _chain_counts(2)
        assert not np.array_equal(c0, c1)  # different seeds give different chains
        ref = (c0.astype(np.int64) + c1).astype(np.int32)
        for rank in range(world):
            merged = np.load(os.path.join(td, f"merged_{rank}.npy"))
            meta = np.load(os.path.join(td, f"meta_{rank}.npy"))
            assert np.array_equal(merged, ref)
            assert meta[0] == 2 * S and meta[1] == 2 and list(meta[2:]) == [0, 1]
            assert np.array_equal(np.load(os.path.join(td, f"K_{rank}.npy")), np.stack([K0, K1]))
        post = ref / (2 * S)
        assert np.all(np.diag(post) == 1.0) and np.array_equal(post, post.T)


def test_single_process_merge_is_identity():
    import redclust_amd as rc
    t = torch.arange(12, dtype=torch.int32).reshape(3, 4)
    out, total, chains = rc.merge_chains(t, 7, dict(rank=0))
    assert out is t and total == 7 and chains == [dict(rank=0)]
    assert rc.chain_seed(1, 3) == 4


def _agree_worker(rank, world, initfile, outdir, failing_rank):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import redclust_amd as rc
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    counts = torch.full((4, 4), rank + 1, dtype=torch.int32)

    def probe():
        if rank == failing_rank:
            raise RuntimeError("librccl could not be opened (injected)")

    def lib_merge(ctx, dev, ns, group):            # stands in for rc_comm_allreduce_counts: must be entered by ALL ranks or none
        t = ctx.clone(); dist.all_reduce(t); ctx.copy_(t)
        return world * ns, 0.0

    def fallback(ctx, dev, ns, group):
        _c, total, _tr = rc.merge_chains(ctx, ns, {}, group)
        return total

    total, _ms, path = rc.agreed_merge(counts, 0, 3, probe=probe, lib_merge=lib_merge, fallback=fallback)
    np.save(os.path.join(outdir, f"agree_{rank}.npy"), np.array([total, int(path.startswith("libredclust")), int(counts[0, 0])]))
    dist.destroy_process_group()


@pytest.mark.parametrize("failing_rank", [-1, 1])
def test_all_ranks_agree_on_the_merge_path(failing_rank):
    """chains.agreed_merge: if ONE rank cannot open the library's RCCL, EVERY rank takes the torch.distributed fallback (a rank
    raising alone would leave the others blocked in the unique-id broadcast or in ncclCommInitRank); if none fails, all use
    the library path.  Either way the counts are the sum over the chains."""
    world = 2
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(_agree_worker, args=(world, os.path.join(td, "init"), td, failing_rank), nprocs=world, join=True)
        for rank in range(world):
            total, lib_path, c00 = np.load(os.path.join(td, f"agree_{rank}.npy"))
            assert total == 6 and c00 == 3
            assert lib_path == (1 if failing_rank < 0 else 0)
