"""The slot capacity grows on demand (-m gpu).  The reference's state has room for n clusters (`clustsizes` of length n,
src/types.jl:131-137; a new cluster is offered whenever maxK allows, src/mcmc.jl:198-199); the library's slot tables start
small and are doubled when a state, a sweep or a split-merge proposal needs one more slot: the sweep that ran out is resumed at
the point that needed it, the sweeps enqueued behind it are replayed, and the chain is the one the oracle walks."""
import os

import numpy as np
import pytest

import np_transcription as T
import oracle_lib as O
import redclust_amd as rc
from helpers import golden_case, load_golden, rp_schedule

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def paper(d=1):
    z = np.load(os.path.join(HERE, "golden", "paper_datasets.npz"))
    return z[f"D{d}"], z[f"labels{d}"]


def same_state(ctx, orc, what):
    lab, sizes, K = ctx.get_state()
    assert np.array_equal(lab, orc.clusts), (what, int(np.sum(lab != orc.clusts)))
    assert np.array_equal(sizes, orc.sizes) and K == orc.K, what


@pytest.mark.parametrize("blocking", [True, False])
@pytest.mark.parametrize("kcap", [1, 4, 16])
def test_sweep_that_runs_out_of_slots_is_resumed(kcap, blocking):
    """repulsion = false shatters paper dataset 1 into dozens of clusters (golden d1_norep): from one cluster and a capacity
    of 1, 4 or 16 slots the first sweeps overflow several times — in the middle of a sweep, and with later sweeps already
    enqueued behind it (blocking = False)."""
    D, truth = paper(1)
    P = dict(T.likelihood_hyperparams(D, truth), repulsion=False)
    init = np.ones(100, np.int64)
    orc = O.Oracle(D, P)
    orc.set_state(init)
    ctx = rc.Context(D, logD=orc.logD, kcap=kcap)
    ctx.set_params(**P)
    ctx.set_state(init)
    assert ctx.capacity_info()["kcap"] == kcap
    nsweeps, changes = 6, []
    for t in range(nsweeps):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, 77, t, blocking=blocking)
        orc.sweep_stable(r, p, 77, t)
        changes.append(orc.last_changes)
        if blocking:
            same_state(ctx, orc, (kcap, t))
            assert ctx.sweep_stats()["n_changes"] == orc.last_changes, (kcap, t)
    ctx.synchronize()
    same_state(ctx, orc, (kcap, "end"))
    if not blocking:
        assert ctx.sweep_stats()["n_changes"] == changes[-1]
    info = ctx.capacity_info()
    assert info["n_grows"] >= 1 and info["kcap"] >= orc.K and orc.K > 16, (info, orc.K)
    ll, ref = ctx.loglik(), orc.loglik_stable()
    assert abs(ll - ref) <= 1e-9 * abs(ref)
    ctx.close()


def test_record_sample_behind_an_asynchronous_sweep_that_overflows():
    """rc_record_sample reads the label state: behind an rc_gibbs_sweep_async that runs out of slots it must wait for the
    recovery (capacity doubled, sweep resumed) before it snapshots — the recorded labels and the co-clustering counts are those of
    the completed sweep, not of the half-swept state mapped through the re-installed layout."""
    D, truth = paper(1)
    P = dict(T.likelihood_hyperparams(D, truth), repulsion=False)
    init = np.ones(100, np.int64)
    orc = O.Oracle(D, P)
    orc.set_state(init)
    ctx = rc.Context(D, logD=orc.logD, kcap=4)
    ctx.set_params(**P)
    ctx.set_state(init)
    ctx.cocluster_reset()
    expect = np.zeros((100, 100), np.uint32)
    for t in range(3):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, 77, t, blocking=False)
        canon = ctx.record_sample(True)                 # no synchronisation in between
        orc.sweep_stable(r, p, 77, t)
        ref = np.zeros(100, np.int64)
        O.lib().orc_sortlabels(100, orc.clusts, ref)
        assert np.array_equal(canon, ref), (t, int(np.sum(canon != ref)))
        expect += (orc.clusts[:, None] == orc.clusts[None, :]).astype(np.uint32)
    assert ctx.capacity_info()["n_grows"] >= 1 and orc.K > 4
    assert np.array_equal(ctx.cocluster_counts(), expect)
    same_state(ctx, orc, "end")
    ctx.close()


def test_set_state_with_more_clusters_than_slots_and_the_fixed_capacity_switch():
    D, truth = paper(1)
    P = T.likelihood_hyperparams(D, truth)
    singles = np.arange(1, 101, dtype=np.int64)
    orc = O.Oracle(D, P)
    orc.set_state(singles)
    ctx = rc.Context(D, logD=orc.logD, kcap=16)
    ctx.set_params(**P)
    ctx.set_state(singles)                              # 100 clusters into 16 slots: the tables grow
    assert ctx.capacity_info()["kcap"] >= 100
    for t in range(3):
        ctx.gibbs_sweep(1.0, 0.5, 5, t)
        orc.sweep_stable(1.0, 0.5, 5, t)
        same_state(ctx, orc, t)
    ctx.close()
    os.environ["RC_KCAP_FIXED"] = "1"                   # the error path stays reachable (and reported, not silently dropped)
    try:
        ctx = rc.Context(D, kcap=16)
        ctx.set_params(**dict(P, repulsion=False))
        with pytest.raises(rc.RedClustHIPError, match="RC_ERR_CAPACITY"):
            ctx.set_state(singles)
        g, d = load_golden()
        ctx.set_state(g["d1_norep_init"].astype(np.int64))
        with pytest.raises(rc.RedClustHIPError, match="RC_ERR_CAPACITY"):
            for t in range(4):
                ctx.gibbs_sweep(1.0, 0.5, 3, t)
        ctx.close()
    finally:
        del os.environ["RC_KCAP_FIXED"]


@pytest.mark.parametrize("numMH", [0, 1])
@pytest.mark.parametrize("start", ["one_cluster", "singletons"])
def test_chain_with_default_capacity_through_growth(start, numMH):
    """rc_run_chain with kcap = 0 (what runsampler and the Julia glue pass): the repulsion-free model on paper dataset 1 from
    one cluster (shatters) and from all singletons — free-running r / p, split-merge proposals included — equals the oracle's
    loop: labels, K, r, p, acceptances exactly, log-posterior to 1e-9."""
    D, truth = paper(1)
    P = dict(T.likelihood_hyperparams(D, truth), repulsion=False)
    init = np.ones(100, np.int64) if start == "one_cluster" else np.arange(1, 101, dtype=np.int64)
    orc = O.Oracle(D, P)
    ctx = rc.Context(D, logD=orc.logD)                  # kcap = 0
    ctx.set_params(**P)
    ctx.set_state(init)
    ctx.cocluster_reset()
    if numMH:
        ctx.attach_host_matrices(D, orc.logD)
    iters = 40
    ch = ctx.run_chain(iters, 4, 3, 5, numMH, 2024, 1.0, 0.5, 0.8)
    ref = O.run_chain(orc, init, 1.0, 0.5, iters, 4, 3, 5, numMH, 2024, proposalsd_r=0.8, stable=True)
    for k, kr in (("clusts", "clusts"), ("K", "K"), ("r", "r"), ("p", "p"), ("r_all", "r_all"), ("p_all", "p_all"), ("r_acceptances", "r_acc")):
        assert np.array_equal(ch[k], ref[kr]), (k, start, numMH)
    if numMH:
        assert np.array_equal(ch["splitmerge_acceptances"], ref["sm_acc"]) and np.array_equal(ch["splitmerge_splits"], ref["sm_split"])
    assert np.allclose(ch["logposterior"], ref["logposterior"], rtol=1e-9, atol=0)
    if not (numMH and start == "one_cluster"):          # (as written, an accepted split is dropped with the iteration's sweep — quirk Q1 — and from one cluster every proposal is an accepted split: K stays 1)
        assert ref["K"].max() > 16
    lab, sizes, K = ctx.get_state()
    assert np.array_equal(lab, orc.clusts) and K == orc.K
    ctx.close()


def test_more_clusters_than_the_old_default_capacity():
    """n = 700 singletons (more than the 512 slots that used to be the default) with kcap = 0: K stays in the hundreds for
    the first sweeps; a chain with split-merge proposals equals the oracle's."""
    data = rc.generatemixture(700, 6, seed=5, sigma=0.2)
    D, truth = data["distancematrix"], data["clusts"]
    P = dict(rc.likelihood_hyperparams(D, truth), repulsion=False)
    init = np.arange(1, 701, dtype=np.int64)
    orc = O.Oracle(D, P)
    ctx = rc.Context(D, logD=orc.logD)
    ctx.set_params(**P)
    ctx.set_state(init)
    assert ctx.capacity_info()["kcap"] >= 700
    ctx.cocluster_reset()
    ctx.attach_host_matrices(D, orc.logD)
    ch = ctx.run_chain(12, 0, 2, 5, 1, 9, 1.0, 0.5, 1.0)
    ref = O.run_chain(orc, init, 1.0, 0.5, 12, 0, 2, 5, 1, 9, stable=True)
    assert np.array_equal(ch["clusts"], ref["clusts"]) and np.array_equal(ch["K"], ref["K"]) and np.array_equal(ch["r"], ref["r"])
    assert np.array_equal(ch["splitmerge_acceptances"], ref["sm_acc"])
    assert np.allclose(ch["logposterior"], ref["logposterior"], rtol=1e-9, atol=0)
    ctx.close()


def test_largest_slot_tables_with_a_forced_batch_capacity():
    """kcap = 4096 with the batch capacity forced to 512 (RC_RES_MAXB) needs as much LDS as a CU has (160 KiB, give or take a
    kilobyte): rc_create must accept the request — RC_RES_MAXB is an upper bound, lowered like the default choice when the tables
    would not fit (round 4 put the table of the score logarithms into LDS) — and a moving sweep through it equals the oracle's."""
    n, K = 4200, 12
    data = rc.generatemixture(n, K, seed=3, sigma=0.3)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    init = truth.copy()
    idx = np.random.default_rng(1).choice(n, 200, replace=False)
    init[idx] = np.random.default_rng(2).integers(1, K + 1, 200)
    saved = os.environ.get("RC_RES_MAXB")
    try:
        os.environ["RC_RES_MAXB"] = "512"
        ctx = rc.Context(D, kcap=4096)
    finally:
        if saved is None: os.environ.pop("RC_RES_MAXB", None)
        else: os.environ["RC_RES_MAXB"] = saved
    ctx.set_params(**P)
    ctx.set_state(init)
    assert ctx.capacity_info()["kcap"] == 4096 and 256 <= ctx.capacity_info()["batch_capacity"] <= 512
    eD, eL = ctx.debug_rowsums(int(init[0]))[2:4]
    orc = O.Oracle(ctx.get_matrix(0), P, logD=ctx.get_matrix(1), eL=eL, eD=eD)
    orc.set_state(init)
    for t in range(2):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, 5, t); orc.sweep_stable(r, p, 5, t)
        lab, sizes, Kc = ctx.get_state()
        assert np.array_equal(lab, orc.clusts) and np.array_equal(sizes, orc.sizes) and Kc == orc.K
    ctx.close()
