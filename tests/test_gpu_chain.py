"""GPU parity of the native iteration loop rc_run_chain (csrc/chain.inc.hip; runsampler's loop, mcmc.jl:533-556)
against the oracle's loop and the golden chains, free-running (the scalar r / p updates included) and teacher-forced."""
import os

import numpy as np
import pytest

import np_transcription as T
import oracle_lib as O
import redclust_amd as rc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
LL_RTOL = 1e-9


def paper(d=1):
    z = np.load(os.path.join(HERE, "golden", "paper_datasets.npz"))
    return z[f"D{d}"], z[f"labels{d}"]


def make(D, P, init):
    orc = O.Oracle(D, P)
    ctx = rc.Context(D, logD=orc.logD)
    ctx.set_params(**P)
    ctx.set_state(init)
    ctx.cocluster_reset()
    return orc, ctx


@pytest.mark.parametrize("tag", ["d1_gibbs", "d1_mh"])
def test_free_running_chain_vs_oracle_and_golden(tag):
    g = np.load(os.path.join(HERE, "golden", "golden_chain.npz"))
    D, truth = paper(1)
    P = T.likelihood_hyperparams(D, truth)
    init = g[f"{tag}_init"]
    numMH, iters, seed = int(g[f"{tag}_numMH"]), int(g[f"{tag}_iters"]), int(g[f"{tag}_seed"])
    orc, ctx = make(D, P, init)
    if numMH:
        ctx.attach_host_matrices(D, orc.logD)
    ch = ctx.run_chain(iters, 5, 2, 5, numMH, seed, 1.0, 0.5, 0.7)
    ref = O.run_chain(orc, init, 1.0, 0.5, iters, 5, 2, 5, numMH, seed, proposalsd_r=0.7, stable=True)
    ns = len(ref["K"])
    assert ch["num_samples"] == ns == (iters - 5) // 2
    # scalar draws, labels and split–merge decisions: exact, against the oracle and the golden vectors
    for got, want in ((ch["r_all"], ref["r_all"]), (ch["p_all"], ref["p_all"]), (ch["r"], ref["r"]), (ch["p"], ref["p"]),
                      (ch["clusts"], ref["clusts"]), (ch["K"], ref["K"]), (ch["r_acceptances"], ref["r_acc"]),
                      (ch["r_all"], g[f"{tag}_r_all"]), (ch["p_all"], g[f"{tag}_p_all"]), (ch["clusts"], g[f"{tag}_clusts"]),
                      (ch["K"], g[f"{tag}_K"])):
        assert np.array_equal(got, want)
    if numMH:
        assert np.array_equal(ch["splitmerge_acceptances"], ref["sm_acc"]) and np.array_equal(ch["splitmerge_splits"], ref["sm_split"])
        assert np.array_equal(ch["splitmerge_acceptances"], g[f"{tag}_sm_acc"]) and ch["splitmerge_acceptances"].sum() >= 1
    assert np.allclose(ch["loglik"], ref["loglik"], rtol=LL_RTOL, atol=0)
    assert np.allclose(ch["logposterior"], ref["logposterior"], rtol=1e-6, atol=0)      # north_star tolerance
    assert np.allclose(ch["loglik"], g[f"{tag}_loglik"], rtol=1e-7)                     # literal arithmetic there
    # co-clustering of the recorded samples
    acc = sum((c[:, None] == c[None, :]).astype(np.float64) for c in ref["clusts"])
    assert np.array_equal(ctx.cocluster(ns), acc / ns)
    # the final device state is the oracle's
    lab, sizes, K = ctx.get_state()
    assert np.array_equal(lab, orc.clusts) and K == orc.K
    ctx.close()


def test_teacher_forced_and_intended_mode():
    D, truth = paper(1)
    P = T.likelihood_hyperparams(D, truth)
    init = truth.copy(); init[init == 2] = 1; init[init == 4] = 3; init[init == 9] = 8
    rs = np.full(30, 1.0); ps = np.full(30, 0.5)
    orc, ctx = make(D, P, init)
    ctx.attach_host_matrices(D, orc.logD)
    ch = ctx.run_chain(30, 0, 3, 5, 2, 4322, 1.0, 0.5, 1.0, splitmerge="intended", rp_trace=(rs, ps))
    ref = O.run_chain(orc, init, 1.0, 0.5, 30, 0, 3, 5, 2, 4322, rp_trace=(rs, ps), stable=True, intended=True)
    assert np.array_equal(ch["clusts"], ref["clusts"]) and np.array_equal(ch["splitmerge_acceptances"], ref["sm_acc"])
    assert np.array_equal(ch["splitmerge_splits"], ref["sm_split"]) and np.array_equal(ch["r"], ref["r"])
    assert ch["splitmerge_acceptances"].sum() >= 1
    assert np.allclose(ch["loglik"], ref["loglik"], rtol=LL_RTOL, atol=0)
    ctx.close()


def test_continuation_and_bad_arguments():
    """first_iter continues the streams: 10 + 15 iterations = 25 iterations."""
    D, truth = paper(2)
    P = T.likelihood_hyperparams(D, truth)
    init = np.random.default_rng(1).integers(1, 11, 100).astype(np.int64)
    _, a = make(D, P, init)
    _, b = make(D, P, init)
    full = a.run_chain(25, 0, 1, 5, 0, 5, 1.0, 0.5, 1.0)
    h1 = b.run_chain(10, 0, 1, 5, 0, 5, 1.0, 0.5, 1.0)
    h2 = b.run_chain(15, 0, 1, 5, 0, 5, h1["r_final"], h1["p_final"], 1.0, first_iter=10)
    assert np.array_equal(full["clusts"], np.vstack([h1["clusts"], h2["clusts"]]))
    assert np.array_equal(full["r"], np.concatenate([h1["r"], h2["r"]]))
    assert np.array_equal(full["loglik"], np.concatenate([h1["loglik"], h2["loglik"]]))
    with pytest.raises(rc.RedClustHIPError):
        b.run_chain(5, 0, 0, 5, 0, 5, 1.0, 0.5, 1.0)          # thin = 0
    with pytest.raises(rc.RedClustHIPError):
        b.run_chain(5, 0, 1, 5, 1, 5, 1.0, 0.5, 1.0)          # numMH > 0 without host matrices
    with pytest.raises(rc.RedClustHIPError):
        b.run_chain(5, 0, 1, 5, 0, 5, -1.0, 0.5, 1.0)
    a.close(); b.close()


def test_runsampler_engines_agree():
    """runsampler(engine="native") and the Python loop produce the same MCMCResult under teacher-forced r / p, and the
    native engine fills every field free-running (the reference's @test_nothrow runs, test/test_sampler.jl)."""
    D, truth = paper(1)
    P = rc.likelihood_hyperparams(D, truth)
    params = rc.PriorHyperparamsList(**{k: P[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma")})
    init = np.random.default_rng(2).integers(1, 11, 100).astype(np.int64)
    rs = 0.8 + 0.02 * np.arange(40); ps = 0.3 + 0.005 * np.arange(40)
    res = {}
    for eng in ("native", "python"):
        res[eng] = rc.runsampler(rc.MCMCData(D), rc.MCMCOptionsList(numiters=40, burnin=8, thin=4, numMH=1), params,
                                 rc.MCMCState(init, 1.0, 0.5), verbose=False, seed=31, rp_trace=(rs, ps), engine=eng)
    a, b = res["native"], res["python"]
    assert all(np.array_equal(x, y) for x, y in zip(a.clusts, b.clusts))
    for f in ("K", "r", "p", "loglik", "logposterior", "posterior_coclustering", "splitmerge_acceptances", "splitmerge_splits"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    free = rc.runsampler(rc.MCMCData(D), rc.MCMCOptionsList(numiters=200, burnin=50, thin=5), params,
                         rc.MCMCState(init, 1.0, 0.5), verbose=False, seed=4)
    assert len(free.clusts) == 30 and np.all(free.K > 0) and np.all(np.isfinite(free.logposterior))
    assert 0 < free.r_acceptance_rate < 1 and np.all((free.p > 0) & (free.p < 1)) and free.mean_iter_time > 0
    assert np.all(np.diag(free.posterior_coclustering) == 1.0)


def test_run_chains_in_the_library_one_chain_over_a_real_rccl_communicator():
    """rc_run_chains with n_chains = 1 — all this pool can run: the chain equals rc_run_chain on a context
    of the caller's, the merge goes through a real RCCL communicator of size 1 (ncclCommInitAll inside the library; the
    unique-id / ncclCommInitRank path is exercised too), and the merged matrix is counts / samples."""
    D, truth = paper(1)
    P = T.likelihood_hyperparams(D, truth)
    init = np.random.default_rng(7).integers(1, 11, 100).astype(np.int64)
    orc, ctx = make(D, P, init)
    ctx.attach_host_matrices(D, orc.logD)
    one = ctx.run_chain(40, 10, 3, 5, 1, 11, 1.0, 0.5, 0.7)
    chains, post, total, ms = rc._lib.run_chains([0], P, init, 40, 10, 3, 5, 1, 11, 1.0, 0.5, 0.7, D=D, logD=orc.logD)
    assert len(chains) == 1 and total == one["num_samples"] == 10 and ms >= 0
    for k in ("clusts", "K", "r", "p", "loglik", "logposterior", "r_acceptances", "splitmerge_acceptances", "splitmerge_splits"):
        assert np.array_equal(chains[0][k], one[k]), k
    assert np.array_equal(post, ctx.cocluster(10)) and np.all(np.diag(post) == 1.0)
    # the communicator layer on the caller's context: size-1 all-reduce leaves the counts as they are
    before = ctx.cocluster_counts().copy()
    for comm in (rc.Comm([0]), rc.Comm([0], rank_offset=0, world_size=1, unique_id=rc.Comm.unique_id())):
        tot, _ = comm.allreduce_counts([ctx], [10])
        assert tot == 10 and np.array_equal(ctx.cocluster_counts(), before)
        comm.close()
    assert rc.library_merge(ctx, 0, 10)[0] == 10
    # one chain per GPU: a device listed twice, a device that does not exist, a split world without the id
    for bad in (dict(device_ids=[0, 0]), dict(device_ids=[99]), dict(device_ids=[0], rank_offset=1, world_size=2)):
        with pytest.raises(rc.RedClustHIPError, match="RC_ERR_ARG"):
            rc.Comm(**bad)
    with pytest.raises(rc.RedClustHIPError, match="twice"):
        rc._lib.run_chains([0, 0], P, init, 5, 0, 1, 5, 0, 1, 1.0, 0.5, 0.7, D=D)
    # the points path and the thin Python caller
    pts = np.random.default_rng(1).normal(size=(60, 3))
    data = rc.MCMCData(pts)
    Pp = rc.likelihood_hyperparams(data.D, np.arange(60) % 4 + 1)
    params = rc.PriorHyperparamsList(**{k: Pp[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma")})
    ch, merged, tot = rc.run_chains_single_process(data, rc.MCMCOptionsList(numiters=20, burnin=4, thin=2, numMH=0), params,
                                                   rc.MCMCState(np.arange(60) % 4 + 1, 1.0, 0.5), [0], base_seed=3)
    assert tot == 8 and merged.shape == (60, 60) and np.all(np.diag(merged) == 1.0) and np.array_equal(merged, merged.T)
    # chains.run_chains — the one-process-per-GPU caller (here a world of one, no process group): runsampler on the rank's
    # context, the merge through the library's communicator, rc_cocluster of the merged counts
    res, merged1, traces = rc.run_chains(data, rc.MCMCOptionsList(numiters=20, burnin=4, thin=2, numMH=0), params,
                                         rc.MCMCState(np.arange(60) % 4 + 1, 1.0, 0.5), base_seed=3)
    assert len(traces) == 1 and traces[0]["rank"] == 0 and np.array_equal(merged1, res.posterior_coclustering)
    assert np.array_equal(merged1, merged)             # same data, init, seed and options as chain 0 above
    ctx.close()


@pytest.mark.parametrize("mode", ["as_written", "intended"])
def test_speculative_loop_equals_the_synchronous_loop(mode):
    """rc_run_chain with numMH > 0 speculates that every split-merge proposal is rejected (proposals decided by worker
    threads on state snapshots, several iterations in flight, splits evaluated off the live state, rollback on acceptance).
    It must reproduce the synchronous loop (RC_CHAIN_PIPELINE=0) bit for bit — here on a chain that moves, with few large
    clusters (many split proposals) and several accepted proposals, at every speculation depth."""
    D, truth = paper(2)
    P = T.likelihood_hyperparams(D, truth)
    init = np.random.default_rng(11).integers(1, 4, 100).astype(np.int64)     # three clusters: P(split proposal) ~ 1/3
    L = np.log(D + np.eye(100))
    runs = {}
    for tag, env in (("sync", {"RC_CHAIN_PIPELINE": "0"}), ("spec", {}), ("spec_shallow", {"RC_CHAIN_DEPTH": "2", "RC_CHAIN_WORKERS": "1"}),
                     ("spec_deep", {"RC_CHAIN_DEPTH": "40", "RC_CHAIN_WORKERS": "6"})):
        for k in ("RC_CHAIN_PIPELINE", "RC_CHAIN_DEPTH", "RC_CHAIN_WORKERS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx = rc.Context(D, logD=L); ctx.set_params(**P); ctx.set_state(init); ctx.cocluster_reset()
        ctx.attach_host_matrices(D, L)
        ch = ctx.run_chain(120, 20, 3, 5, 2, 77, 1.0, 0.5, 0.7, splitmerge=mode)
        ch["cocluster"] = ctx.cocluster(max(ch["num_samples"], 1)); ch["final"] = ctx.get_state()[0]
        runs[tag] = ch
        ctx.close()
    for k in ("RC_CHAIN_PIPELINE", "RC_CHAIN_DEPTH", "RC_CHAIN_WORKERS"):
        os.environ.pop(k, None)
    ref = runs["sync"]
    assert ref["splitmerge_acceptances"].sum() >= 2 and ref["splitmerge_splits"].sum() >= 10 and ref["num_samples"] == 33
    assert len(np.unique(ref["K"])) > 1                                           # the chain moves
    for tag in ("spec", "spec_shallow", "spec_deep"):
        for f in ("clusts", "K", "r", "p", "loglik", "logposterior", "r_acceptances", "splitmerge_acceptances", "splitmerge_splits",
                  "r_all", "p_all", "cocluster", "final", "r_final", "p_final", "num_samples"):
            assert np.array_equal(runs[tag][f], ref[f]), (tag, f)


@pytest.mark.parametrize("thin", [1, 3])
def test_pipelined_loop_without_proposals_equals_the_synchronous_loop(thin):
    """numMH = 0 runs through the pipelined loop as well (snapshots only for the iterations that are recorded, the host part of a
    recorded sample — sortlabels, log-likelihood terms, log-prior — computed by the worker pool): every output equals the
    synchronous loop's (RC_CHAIN_PIPELINE=0), on a chain that moves."""
    D, truth = paper(2)
    P = dict(T.likelihood_hyperparams(D, truth), repulsion=False)
    init = np.random.default_rng(3).integers(1, 9, 100).astype(np.int64)
    runs = {}
    for tag, env in (("sync", {"RC_CHAIN_PIPELINE": "0"}), ("pipe", {}), ("pipe_shallow", {"RC_CHAIN_DEPTH": "1", "RC_CHAIN_WORKERS": "1"})):
        for k in ("RC_CHAIN_PIPELINE", "RC_CHAIN_DEPTH", "RC_CHAIN_WORKERS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx = rc.Context(D, kcap=8); ctx.set_params(**P); ctx.set_state(init); ctx.cocluster_reset()      # (the capacity grows on the way)
        ch = ctx.run_chain(90, 7, thin, 5, 0, 5, 1.0, 0.5, 0.7)
        ch["cocluster"] = ctx.cocluster(max(ch["num_samples"], 1)); ch["final"] = ctx.get_state()[0]
        runs[tag] = ch
        ctx.close()
    for k in ("RC_CHAIN_PIPELINE", "RC_CHAIN_DEPTH", "RC_CHAIN_WORKERS"):
        os.environ.pop(k, None)
    ref = runs["sync"]
    assert ref["num_samples"] == (90 - 7) // thin and len(np.unique(ref["K"])) > 1
    for tag in ("pipe", "pipe_shallow"):
        for f in ("clusts", "K", "r", "p", "loglik", "logposterior", "r_acceptances", "r_all", "p_all", "cocluster", "final", "r_final", "p_final", "num_samples"):
            assert np.array_equal(runs[tag][f], ref[f]), (tag, f)


def test_parameters_changed_between_proposals_invalidate_the_cached_likelihood():
    """rc_set_params between two rc_splitmerge calls with unchanged labels (allowed by the C ABI): the second proposal must
    use block sums / log-likelihood terms of the NEW parameters — it equals the proposal of a fresh context that only ever
    saw the new parameters."""
    D, truth = paper(1)
    P1 = T.likelihood_hyperparams(D, truth)
    P2 = dict(P1, delta1=P1["delta1"] * 1.7, alpha=P1["alpha"] * 0.6, zeta=P1["zeta"] * 1.3)
    init = truth.copy(); init[init == 2] = 1; init[init == 4] = 3
    L = np.log(D + np.eye(100))
    a = rc.Context(D, logD=L); a.set_params(**P1); a.set_state(init); a.attach_host_matrices(D, L)
    b = rc.Context(D, logD=L); b.set_params(**P2); b.set_state(init); b.attach_host_matrices(D, L)
    first = a.splitmerge(1.0, 0.5, 5, 21, 0, 0)          # fills a's caches under P1
    a.set_state(init)                                    # same labels again, whatever the first proposal did
    ll1 = a.loglik()
    a.set_params(**P2)
    assert a.loglik() == b.loglik() != ll1
    for it in range(1, 6):
        ra, rb = a.splitmerge(1.0, 0.5, 5, 21, it, 0), b.splitmerge(1.0, 0.5, 5, 21, it, 0)
        assert ra == rb, (it, ra, rb, first)
        assert np.array_equal(a.get_state()[0], b.get_state()[0]) and a.loglik() == b.loglik()
    a.close(); b.close()
