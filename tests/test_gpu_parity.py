"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the committed
golden vectors.  Integer outputs (labels, sizes, K, fixed-point row sums, co-clustering counts) must match
EXACTLY; loglik / logprior within the stated relative tolerance."""
import os

import numpy as np
import pytest

import oracle_lib as O
import redclust_amd as rc
from helpers import golden_case, load_golden, rp_schedule

pytestmark = pytest.mark.gpu

LL_RTOL = 1e-9   # loglik / logposterior vs the oracle's stable mode (north_star allows 1e-6)
LIT_RTOL = 1e-7  # vs the literal restatement / golden vectors (cancellation noise of the literal formulas)


def make_pair(D, P, init, kcap=0, bits=64):
    orc = O.Oracle(D, P, bits=bits)
    orc.set_state(init)
    ctx = rc.Context(D, logD=orc.logD, kcap=kcap, storage_bits=bits)
    ctx.set_params(**P)
    ctx.set_state(init)
    return orc, ctx


def assert_state_equal(ctx, orc, msg=""):
    clusts, sizes, K = ctx.get_state()
    assert np.array_equal(clusts, orc.clusts), f"labels differ {msg}"
    assert np.array_equal(sizes, orc.sizes), f"sizes differ {msg}"
    assert K == orc.K, f"K differs {msg}"


def test_fixed_point_rowsums_exact():
    g, d = load_golden()
    D, P, init, seed = golden_case(g, d, "d1_random")
    orc, ctx = make_pair(D, P, init)
    onehot = (init[:, None] == np.arange(1, 101)[None, :]).astype(np.int64)
    SD, SL = orc.Dq @ onehot, orc.Lq @ onehot
    for lab in np.unique(init):
        sd, sl, eD, eL = ctx.debug_rowsums(int(lab))
        assert (eD, eL) == (orc.eD, orc.eL)
        assert np.array_equal(sd, SD[:, lab - 1]) and np.array_equal(sl, SL[:, lab - 1])
    ctx.close()


def test_device_logD_matches_host_within_one_quantum():
    g, d = load_golden()
    D, P, init, seed = golden_case(g, d, "d2_truth")
    orc = O.Oracle(D, P)
    ctx = rc.Context(D)  # logD derived on the device (types.jl:155)
    ctx.set_params(**P)
    ctx.set_state(init)
    onehot = (init[:, None] == np.arange(1, 101)[None, :]).astype(np.int64)
    SL = orc.Lq @ onehot
    for lab in np.unique(init):
        sd, sl, eD, eL = ctx.debug_rowsums(int(lab))
        # the derived-logD mode caps its exponent, so compare in real units: ≤ 1 quantum (of the coarser
        # of the two grids) per summed entry, plus what the library's table log may differ from libm's by
        quantum = np.ldexp(1.0, -min(eL, orc.eL))
        err = np.abs(sl * np.ldexp(1.0, -eL) - SL[:, lab - 1] * np.ldexp(1.0, -orc.eL))
        # (... and, per entry, the 2e-13 of rc_qlog's degree-4 log1p and the rounding of D's own entry: 47 significant bits of the largest)
        assert np.max(err) <= np.sum(init == lab) * (quantum + 2e-13 + np.ldexp(1.0, -eD - 1) / D[D > 0].min())
    ctx.close()


@pytest.mark.parametrize("tag", ["d1_truth", "d1_random", "d1_norep", "d1_maxK6", "d1_singletons", "d2_truth",
                                 "d2_random", "d3_truth", "d3_random"])
def test_golden_sweeps(tag):
    """Teacher-forced (r, p) sweeps on the paper datasets: labels/sizes/K exactly equal to the golden vectors
    (independent NumPy transcription) and to the oracle; loglik, logprior, canonical labels too."""
    g, d = load_golden()
    D, P, init, seed = golden_case(g, d, tag)
    orc, ctx = make_pair(D, P, init)
    for t in range(4):
        r, p = float(g["r_seq"][t]), float(g["p_seq"][t])
        ctx.gibbs_sweep(r, p, seed, t)
        orc.sweep_stable(r, p, seed, t)
        assert_state_equal(ctx, orc, f"({tag}, sweep {t})")
        clusts, sizes, K = ctx.get_state()
        assert np.array_equal(clusts, g[f"{tag}_labels"][t]) and np.array_equal(sizes, g[f"{tag}_sizes"][t])
        assert K == int(g[f"{tag}_K"][t])
        assert ctx.sweep_stats()["n_changes"] == orc.last_changes
        ll = ctx.loglik()
        assert abs(ll - orc.loglik_stable()) <= LL_RTOL * max(1.0, abs(ll))
        assert abs(ll - float(g[f"{tag}_loglik"][t])) <= LIT_RTOL * max(1.0, abs(ll))
        lp = ctx.logprior(r, p)
        assert abs(lp - float(g[f"{tag}_logprior"][t])) <= 1e-10 * max(1.0, abs(lp))
        assert abs(lp - orc.logprior(r, p)) <= 1e-12 * max(1.0, abs(lp))
        canon = ctx.record_sample(True)
        assert np.array_equal(canon, g[f"{tag}_canon"][t])
    ctx.close()


@pytest.mark.parametrize("n,K,seed", [(257, 5, 1), (1000, 12, 2), (2000, 20, 3)])
def test_synthetic_moving_and_stationary(n, K, seed):
    """generatemixture-distributed data; random init (many label changes per sweep, births and deaths) and
    the generating labels (stationary); 6 sweeps each, exact trajectory vs the oracle."""
    data = rc.generatemixture(n, K, seed=seed, sigma=0.2)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    rng = np.random.default_rng(seed)
    for init in (rng.integers(1, K + 1, size=n).astype(np.int64), truth):
        orc, ctx = make_pair(D, P, init)
        for t in range(6):
            r, p = rp_schedule(t)
            ctx.gibbs_sweep(r, p, 1000 + seed, t)
            orc.sweep_stable(r, p, 1000 + seed, t)
            assert_state_equal(ctx, orc, f"(n={n}, sweep {t})")
            st = ctx.sweep_stats()
            assert st["n_changes"] == orc.last_changes and 1 <= st["n_rounds"] <= 2 * st["n_changes"] + 2
        ll = ctx.loglik()
        assert abs(ll - orc.loglik_stable()) <= LL_RTOL * abs(ll)
        assert abs(ll - orc.loglik_literal()) <= 1e-6 * abs(ll)
        ctx.close()


@pytest.mark.parametrize("tag", ["d1_random", "d2_truth", "d1_singletons"])
def test_storage32_golden_cases(tag):
    """32-bit fixed-point storage (config-5 style): exact parity with the oracle's stable mode on the same 32-bit
    quantised matrices; row sums exact; loglik within 1e-9 of that oracle and within 1e-6 of the Float64 literal."""
    g, d = load_golden()
    D, P, init, seed = golden_case(g, d, tag)
    orc, ctx = make_pair(D, P, init, bits=32)
    assert int(np.abs(orc.Dq).max()) < 2 ** 30 and int(np.abs(orc.Lq).max()) < 2 ** 30
    onehot = (init[:, None] == np.arange(1, 101)[None, :]).astype(np.int64)
    for lab in np.unique(init)[:5]:
        sd, sl, eD, eL = ctx.debug_rowsums(int(lab))
        assert (eD, eL) == (orc.eD, orc.eL)
        assert np.array_equal(sd, (orc.Dq @ onehot)[:, lab - 1]) and np.array_equal(sl, (orc.Lq @ onehot)[:, lab - 1])
    for t in range(6):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, seed, t)
        orc.sweep_stable(r, p, seed, t)
        assert_state_equal(ctx, orc, f"(32-bit {tag}, sweep {t})")
    ll = ctx.loglik()
    assert abs(ll - orc.loglik_stable()) <= LL_RTOL * max(1.0, abs(ll))
    assert abs(ll - orc.loglik_literal()) <= 1e-6 * max(1.0, abs(ll))
    ctx.close()


def test_storage32_synthetic():
    data = rc.generatemixture(1500, 40, seed=5, sigma=0.2)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    init = np.random.default_rng(5).integers(1, 41, size=1500).astype(np.int64)
    orc, ctx = make_pair(D, P, init, bits=32, kcap=512)  # the first sweep from a random init shatters into ~300 clusters
    for t in range(5):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, 77, t)
        orc.sweep_stable(r, p, 77, t)
        assert_state_equal(ctx, orc, f"(32-bit synthetic, sweep {t})")
    ctx.close()


def test_literal_restatement_trajectory():
    """GPU vs the LITERAL restatement (reference formulas as written) on N=100: same labels for 30 sweeps."""
    g, d = load_golden()
    D, P, init, seed = golden_case(g, d, "d1_random")
    orc, ctx = make_pair(D, P, init)
    for t in range(30):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, 5, t)
        orc.sweep_literal(r, p, 5, t)
        assert_state_equal(ctx, orc, f"(literal, sweep {t})")
    ctx.close()


def test_cocluster_counts_and_posterior():
    g, d = load_golden()
    D, P, init, seed = golden_case(g, d, "d2_random")
    orc, ctx = make_pair(D, P, init)
    ctx.cocluster_reset()
    ref = np.zeros((100, 100), np.uint32)
    S = 7
    for t in range(S):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, seed, t)
        orc.sweep_stable(r, p, seed, t)
        ctx.record_sample(False)
        orc.L.orc_cocluster_add(100, orc.clusts, ref.reshape(-1))
    cnt = ctx.cocluster_counts()
    assert np.array_equal(cnt, ref)
    post = ctx.cocluster(S)
    assert np.array_equal(post, ref / S)  # counts ./ numsamples, mcmc.jl:560
    assert np.all(np.diag(post) == 1.0) and np.array_equal(post, post.T)
    ctx.close()


def test_cocluster_many_samples_batched():
    """More recorded samples than one accumulation batch (32): full flushes + a partial one, odd n (padding)."""
    g, d = load_golden()
    D, P, init, seed = golden_case(g, d, "d1_random")
    n = 97
    D = np.ascontiguousarray(D[:n, :n]); init = init[:n].copy()
    orc, ctx = make_pair(D, P, init)
    ctx.cocluster_reset()
    ref = np.zeros((n, n), np.uint32)
    S = 75
    for t in range(S):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, seed, t)
        orc.sweep_stable(r, p, seed, t)
        ctx.record_sample(t % 10 == 0)
        orc.L.orc_cocluster_add(n, orc.clusts, ref.reshape(-1))
        if t == 40:
            assert np.array_equal(ctx.cocluster_counts(), ref)  # reading flushes the queue; accumulation continues
    assert_state_equal(ctx, orc)
    assert np.array_equal(ctx.cocluster_counts(), ref)
    assert np.array_equal(ctx.cocluster(S), ref / S)
    ctx.cocluster_reset()
    assert not ctx.cocluster_counts().any()
    ctx.close()


def test_edge_cases_and_errors():
    g, d = load_golden()
    D, P, init, seed = golden_case(g, d, "d1_truth")
    # asymmetric D is rejected like MCMCData (types.jl:149-151)
    bad = D.copy(); bad[0, 1] += 1e-9
    with pytest.raises(rc.RedClustHIPError, match="symmetric"):
        rc.Context(bad)
    z = D.copy(); z[2, 3] = z[3, 2] = 0.0
    with pytest.raises(rc.RedClustHIPError, match="positive"):
        rc.Context(z)
    # a caller's logD must be symmetric as well: the symmetric kernels read its upper triangle only
    Lbad = np.log(D + np.eye(100)); Lbad[5, 6] += 1e-9
    with pytest.raises(ValueError, match="logD must be symmetric"):
        rc.Context(D, logD=Lbad)
    # the zero-distance decision (DESIGN.md "Zero distances"): a ValueError like the reference's ArgumentError class of
    # input errors, raised by the library for a matrix and for duplicate points alike, and by the MCMCData glue
    with pytest.raises(ValueError, match="jitter"):
        rc.Context(z)
    pts = np.random.default_rng(0).normal(size=(50, 3)); pts[17] = pts[4]
    with pytest.raises(ValueError, match="positive"):
        rc.Context.from_points(pts)
    with pytest.raises(ValueError, match="jitter"):
        rc.MCMCData(z)
    ctx = rc.Context(D)
    with pytest.raises(rc.RedClustHIPError, match="RC_ERR_STATE"):
        ctx.gibbs_sweep(1.0, 0.5, 1, 0)
    ctx.set_params(**P)
    with pytest.raises(rc.RedClustHIPError, match="outside 1..n"):
        ctx.set_state(np.zeros(100, np.int64))
    ctx.set_state(init)
    with pytest.raises(rc.RedClustHIPError, match="RC_ERR_ARG"):
        ctx.gibbs_sweep(-1.0, 0.5, 1, 0)
    with pytest.raises(rc.RedClustHIPError, match="RC_ERR_ARG"):
        ctx.gibbs_sweep(1.0, 1.0, 1, 0)
    ctx.close()
    # (capacity: the slot tables grow on demand — tests/test_gpu_capacity.py, which also covers the RC_KCAP_FIXED error path)
    # tiny problems: n = 1, 2, 3
    for n in (1, 2, 3):
        Dn = np.ascontiguousarray(D[:n, :n])
        orc, c2 = make_pair(Dn, P, np.ones(n, np.int64))
        for t in range(5):
            c2.gibbs_sweep(1.0, 0.5, 11, t)
            orc.sweep_stable(1.0, 0.5, 11, t)
            assert_state_equal(c2, orc, f"(n={n})")
        c2.close()


def test_runsampler_surface():
    """runsampler(data, options, params, init) with numMH = 0 ('pure Gibbs', test/test_sampler.jl:7): fills every
    MCMCResult field; teacher-forced r/p reproduce the oracle's recorded trace exactly."""
    g, d = load_golden()
    D, P, init, seed = golden_case(g, d, "d3_random")
    params = rc.PriorHyperparamsList(**{k: P[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma")})
    opts = rc.MCMCOptionsList(numiters=40, burnin=10, thin=3, numMH=0)
    rs = np.array([rp_schedule(t)[0] for t in range(40)]); ps = np.array([rp_schedule(t)[1] for t in range(40)])
    res = rc.runsampler(rc.MCMCData(D), opts, params, rc.MCMCState(init, 1.0, 0.5), verbose=False, seed=77,
                        rp_trace=(rs, ps))
    assert opts.numsamples == 10 and len(res.clusts) == 10
    orc = O.Oracle(D, P); orc.set_state(init)
    j = 0
    ref_counts = np.zeros((100, 100), np.uint32)
    for i in range(1, 41):
        orc.sweep_stable(rs[i - 1], ps[i - 1], 77, i - 1)
        if i > 10 and (i - 10) % 3 == 0:
            assert np.array_equal(res.clusts[j], orc.sortlabels()) and res.K[j] == orc.K
            assert abs(res.loglik[j] - orc.loglik_stable()) <= LL_RTOL * abs(res.loglik[j])
            lp = orc.loglik_stable() + orc.logprior(rs[i - 1], ps[i - 1])
            assert abs(res.logposterior[j] - lp) <= 1e-6 * abs(lp)   # north_star tolerance on log-posterior
            orc.L.orc_cocluster_add(100, orc.clusts, ref_counts.reshape(-1))
            j += 1
    assert j == 10 and np.array_equal(res.posterior_coclustering, ref_counts / 10)
    # free-running r/p: smoke test as the reference's own sampler tests (@test_nothrow)
    res2 = rc.runsampler(rc.MCMCData(D), rc.MCMCOptionsList(numiters=60, numMH=0), params,
                         rc.MCMCState(init, 1.0, 0.5), verbose=False, seed=3)
    assert res2.K.shape == (48,) and np.isfinite(res2.logposterior).all() and 0 <= res2.r_acceptance_rate <= 1
    assert res2.r_ess > 0 and res2.mean_iter_time > 0 and res2.posterior_coclustering.shape == (100, 100)
    assert len(res2.r_acf) == 18 and res2.r_acf[0] == 1.0  # autocor default lags 0:min(n-1, round(10·log10 n))


@pytest.mark.parametrize("tag", ["d1_merged", "d3_merged", "d3_splitup", "d2_truth"])
def test_splitmerge_as_written(tag):
    """sample_labels! as written (numMH=3, numGibbs=5; quirks Q1–Q3): rc_splitmerge + checkpoint/restore + sweep
    against the oracle (literal restricted scans, stable loglik and sweep) and the golden flags; the S table must
    survive every apply / revert bit-exactly (the sweeps after it stay on the oracle's trajectory)."""
    from test_splitmerge_cpu import mh_case
    gm, D, P, init, seed = mh_case(tag)
    orc, ctx = make_pair(D, P, init)
    ctx.attach_host_matrices(D, orc.logD)
    for it in range(20):
        r, p = 1.0 + 0.05 * it, 0.5
        na, acc_ref, spl_ref = orc.sample_labels(r, p, 3, 5, seed, it, mode=1)
        ctx.checkpoint()
        acc, spl = [], []
        for mh in range(3):
            a, s = ctx.splitmerge(r, p, 5, seed, it, mh)
            acc.append(a); spl.append(s)
        if any(acc):
            ctx.restore()
        else:
            ctx.gibbs_sweep(r, p, seed, it)
        assert acc == list(acc_ref) and spl == list(spl_ref), (tag, it)
        assert acc == list(gm[f"{tag}_accept"][it]) and spl == list(gm[f"{tag}_split"][it])
        assert_state_equal(ctx, orc, f"({tag}, iteration {it})")
        assert abs(ctx.loglik() - orc.loglik_stable()) <= LL_RTOL * max(1.0, abs(orc.loglik_stable()))
    ctx.close()


def test_splitmerge_intended_mode_and_runsampler_defaults():
    """'intended' semantics (accepted proposals are kept and swept) proposal by proposal against the oracle, then the
    reference's default options (numMH = 1, numGibbs = 5) through runsampler in both modes."""
    from test_splitmerge_cpu import mh_case
    gm, D, P, init, seed = mh_case("d1_merged")
    orc, ctx = make_pair(D, P, init)
    ctx.attach_host_matrices(D, orc.logD)
    accepted = 0
    for it in range(25):
        for mh in range(2):
            inf = orc.mh_proposal(1.0, 0.5, 5, seed, it, mh, mode=1)   # replaces the oracle's state on acceptance
            a, s = ctx.splitmerge(1.0, 0.5, 5, seed, it, mh)
            assert (a, s) == (bool(inf.accept), bool(inf.split)), (it, mh)
            accepted += a
            assert_state_equal(ctx, orc, f"(intended, {it}, {mh})")
        ctx.gibbs_sweep(1.0, 0.5, seed, it)
        orc.sweep_stable(1.0, 0.5, seed, it)
        assert_state_equal(ctx, orc, f"(intended sweep {it})")
    assert accepted >= 1
    ctx.close()
    params = rc.PriorHyperparamsList(**{k: P[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma")})
    for mode in ("as_written", "intended"):
        res = rc.runsampler(rc.MCMCData(D), rc.MCMCOptionsList(numiters=60), params, rc.MCMCState(init, 1.0, 0.5),
                            verbose=False, seed=5, splitmerge=mode)
        assert res.splitmerge_acceptances.shape == (60,) and res.splitmerge_splits.shape == (60,)
        assert 0.0 <= res.splitmerge_acceptance_rate <= 1.0 and np.isfinite(res.logposterior).all()
        assert res.K.shape == (48,) and res.posterior_coclustering.shape == (100, 100)


def test_device_pairwise_distances():
    """MCMCData(points) on the device (rc_create_from_points, types.jl:159-162) vs the oracle's restatement of
    Distances.jl's pairwise Euclidean: exact symmetry and zero diagonal, values within 1e-12 relative (the dot
    products' summation order is unspecified in the reference too), and the same sweep trajectory as a context
    built from the resulting matrices."""
    data = rc.generatemixture(777, 9, seed=4, sigma=0.2, dim=37)
    pts, truth = data["points"], data["clusts"]
    ref = np.zeros((777, 777))
    O.lib().orc_pairwise_euclidean(777, 37, np.ascontiguousarray(pts), ref.reshape(-1))
    ctx = rc.Context.from_points(pts)
    Dd, Ld = ctx.get_matrix(0), ctx.get_matrix(1)
    assert np.array_equal(Dd, Dd.T) and np.all(np.diag(Dd) == 0) and np.all(np.diag(Ld) == 0)
    assert np.max(np.abs(Dd - ref)) <= 1e-12 * ref.max()
    off = ~np.eye(777, dtype=bool)
    assert np.max(np.abs(Ld[off] - np.log(ref[off]))) <= 1e-11
    P = rc.likelihood_hyperparams(ref, truth)
    ctx2 = rc.Context(Dd, logD=Ld)           # re-quantises to the same integers
    init = np.random.default_rng(4).integers(1, 10, size=777).astype(np.int64)
    for c in (ctx, ctx2):
        c.set_params(**P); c.set_state(init)
    for t in range(5):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, 3, t); ctx2.gibbs_sweep(r, p, 3, t)
        a, b = ctx.get_state(), ctx2.get_state()
        assert np.array_equal(a[0], b[0]) and a[2] == b[2]
    assert ctx.loglik() == ctx2.loglik()
    ctx.close(); ctx2.close()
    # the host surface: MCMCData(points) + runsampler never builds the n×n matrix on the host for numMH = 0
    md = rc.MCMCData(pts)
    params = rc.PriorHyperparamsList(**{k: P[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma")})
    res = rc.runsampler(md, rc.MCMCOptionsList(numiters=20, numMH=0), params, rc.MCMCState(truth, 1.0, 0.5), verbose=False)
    assert md._D is None and res.K.shape == (16,) and np.isfinite(res.logposterior).all()
    assert np.max(np.abs(md.D - ref)) <= 1e-12 * ref.max() and np.array_equal(md.D, md.D.T)


@pytest.mark.parametrize("n,bits", [(333, 64), (1029, 64), (1200, 32), (2050, 32)])
def test_symmetric_and_full_row_reduction_agree_exactly(n, bits):
    """k_bulk_sym (reads the upper triangle only) and k_bulk (reads everything) must give the same row-sum table bit
    for bit for sorted AND arbitrary labels, odd sizes, both storage widths — and the same sweeps."""
    data = rc.generatemixture(n, 9, seed=n, sigma=0.2, dim=12)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    rng = np.random.default_rng(n)
    labelings = [truth, rng.integers(1, 10, size=n).astype(np.int64), rng.permutation(n).astype(np.int64) % 7 + 1]
    ctxs = {}
    for which in ("perm", "sym"):
        c = rc.Context(D, storage_bits=bits, kcap=64)
        c.set_params(**P); c.set_bulk_kernel(which)
        ctxs[which] = c
    for lab in labelings:
        for c in ctxs.values():
            c.set_state(lab)
        for k in np.unique(lab):
            a, b = ctxs["perm"].debug_rowsums(int(k)), ctxs["sym"].debug_rowsums(int(k))
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (n, bits, k)
        assert ctxs["perm"].bulk_kernel_info()[0] == "k_bulk" and ctxs["sym"].bulk_kernel_info()[0] == "k_bulk_sym"
        for t in range(3):
            for c in ctxs.values():
                c.gibbs_sweep(1.0, 0.5, 5, t)
            sa, sb = ctxs["perm"].get_state(), ctxs["sym"].get_state()
            assert np.array_equal(sa[0], sb[0]) and sa[2] == sb[2]
        assert ctxs["perm"].loglik() == ctxs["sym"].loglik()
    for c in ctxs.values():
        c.close()


def test_internal_layout_shuffled_points():
    """Points arrive in random order: the library lays them out cluster-contiguously inside (so the symmetric row
    reduction is chosen), yet sweeps, row sums, matrices and the co-clustering matrix all come back in the caller's
    order and match the oracle / a context with the layout disabled."""
    n, K = 1500, 7
    data = rc.generatemixture(n, K, seed=77, sigma=0.3, dim=10)
    rng = np.random.default_rng(3)
    sh = rng.permutation(n)
    D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)])
    truth = data["clusts"][sh]
    assert np.count_nonzero(np.diff(truth)) > n // 2          # thoroughly unsorted
    P = rc.likelihood_hyperparams(D, truth)
    orc = O.Oracle(D, P)
    ctx = rc.Context(D, kcap=64, logD=orc.logD)
    ctx.set_params(**P)
    ctx.set_state(truth)
    assert np.allclose(ctx.get_matrix(0), D, rtol=0, atol=np.ldexp(1.0, -40) * D.max())
    orc.set_state(truth)
    for k in (1, 4, 7):
        sd, sl = ctx.debug_rowsums(k)[:2]
        m = truth == k
        assert np.array_equal(sd, orc.Dq[:, m].sum(axis=1)) and np.array_equal(sl, orc.Lq[:, m].sum(axis=1))
    ctx.cocluster_reset()
    acc = np.zeros((n, n))
    for t in range(6):
        ctx.gibbs_sweep(1.1, 0.45, 9, t)
        orc.sweep_stable(1.1, 0.45, 9, t)
        lab, sizes, Kc = ctx.get_state()
        assert np.array_equal(lab, orc.clusts) and np.array_equal(sizes, orc.sizes) and Kc == orc.K
        ctx.record_sample(False)
        acc += lab[:, None] == lab[None, :]
        if t == 0:   # fresh layout: K label runs, so the symmetric kernel was chosen
            assert ctx.bulk_kernel_info()[0] == "k_bulk_sym"
    assert np.array_equal(ctx.cocluster(6), acc / 6)
    assert abs(ctx.loglik() - orc.loglik_stable()) <= 1e-9 * abs(orc.loglik_stable())
    ctx.close()


def test_automatic_relayout_is_invisible():
    """Overlapping clusters in shuffled order: label movement fragments the internal layout, the library re-lays the
    points out on its own, and the chain is bit-identical to one that keeps the caller's order (RC_NO_RELAYOUT=1)."""
    n, K = 2048, 4
    data = rc.generatemixture(n, K, seed=5, sigma=0.6, dim=6)
    sh = np.random.default_rng(8).permutation(n)
    D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)])
    truth = data["clusts"][sh]
    P = dict(rc.likelihood_hyperparams(D, truth), maxK=12)
    ctx = rc.Context(D, kcap=64)
    ctx.set_params(**P)
    ctx.set_state(truth)
    ref = rc.Context(D, kcap=64)
    ref.set_params(**P)
    ref.set_bulk_kernel("perm")            # forced kernel: never re-lays out after rc_set_state
    ref.set_state(truth)
    l0 = ctx.layout_info()[0]
    moved = 0
    for t in range(120):
        ctx.gibbs_sweep(1.0, 0.5, 3, t, blocking=False)
        ref.gibbs_sweep(1.0, 0.5, 3, t, blocking=False)
        if t % 10 == 9:
            a, b = ctx.get_state(), ref.get_state()
            cnt = np.bincount(a[0], minlength=n + 1)[1:]
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], (
                t, "labels equal" if np.array_equal(a[0], b[0]) else "labels differ",
                "ctx sizes consistent with its labels" if np.array_equal(a[1], cnt) else ("ctx sizes INCONSISTENT", np.flatnonzero(a[1] != cnt), a[1][a[1] != cnt], cnt[a[1] != cnt]),
                "ref sizes consistent with its labels" if np.array_equal(b[1], np.bincount(b[0], minlength=n + 1)[1:]) else "ref sizes INCONSISTENT",
                ctx.layout_info(), ctx.sweep_stats(), ref.sweep_stats(), a[2], b[2])
            moved += ctx.sweep_stats()["n_changes"]
    assert moved > 0
    assert ctx.loglik() == ref.loglik()
    assert ctx.layout_info()[0] > l0, "expected at least one automatic re-layout (runs=%d)" % ctx.layout_info()[1]
    ctx.close(); ref.close()


def test_relayout_between_n64_and_n36_clusters():
    """The moving regime with MANY clusters (n = 4096, K ≈ 75: between n/64 and n/36) fragments the layout beyond the n/32 label runs
    the symmetric row reduction accepts (≈ 240 runs: the full-read kernel takes over).  A fresh layout has K plus a few runs again, so the
    library re-lays the points out there too — after 128 sweeps at the earliest, with a doubling interval (round 4; before, a chain
    that passed K = n/64 stayed on the full-read kernel for good) — and the chain is bit-identical to one that never re-lays out."""
    n, K = 4096, 25
    data = rc.generatemixture(n, K, seed=4, sigma=0.22)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    ctx = rc.Context(D); ctx.set_params(**P); ctx.set_state(truth)
    ref = rc.Context(D); ref.set_params(**P)
    ref.set_bulk_kernel("perm")            # forced kernel: never re-lays out after rc_set_state
    ref.set_state(truth)
    seen_far = False
    for t in range(280):
        ctx.gibbs_sweep(1.0, 0.5, 9, t, blocking=False)
        ref.gibbs_sweep(1.0, 0.5, 9, t, blocking=False)
        if t % 40 == 39:
            a, b = ctx.get_state(), ref.get_state()
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], (t, ctx.layout_info(), ctx.sweep_stats(), ref.sweep_stats())
            layouts, runs = ctx.layout_info()
            if layouts == 0:
                seen_far = seen_far or (a[2] * 64 > n and a[2] * 36 <= n and runs * 32 > n)
    assert seen_far, ("the chain never was in the zone this test is about", ctx.layout_info(), ctx.sweep_stats())
    layouts, runs = ctx.layout_info()
    assert layouts >= 1 and runs * 32 <= n and ctx.bulk_kernel_name().startswith("k_bulk_syml2"), (layouts, runs, ctx.bulk_kernel_name())
    assert ctx.loglik() == ref.loglik()
    ctx.close(); ref.close()


@pytest.mark.parametrize("variant", ["plain", "maxK", "norep", "bits32"])
def test_long_trajectory_with_continual_movement(variant):
    """150 sweeps on overlapping clusters (σ large: labels keep moving every sweep — batches, violations, births and
    deaths in the resolver) against the oracle, state compared after every sweep."""
    n, K = 311, 6
    data = rc.generatemixture(n, K, seed=21, sigma=0.55, dim=8)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    bits = 64
    if variant == "maxK":
        P = dict(P, maxK=9)
    elif variant == "norep":
        P = dict(P, repulsion=False)
    elif variant == "bits32":
        bits = 32
    init = np.random.default_rng(3).integers(1, K + 1, size=n).astype(np.int64)
    orc, ctx = make_pair(D, P, init, kcap=320, bits=bits)
    moved = 0
    for t in range(150):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, 2024, t, blocking=(t % 3 != 0))
        orc.sweep_stable(r, p, 2024, t)
        assert_state_equal(ctx, orc, f"({variant}, sweep {t})")
        st = ctx.sweep_stats()
        assert st["n_changes"] == orc.last_changes
        moved += st["n_changes"]
    assert moved > 150  # the chain really keeps moving
    assert abs(ctx.loglik() - orc.loglik_stable()) <= LL_RTOL * max(1.0, abs(orc.loglik_stable()))
    ctx.close()


def test_incremental_mode_is_bit_identical():
    """RC_MODE_INCREMENTAL (row-sum table maintained by exact corrections only) vs RC_MODE_FULL (recomputed every
    sweep) vs the oracle: same labels every sweep, identical loglik bits, identical row sums; switching modes
    mid-run and split–merge moves in incremental mode keep the trajectory."""
    data = rc.generatemixture(1200, 15, seed=9, sigma=0.22)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    init = np.random.default_rng(9).integers(1, 16, size=1200).astype(np.int64)
    orc, full = make_pair(D, P, init, kcap=512)
    inc = rc.Context(D, logD=orc.logD, kcap=512)
    inc.set_params(**P); inc.set_state(init); inc.set_mode("incremental")
    mixed = rc.Context(D, logD=orc.logD, kcap=512)
    mixed.set_params(**P); mixed.set_state(init)
    for t in range(12):
        r, p = rp_schedule(t)
        if t == 4:
            mixed.set_mode("incremental")
        if t == 9:
            mixed.set_mode("full")
        for c in (full, inc, mixed):
            c.gibbs_sweep(r, p, 31, t)
        orc.sweep_stable(r, p, 31, t)
        for c in (full, inc, mixed):
            assert_state_equal(c, orc, f"(mode test, sweep {t})")
        assert full.loglik() == inc.loglik() == mixed.loglik()
    lab = int(orc.clusts[0])
    a, b = full.debug_rowsums(lab), inc.debug_rowsums(lab)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # split–merge proposals (apply / revert of moves) on the incrementally maintained table
    inc.attach_host_matrices(D, orc.logD)
    for it in range(6):
        inf = orc.mh_proposal(1.0, 0.5, 3, 8, it, 0, mode=1)
        a_, s_ = inc.splitmerge(1.0, 0.5, 3, 8, it, 0)
        assert (a_, s_) == (bool(inf.accept), bool(inf.split))
        inc.gibbs_sweep(1.0, 0.5, 8, 100 + it); orc.sweep_stable(1.0, 0.5, 8, 100 + it)
        assert_state_equal(inc, orc, f"(incremental + MH, {it})")
    for c in (full, inc, mixed):
        c.close()


def test_full_size_properties():
    """BASELINE config 3 (N=8192, K=50): size-independent properties — Σ sizes = n, K = #non-empty, checksum of
    the row-sum table (Σ_k S[k][i] = Σ_j X[i,j] for D and logD, exact in fixed point), stationarity of the generating
    labels, async == blocking, co-clustering diagonal/symmetry.  (The oracle comparison at this size is
    tests/test_gpu_headline.py; config 5 is there too.)"""
    n, K = 8192, 50
    data = rc.generatemixture(n, K, seed=1)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    ctx = rc.Context(D, kcap=128)
    ctx.set_params(**P)
    ctx.set_state(truth)
    tot_d = np.zeros(n, np.int64); tot_l = np.zeros(n, np.int64)
    for lab in np.unique(truth):
        sd, sl, eD, eL = ctx.debug_rowsums(int(lab))
        tot_d += sd; tot_l += sl
    ref_d = np.rint(np.ldexp(D, eD)).astype(np.int64).sum(axis=1)
    assert np.array_equal(tot_d, ref_d)
    ref_l = np.rint(np.ldexp(ctx.get_matrix(1), eL)).astype(np.int64).sum(axis=1)   # the device's (derived) logD = q·2^-eL exactly
    assert np.array_equal(tot_l, ref_l)
    for t in range(3):
        ctx.gibbs_sweep(1.0, 0.5, 42, t)
    c1, s1, K1 = ctx.get_state()
    assert s1.sum() == n and K1 == np.sum(s1 > 0) and np.array_equal(np.bincount(c1, minlength=n + 1)[1:], s1)
    ll1 = ctx.loglik()
    # same three sweeps enqueued without host synchronisation in between
    ctx.set_state(truth)
    for t in range(3):
        ctx.gibbs_sweep(1.0, 0.5, 42, t, blocking=False)
    ctx.synchronize()
    c2, s2, K2 = ctx.get_state()
    assert np.array_equal(c1, c2) and K1 == K2 and ctx.loglik() == ll1
    # random init: a moving sweep keeps the invariants
    init = np.random.default_rng(0).integers(1, K + 1, size=n).astype(np.int64)
    ctx.set_state(init)
    ctx.gibbs_sweep(1.0, 0.5, 43, 0)
    c3, s3, K3 = ctx.get_state()
    assert s3.sum() == n and K3 == np.sum(s3 > 0) and ctx.sweep_stats()["n_changes"] > 0
    ctx.cocluster_reset()
    ctx.record_sample(False); ctx.record_sample(False)
    post = ctx.cocluster(2)
    assert np.all(np.diag(post) == 1.0) and np.array_equal(post, post.T)
    assert np.array_equal(post == 1.0, c3[:, None] == c3[None, :])
    ctx.close()


def test_within_between_split_from_block_sums():
    """fitprior's A / B split (prior.jl:73-75) under given labels from the device block sums == a host pass over D."""
    data = rc.generatemixture(700, 6, seed=12, sigma=0.3, dim=8)
    sh = np.random.default_rng(0).permutation(700)
    D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)]); lab = data["clusts"][sh]
    ctx = rc.Context(D)
    P_dev = rc.likelihood_hyperparams_device(ctx, lab)
    P_host = rc.likelihood_hyperparams(D, lab)
    w = ctx.within_between()
    iu = np.triu_indices(700, 1)
    same = (lab[:, None] == lab[None, :])[iu]
    assert w["count_within"] == int(same.sum()) and w["count_between"] == int((~same).sum())
    assert np.isclose(w["sum_within"], D[iu][same].sum(), rtol=1e-12) and np.isclose(w["sum_between"], D[iu][~same].sum(), rtol=1e-12)
    assert np.isclose(w["sumlog_within"], np.log(D[iu][same]).sum(), rtol=1e-10)
    assert np.isclose(w["sumlog_between"], np.log(D[iu][~same]).sum(), rtol=1e-10)
    for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma"):
        assert np.isclose(P_dev[k], P_host[k], rtol=1e-9), k
    ctx.close()


@pytest.mark.parametrize("cache", ["0", "1"])
def test_pruned_candidates_change_no_draw(cache):
    """The resolver skips the Gumbel noise of candidates that provably cannot win (noise <= 36.74; a cheap upper bound of the
    noise-free score first, the exact score second; pruned entries NaN-tagged in the score cache).  Forced on in every sweep,
    forced off, and the oracle: the same moving chain (births, deaths, singletons), with the score cache off and always on."""
    data = rc.generatemixture(700, 6, seed=5, sigma=0.45)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    init = np.random.default_rng(2).integers(1, 40, 700).astype(np.int64)
    keys = ("RC_SCORE_CACHE",)
    saved = {k: os.environ.get(k) for k in keys}
    try:
        os.environ["RC_SCORE_CACHE"] = cache
        ctx = rc.Context(D)
        ctx.set_params(**P)
        L = ctx.get_matrix(1)
        ctx.set_state(init)
        eD, eL = ctx.debug_rowsums(int(init[0]))[2:4]
        orc = O.Oracle(ctx.get_matrix(0), P, logD=L, eL=eL, eD=eD)
        orc.set_state(init)
        other = rc.Context(D)
        other.set_params(**P)
        other.set_state(init)
        ctx.set_option("prune", 1)        # in every sweep (rc_set_option: a per-context option, nothing reads the environment per sweep)
        other.set_option("prune", 0)      # never
        moved = 0
        for t in range(12):
            r, p = rp_schedule(t)
            ctx.gibbs_sweep(r, p, 31, t)
            other.gibbs_sweep(r, p, 31, t)
            orc.sweep_stable(r, p, 31, t)
            a, b = ctx.get_state(), other.get_state()
            assert np.array_equal(a[0], orc.clusts) and np.array_equal(b[0], orc.clusts), (cache, t)
            assert a[2] == b[2] == orc.K and ctx.sweep_stats()["n_changes"] == other.sweep_stats()["n_changes"] == orc.last_changes
            moved += orc.last_changes
        assert moved > 100
        ctx.close(); other.close()
    finally:
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


@pytest.mark.parametrize("mode", ["full", "incremental"])
def test_point_cache_in_lds_changes_no_draw(mode):
    """k_resolve keeps the internal indices and slots of each workgroup's points in LDS while a workgroup owns at most four
    chunks (n <= 128 x #CUs) and reads pi[] / slot_of[] from global memory in every pass beyond that.  The second path, which no
    test size reaches, is forced with rc_set_option("lds_point_cache", 0): the same moving chain (births, deaths, renames) as
    the default path and as the oracle, in both modes."""
    data = rc.generatemixture(900, 8, seed=9, sigma=0.4)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    init = np.random.default_rng(4).integers(1, 60, 900).astype(np.int64)
    ctx, other = rc.Context(D), rc.Context(D)
    for c in (ctx, other):
        c.set_params(**P); c.set_state(init); c.set_mode(mode)
    other.set_option("lds_point_cache", 0)
    eD, eL = ctx.debug_rowsums(int(init[0]))[2:4]
    orc = O.Oracle(ctx.get_matrix(0), P, logD=ctx.get_matrix(1), eL=eL, eD=eD)
    orc.set_state(init)
    moved = 0
    for t in range(10):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, 77, t); other.gibbs_sweep(r, p, 77, t, blocking=bool(t & 1)); orc.sweep_stable(r, p, 77, t)
        a, b = ctx.get_state(), other.get_state()
        assert np.array_equal(a[0], orc.clusts) and np.array_equal(b[0], orc.clusts), (mode, t)
        assert np.array_equal(a[1], orc.sizes) and a[2] == b[2] == orc.K
        moved += orc.last_changes
    assert moved > 100
    ctx.close(); other.close()
