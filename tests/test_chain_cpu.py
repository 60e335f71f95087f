"""CPU checks of the scalar updates (sample_r / sample_p, src/mcmc.jl:80-155) and the runsampler loop restatement:
oracle vs NumPy transcription (exact), the product's host code vs oracle (exact; rc_scalar_updates needs no GPU),
distributional correctness of the build's own samplers, and the golden chains."""
import os

import numpy as np
import pytest
from scipy import stats
from scipy.special import gammaln

import np_transcription as T
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
SIZES = np.array([10, 20, 5, 65], np.int64)


def test_scalar_uniform_stream():
    for seed, it, kind, d in ((1, 0, 0, 0), (2 ** 40 + 3, 2 ** 33 + 5, 1, 17)):
        s = T.ScalarStream(seed, it, kind)
        s.draw = d
        assert O.lib().orc_scalar_uniform(seed, it, kind, d) == s.uniform()


def test_oracle_vs_transcription_exact():
    r = r2 = 1.0
    nacc = 0
    for it in range(400):
        a, acc = O.sample_r(2 ** 40 + 3, it, r, 0.4, SIZES, 1.5, 0.7, 0.8)
        b, accb = T.sample_r(2 ** 40 + 3, it, r2, 0.4, SIZES, 4, 1.5, 0.7, 0.8)
        assert a == b and acc == accb
        r, r2 = a, b
        nacc += acc
        assert O.sample_p(2 ** 40 + 3, it, 4, 100, r, 1.2, 0.9) == T.sample_p(2 ** 40 + 3, it, 4, 100, r, 1.2, 0.9)
    assert 0 < nacc < 400


def test_product_scalar_updates_match_oracle_exactly():
    """The library's host code (csrc/chain.inc.hip) draws the same r and p as the oracle, bit for bit."""
    import ctypes as C
    import redclust_amd as rc
    L = rc.lib()
    r, p = 1.0, 0.5
    for it in range(300):
        ro, po, acc = C.c_double(), C.c_double(), C.c_uint8()
        assert L.rc_scalar_updates(77, it, r, p, SIZES, 4, 100, 1.5, 0.7, 0.8, 1.2, 0.9, C.byref(ro), C.byref(po), C.byref(acc)) == 0
        r_ref, acc_ref = O.sample_r(77, it, r, p, SIZES, 1.5, 0.7, 0.8)
        p_ref = O.sample_p(77, it, 4, 100, r_ref, 1.2, 0.9)
        assert ro.value == r_ref and po.value == p_ref and bool(acc.value) == acc_ref
        r, p = r_ref, p_ref


def test_sample_p_is_beta():
    for (K, n, r, u, v) in ((10, 100, 1.3, 1.0, 1.0), (3, 4, 0.2, 0.3, 0.1), (50, 8192, 2.0, 1.0, 1.0)):
        x = np.array([O.sample_p(5, i, K, n, r, u, v) for i in range(8000)])
        assert stats.kstest(x, stats.beta(n - K + u, r * K + v).cdf).pvalue > 1e-3


def test_sample_r_targets_the_conditional_of_r():
    """The MH chain of sample_r alone (state fixed) must have the stationary density the reference's expression
    defines: r^(η-1) e^(-σ r) [(1-p)^r / Γ(r)]^K Π Γ(n_k - 1 + r)   (mcmc.jl:118-123)."""
    eta, sigma, p, K = 2.0, 1.0, 0.3, len(SIZES)
    grid = np.linspace(1e-4, 400, 400001)
    lp = (eta - 1) * np.log(grid) + K * (grid * np.log(1 - p) - gammaln(grid)) - grid * sigma
    lp += sum(gammaln(nk - 1 + grid) for nk in SIZES)
    w = np.exp(lp - lp.max())
    cdf = np.cumsum(w); cdf /= cdf[-1]
    assert w[-1] < 1e-12 * w.max()
    r, xs = 20.0, []
    for it in range(60000):
        r, _ = O.sample_r(11, it, r, p, SIZES, eta, sigma, 4.0)
        if it >= 2000 and it % 40 == 39:
            xs.append(r)
    assert stats.kstest(np.array(xs), lambda x: np.interp(x, grid, cdf)).pvalue > 1e-3


@pytest.mark.parametrize("tag", ["d1_gibbs", "d1_mh"])
def test_golden_chain_oracle(tag):
    """The oracle's loop reproduces the transcription's free-running chain: same r / p draws (exact), same labels,
    same split–merge decisions; loglik within rounding (literal arithmetic on both sides)."""
    g = np.load(os.path.join(HERE, "golden", "golden_chain.npz"))
    d = np.load(os.path.join(HERE, "golden", "paper_datasets.npz"))
    D, truth = d["D1"], d["labels1"]
    P = T.likelihood_hyperparams(D, truth)
    orc = O.Oracle(D, P)
    numMH, iters, seed = int(g[f"{tag}_numMH"]), int(g[f"{tag}_iters"]), int(g[f"{tag}_seed"])
    rec = O.run_chain(orc, g[f"{tag}_init"], 1.0, 0.5, iters, 5, 2, 5, numMH, seed, proposalsd_r=0.7, stable=False)
    assert np.array_equal(rec["r_all"], g[f"{tag}_r_all"]) and np.array_equal(rec["p_all"], g[f"{tag}_p_all"])
    assert np.array_equal(rec["clusts"], g[f"{tag}_clusts"]) and np.array_equal(rec["K"], g[f"{tag}_K"])
    assert np.array_equal(rec["r_acc"], g[f"{tag}_r_acc"])
    if numMH:
        assert np.array_equal(rec["sm_acc"], g[f"{tag}_sm_acc"]) and np.array_equal(rec["sm_split"], g[f"{tag}_sm_split"])
        assert rec["sm_acc"].sum() >= 1
    assert np.allclose(rec["loglik"], g[f"{tag}_loglik"], rtol=1e-10) and np.allclose(rec["logposterior"], g[f"{tag}_logposterior"], rtol=1e-10)


def test_mcmcdata_refuses_zero_off_diagonal_distances():
    """The zero-distance decision (DESIGN.md "Zero distances", src/types.jl:145-157): the reference takes log(0) = -Inf
    silently; the glue refuses the input with the ArgumentError-class exception and says what to do."""
    import redclust_amd as rc
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(30, 2))
    D = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
    assert rc.MCMCData(D).n == 30
    z = D.copy(); z[5, 9] = z[9, 5] = 0.0
    with pytest.raises(ValueError, match="jitter"):
        rc.MCMCData(z)
    bad = D.copy(); bad[1, 2] += 1e-9
    with pytest.raises(ValueError, match="symmetric"):
        rc.MCMCData(bad)
    inf = D.copy(); inf[3, 4] = inf[4, 3] = np.inf
    with pytest.raises(ValueError, match="finite"):
        rc.MCMCData(inf)
    assert issubclass(rc.RedClustDomainError, ValueError) and issubclass(rc.RedClustDomainError, rc.RedClustHIPError)


def test_run_chains_argument_checks_need_no_gpu():
    """rc_run_chains / rc_comm_create validate their arguments before they touch a device."""
    import ctypes as C
    import redclust_amd as rc
    L = rc.lib()
    assert L.rc_run_chains(0, None, None, None, None, None, None, None) == -1          # RC_ERR_ARG
    assert b"n_chains" in L.rc_last_error(None)
    h = C.c_void_p()
    assert L.rc_comm_create(0, None, 0, 0, None, C.byref(h)) == -1
    assert L.rc_comm_destroy(None) == 0
