"""Derived-logD mode (no logD given: the library evaluates Lq = rint(log(Dq·2^-eD)·2^eL) on the fly with its own
table-based log instead of storing the matrix — DESIGN.md §2).  The oracle is handed the device's logD
(rc_get_matrix), after which everything must match bit for bit exactly as in the stored mode."""
import os

import numpy as np
import pytest

import oracle_lib as O
import redclust_amd as rc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def derived_pair(D, P, init, kcap=0):
    ctx = rc.Context(D, kcap=kcap)                  # no logD: derived mode
    L = ctx.get_matrix(1)
    ctx.set_params(**P)
    ctx.set_state(init)
    eD, eL = ctx.debug_rowsums(int(init[0]))[2:4]   # the library's exponents (derived mode: eL capped at 50 - exponent(max|logD|), eD at 51 - exponent(max D))
    orc = O.Oracle(D, P, logD=L, eL=eL, eD=eD)
    ctx.set_params(**P)
    ctx.set_state(init)
    orc.set_state(init)
    return orc, ctx, L


def is_derived(ctx):
    ctx.gibbs_sweep(1.0, 0.5, 0, 10 ** 9)            # any sweep, so that the kernel info is populated
    name, nbytes = ctx.bulk_kernel_info()
    n = ctx.n
    return nbytes in (n * (n + 1) / 2 * 8, n * (n + 1) / 2 * 6, n * n * 8.0)   # (6: the 48-bit packed copy k_bulk_syml2 streams)


def test_derived_log_values_and_rowsums():
    d = np.load(os.path.join(HERE, "golden", "paper_datasets.npz"))
    D, truth = d["D1"], d["labels1"]
    P = rc.likelihood_hyperparams(D, truth)
    init = np.random.default_rng(0).integers(1, 11, 100).astype(np.int64)
    orc, ctx, L = derived_pair(D, P, init)
    ref = np.log(np.where(np.eye(100, dtype=bool), 1.0, D))
    quantum = np.ldexp(1.0, -orc.eL)
    assert np.all(np.diag(L) == 0) and np.array_equal(L, L.T)
    # rc_qlog vs libm's log of the host's D: the degree-4 polynomial of log1p (next term r^5/5 < 1.8e-13 for |r| < 1/257), the
    # rounding of the fixed-point entry itself (D is stored to 2^-eD, 47 significant bits of the largest entry: relative 2^-(eD+1)/D)
    # and the rounding to the quantum of logD
    bound = quantum + 2e-13 + np.ldexp(1.0, -orc.eD - 1) / np.where(np.eye(100, dtype=bool), 1.0, D)
    assert np.all(np.abs(L - ref) <= bound), float(np.max(np.abs(L - ref)))
    assert np.max(np.abs(L - ref)) <= 1e-12
    onehot = (init[:, None] == np.arange(1, 101)[None, :]).astype(np.int64)
    for lab in np.unique(init):
        sd, sl, eD, eL = ctx.debug_rowsums(int(lab))
        assert (eD, eL) == (orc.eD, orc.eL)
        assert np.array_equal(sd, (orc.Dq @ onehot)[:, lab - 1]) and np.array_equal(sl, (orc.Lq @ onehot)[:, lab - 1])
    assert is_derived(ctx)
    ctx.close()


@pytest.mark.parametrize("kernel", ["sym", "perm"])
@pytest.mark.parametrize("n,K,sigma", [(333, 5, 0.5), (1029, 9, 0.25), (2050, 12, 0.15)])
def test_derived_sweeps_match_oracle(kernel, n, K, sigma):
    data = rc.generatemixture(n, K, seed=n, sigma=sigma, dim=max(K, 6))
    sh = np.random.default_rng(n).permutation(n)
    D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)]); truth = data["clusts"][sh]
    P = rc.likelihood_hyperparams(D, truth)
    init = truth.copy()
    init[::5] = np.random.default_rng(1).integers(1, K + 1, len(init[::5]))
    orc, ctx, _ = derived_pair(D, P, init, kcap=128)
    ctx.set_bulk_kernel(kernel)
    moved = 0
    for t in range(8):
        ctx.gibbs_sweep(1.0 + 0.1 * t, 0.5, 42, t)
        orc.sweep_stable(1.0 + 0.1 * t, 0.5, 42, t)
        lab, sizes, Kc = ctx.get_state()
        assert np.array_equal(lab, orc.clusts) and np.array_equal(sizes, orc.sizes) and Kc == orc.K, t
        moved += ctx.sweep_stats()["n_changes"]
    assert moved > 0
    assert abs(ctx.loglik() - orc.loglik_stable()) <= 1e-9 * abs(orc.loglik_stable())
    for k in np.unique(orc.clusts)[:4]:                        # the S table after corrections, both directions
        sd, sl = ctx.debug_rowsums(int(k))[:2]
        m = orc.clusts == k
        assert np.array_equal(sd, orc.Dq[:, m].sum(axis=1)) and np.array_equal(sl, orc.Lq[:, m].sum(axis=1))
    ctx.close()


def test_derived_equals_stored_given_the_same_logD_and_modes_agree():
    """A stored-mode context fed the derived logD walks the same chain; incremental mode and split–merge
    apply / revert (k_apply_moves) stay exact in the derived mode."""
    data = rc.generatemixture(600, 5, seed=3, sigma=0.45, dim=6)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    orc, a, L = derived_pair(D, P, truth, kcap=64)
    b = rc.Context(D, logD=L, kcap=64); b.set_params(**P); b.set_state(truth)
    c = rc.Context(D, kcap=64); c.set_params(**P); c.set_state(truth); c.set_mode("incremental")
    a.attach_host_matrices(D, L); b.attach_host_matrices(D, L)
    for t in range(25):
        for x in (a, b, c):
            x.gibbs_sweep(1.0, 0.5, 7, t)
        orc.sweep_stable(1.0, 0.5, 7, t)
        for name, x in (("derived", a), ("stored", b), ("incremental", c)):
            lab = x.get_state()[0]
            assert np.array_equal(lab, orc.clusts), (name, t, np.flatnonzero(lab != orc.clusts)[:8], x.sweep_stats(), orc.last_changes,
                                                     x.bulk_kernel_name())
        if t % 5 == 4:
            inf = orc.mh_proposal(1.0, 0.5, 5, 7, t, 0, mode=1)
            ra = a.splitmerge(1.0, 0.5, 5, 7, t, 0); rb = b.splitmerge(1.0, 0.5, 5, 7, t, 0)
            assert ra == rb == (bool(inf.accept), bool(inf.split)), (t, ra, rb)
            if ra[0]:
                c.set_state(a.get_state()[0])
            for name, x in (("derived", a), ("stored", b)):
                assert np.array_equal(x.get_state()[0], orc.clusts), (name, "after the proposal of iteration", t)
        sa, sb = a.get_state(), b.get_state()
        assert np.array_equal(sa[0], sb[0]) and sa[2] == sb[2]
    # same partition, block sums exact in each context — but the derived-mode context stores D with fewer fraction bits
    # (create_impl caps eD at 47 - ex for the 48-bit packed copy; the stored-logD context keeps 62 - ex - ceil(log2 n)), so
    # the integer block sums differ in their last quanta: equal to the last bits, not bit for bit.  (The order of the
    # long-double sum is the same in all three: loglik_host sums in ascending label order.)
    la, lb, lc = a.loglik(), b.loglik(), c.loglik()
    assert abs(la - lb) <= 1e-12 * abs(la) and abs(la - lc) <= 1e-12 * abs(la), (la, lb, lc)
    for x in (a, b, c):
        x.close()


def test_fallback_to_stored_log_when_an_entry_quantises_to_zero():
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(40, 3))
    D = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    D[3, 7] = D[7, 3] = 1e-40                        # positive, but far below the fixed-point quantum
    ctx = rc.Context(D)
    L = ctx.get_matrix(1)
    assert np.isclose(L[3, 7], np.log(1e-40), rtol=1e-12) and L[3, 3] == 0
    ctx.set_params(**rc.likelihood_hyperparams(D, np.arange(40) % 4 + 1))
    ctx.set_state(np.arange(40) % 4 + 1)
    ctx.gibbs_sweep(1.0, 0.5, 1, 0)
    assert ctx.bulk_kernel_info()[1] in (2 * 40 * 41 / 2 * 8, 2 * 40 * 40 * 8.0)     # two matrices are read: stored mode
    ctx.close()


def test_two_contexts_stay_identical_over_a_long_moving_chain():
    """Regression for the grid-barrier publication race (DESIGN.md "Grid barrier and global stores"): two contexts with
    different kernels walk the same moving chain asynchronously for 1500 sweeps; every third sweep both states are
    pulled and must agree, and each size table must match its own labels."""
    n, K = 2048, 4
    data = rc.generatemixture(n, K, seed=5, sigma=0.6, dim=6)
    sh = np.random.default_rng(8).permutation(n)
    D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)]); truth = data["clusts"][sh]
    P = dict(rc.likelihood_hyperparams(D, truth), maxK=12)
    A = rc.Context(D, kcap=64); A.set_params(**P); A.set_state(truth)
    B = rc.Context(D, kcap=64); B.set_params(**P); B.set_bulk_kernel("perm"); B.set_state(truth)
    for t in range(1500):
        A.gibbs_sweep(1.0, 0.5, 3, t, blocking=False); B.gibbs_sweep(1.0, 0.5, 3, t, blocking=False)
        if t % 3 == 2:
            a, b = A.get_state(), B.get_state()
            assert np.array_equal(a[0], b[0]) and a[2] == b[2], t
            assert np.array_equal(a[1], np.bincount(a[0], minlength=n + 1)[1:]), t
            assert np.array_equal(b[1], np.bincount(b[0], minlength=n + 1)[1:]), t
    A.close(); B.close()


@pytest.mark.parametrize("kernel", ["perm", "sym"])
def test_many_small_clusters_against_oracle(kernel):
    """700 initial clusters of ~2 points (kcap = 1024): most 128-column segments span many clusters, so the symmetric
    kernel runs on its element-wise slow path, and the sweeps are dominated by deaths (structural commits)."""
    n = 1500
    data = rc.generatemixture(n, 12, seed=3, sigma=0.4, dim=12)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    init = np.random.default_rng(0).integers(1, 701, n).astype(np.int64)
    ctx = rc.Context(D, kcap=1024)
    ctx.set_params(**P); ctx.set_bulk_kernel(kernel); ctx.set_state(init)
    L = ctx.get_matrix(1)
    eD, eL = ctx.debug_rowsums(int(init[0]))[2:4]
    orc = O.Oracle(D, P, logD=L, eL=eL, eD=eD)
    orc.set_state(init)
    for t in range(8):
        ctx.gibbs_sweep(1.0, 0.5, 4, t)
        orc.sweep_stable(1.0, 0.5, 4, t)
        lab, sizes, K = ctx.get_state()
        assert np.array_equal(lab, orc.clusts) and np.array_equal(sizes, orc.sizes) and K == orc.K, t
    assert K < 400 and abs(ctx.loglik() - orc.loglik_stable()) <= 1e-9 * abs(orc.loglik_stable())
    ctx.close()
