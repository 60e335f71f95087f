"""Oracle parity AT THE BENCHMARKED SIZES, on the code path bench.py times (-m gpu).

* BASELINE config 3 (N=8192, K=50, the roofline configuration): the context is created exactly as bench.py creates it —
  D only (logD derived on the fly), no forced kernel, so the automatic choice k_bulk_syml2<true, true> + k_resolve runs — and is
  compared with the CPU oracle sweep by sweep: labels / sizes / K / change counts exactly, fixed-point row sums of both
  matrices bit for bit, loglik within 1e-9 (stable restatement) and 1e-6 (literal restatement) relative.
* a size between the kernel-choice threshold (4096) and the headline size, on the automatic path across a re-layout.
* BASELINE config 5 (N=32768, K=200, 32-bit storage) without an n×n host matrix: the context is built from the points
  (rc_create_from_points) and the sweep is checked point by point by the table-driven oracle (orc_sweep_table), which is
  handed the device's exact row-sum table and the matrix rows of the points that moved.
The oracle is fed the device's own logD (rc_get_matrix / rc_get_matrix_rows), as in tests/test_gpu_derived_log.py: the
library's table log differs from libm's by ≤ 5e-16 before quantisation, which is value-checked there.
"""
import numpy as np
import pytest

import oracle_lib as O
import redclust_amd as rc
from helpers import assert_derived_log_close, rp_schedule

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def headline():
    n, K = 8192, 50
    data = rc.generatemixture(n, K, seed=1)               # bench.py's data set
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    ctx = rc.Context(D)                                   # as bench.py: D only, default (automatic) capacity, automatic kernel
    ctx.set_params(**P)
    ctx.set_state(truth)
    L = ctx.get_matrix(1)
    eD, eL = ctx.debug_rowsums(1)[2:4]
    # The oracle below is handed the DEVICE's logD (the integers every kernel uses).  That is only a parity statement if those
    # values are log(D): checked here, at this size, against the host's libm, entry by entry with the bound DESIGN.md §3 derives
    # (helpers.assert_derived_log_close); the diagonal is 0 (types.jl:155).  The literal log-likelihood is computed from the host's log(D) as well (test_headline_config_against_oracle).
    hostL = np.log(np.where(np.eye(n, dtype=bool), 1.0, D))
    # per entry: 2^-eL + 2.6e-13 + 2^-(eD+1) / D[i,j]  (~400x tighter on this data than the worst case 2^-33 of the mode)
    worst, ratio = assert_derived_log_close(L, D, eD, eL)
    assert worst <= 1e-12, worst
    orc = O.Oracle(D, P, logD=L, eL=eL, eD=eD)
    yield dict(n=n, K=K, D=D, truth=truth, P=P, ctx=ctx, orc=orc, L=L, hostL=hostL)
    ctx.close()


def _init(kind, truth, K):
    n = len(truth)
    if kind == "stationary":
        return truth.copy()
    if kind == "perturbed":                                # 2 % of the labels re-drawn
        init = truth.copy()
        idx = np.random.default_rng(11).choice(n, n // 50, replace=False)
        init[idx] = np.random.default_rng(12).integers(1, K + 1, size=len(idx))
        return init
    return np.random.default_rng(13).integers(1, K + 1, size=n).astype(np.int64)   # uniform on 1..K


@pytest.mark.parametrize("kind", ["stationary", "perturbed", "uniform"])
def test_headline_config_against_oracle(headline, kind):
    h = headline
    n, K, ctx, orc = h["n"], h["K"], h["ctx"], h["orc"]
    init = _init(kind, h["truth"], K)
    ctx.set_state(init)
    orc.set_state(init)
    nsweeps = 4
    names, moved = [], 0
    for t in range(nsweeps):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, 8192, t)
        names.append(ctx.bulk_kernel_name())
        orc.sweep_stable(r, p, 8192, t)
        lab, sizes, Kc = ctx.get_state()
        assert np.array_equal(lab, orc.clusts), (kind, t, int(np.sum(lab != orc.clusts)))
        assert np.array_equal(sizes, orc.sizes) and Kc == orc.K, (kind, t)
        st = ctx.sweep_stats()
        assert st["n_changes"] == orc.last_changes and st["K"] == orc.K, (kind, t, st, orc.last_changes)
        moved += st["n_changes"]
    # the sweep right after rc_set_state sees a cluster-contiguous layout: this is the kernel bench.py times
    assert names[0] == "k_bulk_syml2<true, true>", names
    if kind == "stationary":
        assert all(x == "k_bulk_syml2<true, true>" for x in names), names
    else:
        assert moved > (50 if kind == "perturbed" else n // 2)
    # the row-sum table after the corrections of four sweeps, both matrices, bit for bit
    for k in np.unique(orc.clusts)[[0, 7, -1]]:
        sd, sl, eD, eL = ctx.debug_rowsums(int(k))
        m = orc.clusts == k
        assert (eD, eL) == (orc.eD, orc.eL)
        assert np.array_equal(sd, orc.Dq[:, m].sum(axis=1)) and np.array_equal(sl, orc.Lq[:, m].sum(axis=1)), (kind, k)
    ll = ctx.loglik()
    ref = orc.loglik_stable()
    assert abs(ll - ref) <= 1e-9 * abs(ref), (ll, ref)               # tolerance: 1e-9 relative (north_star allows 1e-6)
    lit = orc.loglik_literal()
    assert abs(ll - lit) <= 1e-6 * abs(lit), (ll, lit)               # vs the reference's formulas as written
    # ... and as written on the HOST's log(D) (libm), not the device's: the north_star's 1e-6 against the CPU reference path
    lit_host = O.lib().orc_loglik_literal(n, h["D"].reshape(-1), h["hostL"].reshape(-1), orc.clusts, orc.sizes, orc.P)
    assert abs(ll - lit_host) <= 1e-6 * abs(lit_host), (ll, lit_host)
    lp = ctx.logprior(*rp_schedule(nsweeps - 1))
    assert abs(lp - orc.logprior(*rp_schedule(nsweeps - 1))) <= 1e-12 * abs(lp)
    # the same sweeps enqueued without host synchronisation (what bench.py does) end in the same state
    final = orc.clusts.copy()
    ctx.set_state(init)
    for t in range(nsweeps):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, 8192, t, blocking=False)
    ctx.synchronize()
    lab, sizes, Kc = ctx.get_state()
    assert np.array_equal(lab, final) and Kc == orc.K and ctx.loglik() == ll


def test_headline_rowsum_checksums(headline):
    """Σ_k S[k][i] = Σ_j X[i,j] for both matrices (exact in fixed point) under the generating labels, from the kernel
    bench.py times; co-clustering counts of two recorded samples."""
    h = headline
    n, ctx, orc, truth = h["n"], h["ctx"], h["orc"], h["truth"]
    ctx.set_state(truth)
    tot_d = np.zeros(n, np.int64); tot_l = np.zeros(n, np.int64)
    for lab in np.unique(truth):
        sd, sl, eD, eL = ctx.debug_rowsums(int(lab))
        tot_d += sd; tot_l += sl
    assert np.array_equal(tot_d, orc.Dq.sum(axis=1)) and np.array_equal(tot_l, orc.Lq.sum(axis=1))
    assert ctx.bulk_kernel_name() == "k_bulk_syml2<true, true>"
    ctx.cocluster_reset()
    ctx.record_sample(False)
    ctx.gibbs_sweep(1.0, 0.5, 5, 0)
    c = ctx.record_sample(True)
    cnt = ctx.cocluster_counts()
    lab = ctx.get_state()[0]
    assert np.all(np.diag(cnt) == 2) and np.array_equal(cnt, cnt.T)
    exp = (truth[:, None] == truth[None, :]).astype(np.uint32) + (lab[:, None] == lab[None, :])
    assert np.array_equal(cnt, exp)
    canon = np.zeros(n, np.int64)
    O.lib().orc_sortlabels(n, lab, canon)
    assert np.array_equal(c, canon)


def test_auto_path_between_threshold_and_headline_size_crosses_relayout():
    """n = 5000 (4096 < n < 8192), shuffled points, overlapping clusters: the automatic path starts on k_bulk_syml2<true, true>,
    label movement fragments the layout (the full-read kernel takes over), the library re-lays the points out after 32
    sweeps and returns to the symmetric kernel — every sweep compared with the oracle."""
    n, K = 5000, 5
    data = rc.generatemixture(n, K, seed=17, sigma=0.55, dim=6)
    sh = np.random.default_rng(4).permutation(n)
    D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)]); truth = data["clusts"][sh]
    P = dict(rc.likelihood_hyperparams(D, truth), maxK=12)
    ctx = rc.Context(D, kcap=64)
    ctx.set_params(**P)
    ctx.set_state(truth)
    eD, eL = ctx.debug_rowsums(int(truth[0]))[2:4]
    orc = O.Oracle(D, P, logD=ctx.get_matrix(1), eL=eL, eD=eD)
    orc.set_state(truth)
    l0 = ctx.layout_info()[0]
    names, relaid_at = [], None
    for t in range(40):
        ctx.gibbs_sweep(1.0, 0.5, 3, t, blocking=(t % 2 == 0))
        names.append(ctx.bulk_kernel_name())
        orc.sweep_stable(1.0, 0.5, 3, t)
        lab, sizes, Kc = ctx.get_state()
        assert np.array_equal(lab, orc.clusts) and np.array_equal(sizes, orc.sizes) and Kc == orc.K, t
        if relaid_at is None and ctx.layout_info()[0] > l0:
            relaid_at = t
    assert names[0] == "k_bulk_syml2<true, true>" and "k_bulk<long long, true>" in names, names
    assert relaid_at is not None and names[relaid_at] == "k_bulk_syml2<true, true>", (relaid_at, names)
    for k in np.unique(orc.clusts)[:3]:
        sd, sl = ctx.debug_rowsums(int(k))[:2]
        m = orc.clusts == k
        assert np.array_equal(sd, orc.Dq[:, m].sum(axis=1)) and np.array_equal(sl, orc.Lq[:, m].sum(axis=1))
    assert abs(ctx.loglik() - orc.loglik_stable()) <= 1e-9 * abs(orc.loglik_stable())
    ctx.close()


def _device_table(ctx, labels):
    rows = np.unique(labels)
    TD = np.empty((len(rows), ctx.n), np.int64); TL = np.empty((len(rows), ctx.n), np.int64)
    eD = eL = None
    for t, lab in enumerate(rows):
        TD[t], TL[t], eD, eL = ctx.debug_rowsums(int(lab))
    return rows, TD, TL, eD, eL


def test_config5_against_table_oracle():
    """BASELINE config 5: N=32768, K=200, 32-bit storage, built from the points (no n×n host matrix).
    (i) the device's D rows agree with the host's pairwise distances; (ii) the row-sum table of k_bulk_sym32 equals, for
    1024 sampled points and every cluster, the bucketed sums of the device's own matrix rows (both matrices, exact), and for ALL
    points its column totals equal the row totals of an independent kernel and its K x K block sums are symmetric;
    (iii) one stationary and one perturbed sweep are re-enacted point by point by the table-driven oracle: labels, sizes,
    K and the change count must match exactly; (iv) async == blocking."""
    n, K = 32768, 200
    data = rc.generatemixture(n, K, seed=1, points_only=True)
    pts, truth = data["points"], data["clusts"]
    ctx = rc.Context.from_points(pts, kcap=512, storage_bits=32)
    P = rc.likelihood_hyperparams_device(ctx, truth)
    ctx.set_params(**P)
    A = O.size_table(P, n)
    # (i) + (ii) on 1024 sampled points, 64 rows of both matrices at a time
    ctx.set_state(truth)
    rows, TD, TL, eD, eL = _device_table(ctx, truth)
    assert ctx.bulk_kernel_name() == "k_bulk_sym32"
    sample_all = np.sort(np.random.default_rng(0).choice(n, 1024, replace=False))
    sq = np.einsum("ij,ij->i", pts, pts)
    onehot = (truth[:, None] == rows[None, :]).astype(np.int64)
    for b in range(0, len(sample_all), 64):
        sample = sample_all[b:b + 64]
        RD, RL = ctx.get_matrix_rows(0, sample), ctx.get_matrix_rows(1, sample)
        host = np.sqrt(np.maximum(sq[sample][:, None] + sq[None, :] - 2.0 * pts[sample] @ pts.T, 0.0))
        host[np.arange(len(sample)), sample] = 0.0
        assert np.max(np.abs(RD - host)) <= 2.0 ** -29 * host.max()          # 32-bit grid: 2^-30 of the largest entry (+ f64 noise)
        qD = np.rint(np.ldexp(RD, eD)).astype(np.int64); qL = np.rint(np.ldexp(RL, eL)).astype(np.int64)
        assert np.array_equal(np.ldexp(qD.astype(np.float64), -eD), RD)      # value = q·2^-e exactly
        assert np.array_equal(qD @ onehot, TD[:, sample].T) and np.array_equal(qL @ onehot, TL[:, sample].T), b
        assert np.all(qD[np.arange(len(sample)), sample] == 0)              # pairwise distances: zero diagonal
    del onehot
    # (ii') ALL 32768 rows of the table, through identities that need no matrix on the host:
    #   Σ_k S[k][i] = Σ_j X[i,j] for every i, the right-hand side from rc_debug_rowtotals (a plain per-row kernel on the caller-order
    #   copy: not the reduction's layout, tiles or atomics);
    #   B[k][l] = Σ_{i in l} S[k][i] = Σ_{i in l} Σ_{j in k} X[i,j] is symmetric in (k, l) because X is (src/types.jl:149-151) —
    #   a wrong or misplaced entry in any row breaks the pair (k, l) it belongs to unless an equal error sits in the mirrored row.
    totD, totL = ctx.debug_rowtotals()

    def table_identities(labels, rows, TD, TL):
        assert np.array_equal(TD.sum(axis=0), totD) and np.array_equal(TL.sum(axis=0), totL)
        order = np.argsort(labels, kind="stable")
        starts = np.flatnonzero(np.r_[True, labels[order][1:] != labels[order][:-1]])
        assert len(starts) == len(rows)
        for Tm in (TD, TL):
            B = np.add.reduceat(Tm[:, order], starts, axis=1)               # [k][l], exact int64
            assert np.array_equal(B, B.T)

    table_identities(truth, rows, TD, TL)
    diag = np.zeros(n, np.int64)
    idx = np.random.default_rng(0).choice(n, 500, replace=False)
    perturbed = truth.copy()
    perturbed[idx] = np.random.default_rng(1).integers(1, K + 1, size=500)
    total_moved = 0
    for name, init in (("stationary", truth), ("perturbed", perturbed)):
        ctx.set_state(init)
        rows, TD, TL, eD, eL = _device_table(ctx, init)
        table_identities(init, rows, TD, TL)
        ctx.gibbs_sweep(1.0, 0.5, 42, 0)
        lab, sizes, Kc = ctx.get_state()
        st = ctx.sweep_stats()
        xs = np.flatnonzero(lab != init)
        XD = np.rint(np.ldexp(ctx.get_matrix_rows(0, xs), eD)).astype(np.int64)
        XL = np.rint(np.ldexp(ctx.get_matrix_rows(1, xs), eL)).astype(np.int64)
        got = O.sweep_table(P, A, rows, TD, TL, diag, eD, eL, init, 1.0, 0.5, 42, 0, xs, XD, XL)
        assert np.array_equal(got[0], lab), (name, int(np.sum(got[0] != lab)))
        assert np.array_equal(got[1], sizes) and got[2] == Kc and got[3] == st["n_changes"] == len(xs), (name, got[2:], st)
        total_moved += len(xs)
        # two more sweeps, then the same three enqueued back to back
        for t in (1, 2):
            ctx.gibbs_sweep(1.0, 0.5, 42, t)
        c1, s1, K1 = ctx.get_state()
        assert s1.sum() == n and K1 == np.sum(s1 > 0) and np.array_equal(np.bincount(c1, minlength=n + 1)[1:], s1)
        ll1 = ctx.loglik()
        ctx.set_state(init)
        for t in range(3):
            ctx.gibbs_sweep(1.0, 0.5, 42, t, blocking=False)
        ctx.synchronize()
        c2, s2, K2 = ctx.get_state()
        assert np.array_equal(c1, c2) and K1 == K2 and ctx.loglik() == ll1, name
    assert total_moved > 100
    ctx.close()


def test_maximum_slot_capacity_and_many_clusters():
    """kcap at its maximum (4096: the resolver's slot tables take ≈150 KiB of LDS) and a state with more than a thousand
    clusters (every fourth point a singleton): exact against the oracle from a moving start, resolver alone on the CU."""
    n, K = 4200, 30
    data = rc.generatemixture(n, K, seed=9, sigma=0.15)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    ctx = rc.Context(D, kcap=4096)
    ctx.set_params(**P)
    init = truth.copy()
    init[::4] = K + 1 + np.arange(len(init[::4]))             # 1050 singletons with labels 31..1080
    ctx.set_state(init)
    L = ctx.get_matrix(1)
    eD, eL = ctx.debug_rowsums(1)[2:4]
    orc = O.Oracle(D, P, logD=L, eL=eL, eD=eD)
    orc.set_state(init)
    assert ctx.get_state()[2] == K + 1050
    for t in range(3):
        r, p = rp_schedule(t)
        ctx.gibbs_sweep(r, p, 77, t)
        orc.sweep_stable(r, p, 77, t)
        lab, sizes, Kc = ctx.get_state()
        st = ctx.sweep_stats()
        assert np.array_equal(lab, orc.clusts) and np.array_equal(sizes, orc.sizes) and Kc == orc.K, (t, st)
        assert st["n_changes"] == orc.last_changes
    assert abs(ctx.loglik() - orc.loglik_stable()) <= 1e-9 * abs(orc.loglik_stable())
    ctx.close()


def test_moving_regime_at_headline_size_with_and_without_score_cache():
    """N = 8192 on overlapping clusters (sigma = 0.2: the moving regime of bench.py — dozens to hundreds of label changes per
    sweep with births, deaths and relabelings, several resolver rounds): three sweeps from the generating labels against the
    oracle, once with the resolver's score cache off and once with it filled in every sweep (RC_SCORE_CACHE, read when the
    context is created; the default fills it only after a sweep that changed labels) — labels, sizes, K and change counts exact."""
    import os
    n, K = 8192, 50
    data = rc.generatemixture(n, K, seed=2, sigma=0.2)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    saved = os.environ.get("RC_SCORE_CACHE")
    ref = None
    try:
        for mode in ("0", "1"):
            os.environ["RC_SCORE_CACHE"] = mode
            ctx = rc.Context(D, kcap=512)
            ctx.set_params(**P)
            ctx.set_state(truth)
            if ref is None:                                   # the oracle's three sweeps, once
                L = ctx.get_matrix(1)
                eD, eL = ctx.debug_rowsums(1)[2:4]
                orc = O.Oracle(D, P, logD=L, eL=eL, eD=eD)
                orc.set_state(truth)
                ref = []
                for t in range(3):
                    r, p = rp_schedule(t)
                    orc.sweep_stable(r, p, 77, t)
                    ref.append((orc.clusts.copy(), orc.sizes.copy(), orc.K, orc.last_changes))
                del orc, L
                assert sum(x[3] for x in ref) > 100 and ref[-1][2] > K     # it moves, and clusters are born
            rounds = 0
            for t in range(3):
                r, p = rp_schedule(t)
                ctx.gibbs_sweep(r, p, 77, t, blocking=bool(t & 1))
                lab, sizes, Kc = ctx.get_state()
                st = ctx.sweep_stats()
                assert np.array_equal(lab, ref[t][0]), (mode, t, int(np.sum(lab != ref[t][0])))
                assert np.array_equal(sizes, ref[t][1]) and Kc == ref[t][2] and st["n_changes"] == ref[t][3], (mode, t, st)
                rounds += st["n_rounds"]
            assert rounds > 3                                  # more than one resolver round somewhere: the cached passes ran
            ctx.close()
    finally:
        if saved is None: os.environ.pop("RC_SCORE_CACHE", None)
        else: os.environ["RC_SCORE_CACHE"] = saved


def test_maximum_slot_capacity_at_headline_size(headline):
    """kcap = 4096 at n = 8192: the resolver's tables fill the CU's 160 KiB only with a smaller batch capacity, which the library
    picks by itself (512 entries per batch would need 164 KiB); two sweeps from a 2 %-perturbed start agree with the default context."""
    h = headline
    init = _init("perturbed", h["truth"], h["K"])
    big = rc.Context(h["D"], kcap=4096)
    big.set_params(**h["P"])
    out = []
    for ctx in (h["ctx"], big):
        ctx.set_state(init)
        for t in range(2):
            r, p = rp_schedule(t)
            ctx.gibbs_sweep(r, p, 4096, t)
        out.append((ctx.get_state(), ctx.sweep_stats()["n_changes"], ctx.loglik()))
    big.close()
    (a, ca, la), (b, cb, lb) = out
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and ca == cb and la == lb


def _literal_oracle(D, P):
    """The reference's arithmetic as written (oracle literal mode) on the host's own log(D): no fixed-point copies."""
    orc = O.Oracle.__new__(O.Oracle)
    orc.L = O.lib()
    orc.n = D.shape[0]
    orc.D = np.ascontiguousarray(D)
    orc.logD = np.ascontiguousarray(np.log(np.where(np.eye(orc.n, dtype=bool), 1.0, D)))
    orc.P = O.params(P)
    return orc


def _report(name, rec):
    """Divergence reports go to the test output and, when the run's scratch directory exists, to gpurun_out/ (copied to
    profiles/ by hand)."""
    import json
    import os
    print(name, json.dumps(rec))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"parity_{name}.json"), "w") as f:
            json.dump(rec, f, indent=1)


@pytest.mark.parametrize("n,K,sigma,nsweeps", [(2000, 20, 0.1, 12), (2000, 20, 0.25, 12), (8192, 50, 0.1, 8), (8192, 50, 0.2, 6)])
def test_hip_sweep_against_the_reference_arithmetic_as_written(n, K, sigma, nsweeps, headline):
    """the HIP sweep (regrouped arithmetic, fixed-point sums, table log) against the reference's formulas AS
    WRITTEN (lgamma / log of the full sums in double, src/mcmc.jl:221-247) on the host's libm log(D), from the 2 %-perturbed
    start, at BASELINE configs 2 and 3 — on the separated clusters of the benchmark (sigma = 0.1: the chain is back at the
    generating labels after one sweep) and on overlapping ones (sigma = 0.25 / 0.2: tens of label changes, births and deaths
    in every sweep; the 8192 case is bench.py's moving_regime data).  Two comparisons per sweep: free-running (both chains on
    their own; the first sweep at which the label vectors differ is reported) and teacher-forced (the literal sweep restarted
    from the HIP chain's state: the number of labels that differ after ONE sweep from identical states).  The literal formulas
    cancel terms of magnitude 1e8-1e10, so a draw whose two best Gumbel-perturbed scores are closer than
    their rounding noise may legitimately differ; asserted: at most 2 labels per sweep under teacher forcing, and identical
    free-running labels in the first two sweeps."""
    if n == 8192 and sigma == 0.1:
        D, truth, P = headline["D"], headline["truth"], headline["P"]
        ctx = headline["ctx"]
    else:
        data = rc.generatemixture(n, K, seed=3 if n == 2000 else 2, sigma=sigma)
        D, truth = data["distancematrix"], data["clusts"]
        P = rc.likelihood_hyperparams(D, truth)
        ctx = rc.Context(D)
        ctx.set_params(**P)
    init = truth.copy()
    idx = np.random.default_rng(11).choice(n, n // 50, replace=False)
    init[idx] = np.random.default_rng(12).integers(1, K + 1, size=len(idx))
    ctx.set_state(init)
    free = _literal_oracle(D, P); free.set_state(init)
    forced = _literal_oracle(D, P)
    first_div, forced_diff, changes, Ks = None, [], [], []
    for t in range(nsweeps):
        r, p = rp_schedule(t)
        before = ctx.get_state()[0]
        ctx.gibbs_sweep(r, p, 4242, t)
        lab = ctx.get_state()[0]
        st = ctx.sweep_stats()
        changes.append(int(st["n_changes"])); Ks.append(int(st["K"]))
        free.sweep_literal(r, p, 4242, t)
        if first_div is None and not np.array_equal(lab, free.clusts):
            first_div = t
        forced.set_state(before)
        forced.sweep_literal(r, p, 4242, t)
        forced_diff.append(int(np.sum(forced.clusts != lab)))
    _report(f"literal_n{n}_sigma{sigma}", dict(n=n, K=K, sigma=sigma, sweeps=nsweeps, start="2% of the generating labels re-drawn",
                                               label_changes_per_sweep=changes, K_per_sweep=Ks, first_divergence_sweep_free_running=first_div,
                                               labels_differing_teacher_forced=forced_diff))
    assert changes[0] > n // 100
    if sigma > 0.1:
        assert min(changes) > 5, changes                 # labels keep moving
    assert first_div is None or first_div >= 2, first_div
    assert max(forced_diff) <= 2, forced_diff
    if not (n == 8192 and sigma == 0.1):
        ctx.close()


@pytest.mark.parametrize("numMH", [0, 1])
def test_chain_at_baseline_config_2(numMH):
    """BASELINE configs[1] — N = 2000, K = 20, one chain — as a CHAIN: runsampler's loop (src/mcmc.jl:536-556: r, p,
    split-merge, sweep, record) through rc_run_chain against the oracle's loop, 300 iterations, free-running scalar updates,
    numMH = 0 and the reference's default numMH = 1.  Labels / K / r / p / acceptances exactly; log-posterior trace to 1e-9
    against the oracle's regrouped arithmetic and to 1e-6 (the north_star bar) against the reference's formulas as written on
    libm's log(D).  Clusters overlap (sigma = 0.25: ~25 label changes per sweep, K around 80 with births and deaths) and 5 %
    of the labels start re-drawn, so labels move throughout."""
    n, K, iters, burnin, thin = 2000, 20, 300, 50, 5
    data = rc.generatemixture(n, K, seed=21, sigma=0.25)
    D, truth = data["distancematrix"], data["clusts"]
    P = rc.likelihood_hyperparams(D, truth)
    init = truth.copy()
    idx = np.random.default_rng(5).choice(n, n // 20, replace=False)
    init[idx] = np.random.default_rng(6).integers(1, K + 1, size=len(idx))
    ctx = rc.Context(D)                                   # derived logD, default capacity: what runsampler creates
    ctx.set_params(**P)
    ctx.set_state(init)
    ctx.cocluster_reset()
    L = ctx.get_matrix(1)
    eD, eL = ctx.debug_rowsums(int(init[0]))[2:4]
    hostL = np.log(np.where(np.eye(n, dtype=bool), 1.0, D))
    assert_derived_log_close(L, D, eD, eL)
    orc = O.Oracle(D, P, logD=L, eL=eL, eD=eD)
    if numMH:
        ctx.attach_host_matrices(D, L)
    ch = ctx.run_chain(iters, burnin, thin, 5, numMH, 77, 1.0, 0.5, 1.0)
    ref = O.run_chain(orc, init, 1.0, 0.5, iters, burnin, thin, 5, numMH, 77, stable=True)
    assert ch["num_samples"] == len(ref["K"]) == (iters - burnin) // thin
    for k, kr in (("clusts", "clusts"), ("K", "K"), ("r", "r"), ("p", "p"), ("r_all", "r_all"), ("p_all", "p_all"), ("r_acceptances", "r_acc")):
        assert np.array_equal(ch[k], ref[kr]), (k, numMH)
    if numMH:
        assert np.array_equal(ch["splitmerge_acceptances"], ref["sm_acc"]) and np.array_equal(ch["splitmerge_splits"], ref["sm_split"])
    assert np.allclose(ch["logposterior"], ref["logposterior"], rtol=1e-9, atol=0)
    moved = int(np.sum(ch["clusts"][0] != ch["clusts"][-1]))
    lit = _literal_oracle(D, P)
    worst = 0.0
    for j in range(ch["num_samples"]):
        lit.set_state(ch["clusts"][j])
        lp = lit.loglik_literal() + lit.logprior(ch["r"][j], ch["p"][j])
        worst = max(worst, abs(ch["logposterior"][j] - lp) / abs(lp))
    _report(f"chain_config2_numMH{numMH}", dict(n=n, K=K, iterations=iters, samples=int(ch["num_samples"]), numMH=numMH,
                                               labels_moved_first_to_last_sample=moved,
                                               splitmerge_acceptances=int(np.sum(ch["splitmerge_acceptances"])) if numMH else 0,
                                               max_rel_logposterior_error_vs_literal_on_libm_logD=worst))
    assert worst <= 1e-6, worst
    assert moved > 20, moved
    lab, sizes, Kc = ctx.get_state()
    assert np.array_equal(lab, orc.clusts) and Kc == orc.K
    ctx.close()
