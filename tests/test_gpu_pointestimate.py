"""GPU parity of the point-estimation kernels (csrc/pointestimate.inc.hip, through the C ABI) against the oracle and
the golden vectors — pairwise loss matrices of getpointestimate(method="MPEL") (pointestimate.jl:49-58), the pair
measures of evaluateclustering / binderloss / infodist, and the reference's own tests (test_pointestimates.jl)."""
import io
import os

import numpy as np
import pytest

import oracle_lib as O
import redclust_amd as rc
from redclust_amd import _lib

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KINDS = {"binder": 0, "omARI": 1, "VI": 2, "ID": 3}
RTOL = 1e-10  # floating-point sums in a different order than the oracle's; integer part (Σ n_ij²) is exact


def chain_like_samples(rng, m, n, K, noise):
    base = np.sort(rng.integers(1, K + 1, n))
    out = []
    for s in range(m):
        x = base.copy()
        flip = rng.random(n) < noise * (1 + s % 3)
        x[flip] = rng.integers(1, K + 3, int(flip.sum()))
        out.append(x)
    return np.stack(out).astype(np.int64)


@pytest.mark.parametrize("loss", list(KINDS))
def test_golden_loss_matrices(loss):
    g = np.load(os.path.join(HERE, "golden", "golden_pointestimate.npz"))
    S = g["samples"]
    M, cs, i, _ = _lib.loss_matrix(S, KINDS[loss])
    assert np.array_equal(M, M.T) and np.all(np.diag(M) == 0)
    assert np.allclose(M, g[f"lossmatrix_{loss}"], rtol=RTOL, atol=1e-13)
    assert np.allclose(cs, g[f"colsum_{loss}"], rtol=RTOL)
    gi = int(g[f"argmin_{loss}"])
    assert np.isclose(cs[i], g[f"colsum_{loss}"][gi], rtol=1e-12) and i == int(np.argmin(cs))


@pytest.mark.parametrize("m,n,K,noise", [(2, 64, 3, 0.2), (17, 257, 5, 0.1), (40, 1000, 30, 0.02), (33, 300, 12, 0.6),
                                          (5, 4099, 60, 0.3)])
def test_loss_matrix_vs_oracle(m, n, K, noise):
    S = chain_like_samples(np.random.default_rng(m * n), m, n, K, noise)
    S[:, ::7] = S[:, ::7][:, ::-1]      # unsorted point order as well
    for loss, kind in KINDS.items():
        M, cs, i, _ = _lib.loss_matrix(S, kind)
        io_, L, cso = O.mpel(S, kind)
        assert np.allclose(M, L, rtol=RTOL, atol=1e-13), loss
        assert np.allclose(cs, cso, rtol=RTOL) and np.isclose(cs[i], cso[io_], rtol=1e-12)


def test_binder_is_exact_integer_work():
    """Σ n_ij² is accumulated in integers: the Binder/Mirkin entries equal the oracle's bit for bit."""
    S = chain_like_samples(np.random.default_rng(1), 25, 513, 9, 0.15)
    M = _lib.loss_matrix(S, 0)[0]
    L = O.mpel(S, 0)[1]
    assert np.array_equal(M, L)


def test_many_clusters_global_table_and_touched_overflow():
    """K² above the LDS budget (tables in global scratch) and more non-zero cells than the touched list holds."""
    rng = np.random.default_rng(3)
    n, m = 3000, 6
    S = np.stack([rng.integers(1, 260, n) for _ in range(m)]).astype(np.int64)      # K≈259: 4·K² > 160 KB
    S2 = np.stack([rng.integers(1, 41, n) for _ in range(m)]).astype(np.int64)      # 1600 cells, all hit: > 768 touched
    for X in (S, S2):
        for kind in (0, 2):
            M = _lib.loss_matrix(X, kind)[0]
            L = O.mpel(X, kind)[1]
            assert np.allclose(M, L, rtol=RTOL, atol=1e-13)


def test_pair_measures_and_public_functions():
    rng = np.random.default_rng(9)
    a, b = rng.integers(1, 8, 500), rng.integers(1, 5, 500)
    pm, po = _lib.pair_measures(a, b), O.pair_measures(a, b)
    for k in po:
        assert np.isclose(pm[k], po[k], rtol=RTOL, atol=1e-13), k
    assert np.isclose(rc.binderloss(a, b), po["mirkin"]) and np.isclose(rc.binderloss(a, b, normalised=False), po["mirkin"] * 500 * 499 / 2)
    assert np.isclose(rc.infodist(a, b), po["nid"]) and np.isclose(rc.infodist(a, b, normalised=False), po["id"])
    ev = rc.evaluateclustering(a, b)
    assert set(ev) == {"nbloss", "ari", "vi", "nvi", "id", "nid", "nmi"}
    assert np.isclose(ev["nvi"], po["vi"] / np.log(500)) and np.isclose(ev["nid"], po["id"] / np.log(500))
    buf = io.StringIO()
    rc.summarise(buf, a, b)
    assert "Adjusted Rand Index" in buf.getvalue() and "Number of clusters : 7" in buf.getvalue()


def test_reference_testset():
    """test/test_pointestimates.jl, line by line."""
    N, K = 100, 10
    temp = np.random.default_rng(0).integers(1, K + 1, N)
    assert abs(rc.infodist(temp, temp, normalised=True)) < 1e-9
    with pytest.raises(ValueError):
        rc.infodist(temp, np.append(temp, 1))
    assert abs(rc.infodist(temp, temp, normalised=False)) < 1e-9
    assert abs(rc.binderloss(temp, temp, normalised=True)) < 1e-9
    with pytest.raises(ValueError):
        rc.binderloss(temp, np.append(temp, 1))
    assert abs(rc.binderloss(temp, temp, normalised=False)) < 1e-9
    # a short run on paper dataset 1, as the reference's test set-up does (runtests.jl)
    d = np.load(os.path.join(HERE, "golden", "paper_datasets.npz"))
    D, truth = d["D1"], d["labels1"]
    P = rc.likelihood_hyperparams(D, truth)
    params = rc.PriorHyperparamsList(**{k: P[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta", "gamma")})
    options = rc.MCMCOptionsList(numiters=60, burnin=10, thin=1, numGibbs=1, numMH=1)
    init = np.random.default_rng(4).integers(1, 11, 100).astype(np.int64)
    result = rc.runsampler(rc.MCMCData(D), options, params, rc.MCMCState(init, 1.0, 0.5), verbose=False, seed=3)
    for kw in (dict(method="MAP"), dict(method="MLE"), dict(loss="binder", method="MPEL"), dict(loss="omARI", method="MPEL"),
               dict(loss="VI", method="MPEL"), dict(loss="ID", method="MPEL"), dict(loss=rc.varinfo, method="MPEL")):
        clust, i = rc.getpointestimate(result, **kw)
        assert 1 <= i <= len(result.clusts) and np.array_equal(clust, result.clusts[i - 1])
    # the built-in "VI" and the callable varinfo pick a minimiser of the same expected loss
    M, cs = rc.lossmatrix(result, "VI")
    assert np.isclose(cs[rc.getpointestimate(result, loss="VI", method="MPEL")[1] - 1], cs.min(), rtol=1e-12)
    assert np.isclose(cs[rc.getpointestimate(result, loss=rc.varinfo, method="MPEL")[1] - 1], cs.min(), rtol=1e-9)
    with pytest.raises(ValueError):
        rc.getpointestimate(result, loss="ID", method="some other method")
    with pytest.raises(ValueError):
        rc.getpointestimate(result, loss="some other loss function", method="MPEL")


def test_bad_inputs():
    S = np.ones((3, 10), np.int64)
    with pytest.raises(rc.RedClustHIPError):
        _lib.loss_matrix(S, 7)
    S[1, 2] = 0
    with pytest.raises(rc.RedClustHIPError):
        _lib.loss_matrix(S, 0)
    one = np.ones((4, 10), np.int64)         # a single cluster everywhere: ARI := 0, other losses 0
    assert np.all(_lib.loss_matrix(one, 0)[0] == 0) and np.all(_lib.loss_matrix(one, 2)[0] == 0)
    offdiag = ~np.eye(4, dtype=bool)
    assert np.all(_lib.loss_matrix(one, 1)[0][offdiag] == 1.0)
