"""CPU checks of the point-estimation oracle (oracle/rc_oracle.c, pointestimate section) against the independent
NumPy transcription, scikit-learn's implementations of the same published measures, the reference's own value tests
(/root/reference/test/test_pointestimates.jl:1-8) and the committed golden vectors."""
import os

import numpy as np
import pytest

import np_transcription as T
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
KINDS = {"binder": 0, "omARI": 1, "VI": 2, "ID": 3}


def rand_pair(seed, n=300, ka=7, kb=4):
    rng = np.random.default_rng(seed)
    return rng.integers(1, ka + 1, n), rng.integers(1, kb + 1, n)


@pytest.mark.parametrize("seed", range(5))
def test_oracle_matches_transcription(seed):
    a, b = rand_pair(seed, n=100 + 37 * seed, ka=3 + seed, kb=9 - seed)
    pm = O.pair_measures(a, b)
    ari, ri, mirkin, hubert = T.randindex(a, b)
    assert np.allclose([pm["ari"], pm["ri"], pm["mirkin"], pm["hubert"]], [ari, ri, mirkin, hubert], rtol=1e-12, atol=1e-14)
    assert np.isclose(pm["mi"], T.mutualinfo(a, b, normed=False), rtol=1e-11, atol=1e-14)
    assert np.isclose(pm["nmi"], T.mutualinfo(a, b), rtol=1e-11, atol=1e-14)
    assert np.isclose(pm["vi"], T.varinfo(a, b), rtol=1e-11, atol=1e-13)
    assert np.isclose(pm["id"], T.infodist(a, b, normalised=False), rtol=1e-11, atol=1e-13)
    assert np.isclose(pm["nid"], T.infodist(a, b, normalised=True), rtol=1e-11, atol=1e-13)


def test_third_party_measures_against_sklearn():
    """Clustering.jl is not under the reference checkout; pin its published formulas with another implementation."""
    sk = pytest.importorskip("sklearn.metrics")
    for seed in range(4):
        a, b = rand_pair(10 + seed)
        pm = O.pair_measures(a, b)
        assert np.isclose(pm["ari"], sk.adjusted_rand_score(a, b), rtol=1e-10, atol=1e-13)
        assert np.isclose(pm["ri"], sk.rand_score(a, b), rtol=1e-12)
        assert np.isclose(pm["mi"], sk.mutual_info_score(a, b), rtol=1e-10, atol=1e-14)
        assert np.isclose(pm["nmi"], sk.normalized_mutual_info_score(a, b), rtol=1e-10, atol=1e-14)  # arithmetic mean norm


def test_reference_value_tests():
    """test_pointestimates.jl:2-8: distances of a labelling to itself are 0 (atol 1e-9), both normalisations."""
    temp = np.random.default_rng(0).integers(1, 11, 100)
    pm = O.pair_measures(temp, temp)
    assert abs(pm["nid"]) < 1e-9 and abs(pm["id"]) < 1e-9 and abs(pm["mirkin"]) < 1e-9
    assert abs(T.infodist(temp, temp)) < 1e-9 and abs(T.binderloss(temp, temp, normalised=False)) < 1e-9
    assert np.isclose(pm["ari"], 1.0) and np.isclose(pm["vi"], 0.0, atol=1e-12)


def test_degenerate_partitions():
    n = 50
    one = np.ones(n, np.int64)
    singles = np.arange(1, n + 1)
    pm = O.pair_measures(one, one)       # t1 == nc: ARI defined as 0
    assert pm["ari"] == 0.0 and pm["mirkin"] == 0.0 and pm["vi"] == 0.0
    pm = O.pair_measures(one, singles)
    assert np.isclose(pm["vi"], np.log(n)) and np.isclose(pm["ri"], 0.0) and np.isclose(pm["mi"], 0.0, atol=1e-15)
    assert np.isclose(pm["id"], np.log(n))


@pytest.mark.parametrize("loss", list(KINDS))
def test_mpel_oracle_vs_transcription(loss):
    rng = np.random.default_rng(5)
    base = rng.integers(1, 6, 80)
    samples = []
    for s in range(12):
        x = base.copy()
        flip = rng.random(80) < 0.05 * (1 + s % 4)
        x[flip] = rng.integers(1, 8, int(flip.sum()))
        samples.append(x)
    i_t, L_t, cs_t = T.getpointestimate_mpel(samples, loss)
    i_o, L_o, cs_o = O.mpel(np.stack(samples), KINDS[loss])
    assert np.allclose(L_o, L_t, rtol=1e-11, atol=1e-13) and np.allclose(cs_o, cs_t, rtol=1e-11)
    assert np.isclose(cs_o[i_o], cs_t[i_t], rtol=1e-12)


def test_golden_pointestimate():
    g = np.load(os.path.join(HERE, "golden", "golden_pointestimate.npz"))
    S = g["samples"]
    for loss, kind in KINDS.items():
        i, L, cs = O.mpel(S, kind)
        assert np.allclose(L, g[f"lossmatrix_{loss}"], rtol=1e-11, atol=1e-13)
        # repeated samples tie exactly in exact arithmetic, so the index among ties is decided by rounding (in the
        # reference too): the chosen sample must be one of the minimisers and the same partition
        gi = int(g[f"argmin_{loss}"])
        assert np.isclose(cs[i], g[f"colsum_{loss}"][gi], rtol=1e-12)
        assert O.pair_measures(S[i], S[gi])["mirkin"] == 0.0
    ev = T.evaluateclustering(S[-1], g["truth"])
    pm = O.pair_measures(S[-1], g["truth"])
    for k, ok in (("nbloss", "mirkin"), ("ari", "ari"), ("vi", "vi"), ("id", "id"), ("nmi", "nmi")):
        assert np.isclose(pm[ok], float(g[f"eval_{k}"]), rtol=1e-11, atol=1e-13) and np.isclose(ev[k], float(g[f"eval_{k}"]))
