"""CPU tests of the split–merge restatement (sample_labels!, /root/reference/src/mcmc.jl:356-479, and
sample_labels_Gibbs_restricted!, :259-354): C oracle vs the committed golden vectors (independent NumPy
transcription) and vs a fresh run of the transcription, proposal by proposal."""
import os

import numpy as np
import pytest

import np_transcription as T
import oracle_lib as O
from helpers import GOLD, NAMES, load_golden

MH_CASES = ["d1_merged", "d3_merged", "d3_splitup", "d2_truth"]


def mh_case(tag):
    gm = np.load(os.path.join(GOLD, "golden_mh.npz"))
    _, d = load_golden()
    ds = int(tag[1])
    D, truth = d[f"D{ds}"], d[f"labels{ds}"]
    P = T.likelihood_hyperparams(D, truth)
    return gm, D, P, gm[f"{tag}_init"].astype(np.int64), int(gm[f"{tag}_seed"])


@pytest.mark.parametrize("tag", MH_CASES)
@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_sample_labels_matches_golden(tag, mode):
    """as-written loop (numMH=3, numGibbs=5, 20 iterations): accept / split flags and the caller's labels equal the
    golden vectors — in the literal arithmetic and with stable loglik / sweep (mode 1, what the HIP path uses)."""
    gm, D, P, init, seed = mh_case(tag)
    o = O.Oracle(D, P)
    o.set_state(init)
    for it in range(20):
        na, acc, spl = o.sample_labels(1.0 + 0.05 * it, 0.5, 3, 5, seed, it, mode=mode)
        assert np.array_equal(acc, gm[f"{tag}_accept"][it]) and np.array_equal(spl, gm[f"{tag}_split"][it]), it
        assert np.array_equal(o.clusts, gm[f"{tag}_labels"][it]) and o.K == int(gm[f"{tag}_K"][it]), it
        assert na == int(np.sum(acc))


def test_quirk_q1_accepted_iteration_leaves_caller_state_untouched():
    gm, D, P, init, seed = mh_case("d1_merged")
    acc = gm["d1_merged_accept"]
    its = np.flatnonzero(acc.any(axis=1))
    assert len(its) >= 1
    labs = gm["d1_merged_labels"]
    for it in its:
        prev = gm["d1_merged_init"] if it == 0 else labs[it - 1]
        assert np.array_equal(labs[it], prev)  # neither the accepted proposal nor the Gibbs sweep reached the caller


def test_proposal_quantities_vs_transcription():
    gm, D, P, init, seed = mh_case("d3_merged")
    o = O.Oracle(D, P)
    o.set_state(init)
    logD = T.make_logD(D)
    cl = init.copy()
    sz, K = T.state_from_labels(cl)
    for it in range(6):
        for mh in range(2):
            a, s, skipped, final, info = T.mh_proposal(D, logD, cl, sz, K, P, 1.2, 0.4, 4, 77, it, mh)
            inf = o.mh_proposal(1.2, 0.4, 4, 77, it, mh, mode=0)
            assert (a, s, skipped) == (bool(inf.accept), bool(inf.split), bool(inf.skipped))
            assert (info["i"], info["j"], info["nS"]) == (inf.i, inf.j, inf.nS)
            for k in ("log_prior_ratio", "log_lik_ratio", "log_proposal_ratio"):
                x, y = info[k], getattr(inf, k)
                assert (np.isnan(x) and np.isnan(y)) or x == y or abs(x - y) <= 1e-7 * max(1.0, abs(x)), (k, x, y)
            if a:
                cl, sz, K = final
                assert np.array_equal(cl, o.clusts) and K == o.K


def test_maxk_autoreject_and_mh_uniforms():
    gm, D, P, init, seed = mh_case("d3_merged")
    P6 = dict(P, maxK=int(len(np.unique(init))))
    o = O.Oracle(D, P6)
    o.set_state(init)
    skipped = 0
    for it in range(30):
        inf = o.mh_proposal(1.0, 0.5, 2, 5, it, 0, mode=0)
        if o.clusts[inf.i] == o.clusts[inf.j]:
            assert inf.skipped == 1 and inf.accept == 0  # mcmc.jl:384-386
            skipped += 1
    assert skipped > 0
    L = O.lib()
    for args in ((1, 0, 0, 0), (99, 3, 2, 4), (2**40 + 1, 2**33, 1, 7)):
        assert L.orc_uniform_mh(*args) == T.uniform_mh(*args) and 0 < T.uniform_mh(*args) < 1
    assert L.orc_uniform_mh(1, 0, 0, 0) != L.orc_uniform(1, 0, 0, 0)  # separate stream from the sweep's
