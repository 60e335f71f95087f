"""Generates the committed golden fixtures (run in the build container; the reference is read-only there).

  python tests/golden/make_golden.py

1. paper_datasets.npz — the three N=100, K=10 paper datasets of the reference's
   data/example_datasets.h5 (provenance: /root/reference/src/example_data.jl:40-50,56-71): distance
   matrices and generating labels, read straight from the file's contiguous little-endian datasets
   (HDF5 superblock v0; byte offsets verified symmetric / zero-diagonal).  DATA only.
2. golden_sweeps.npz — expected outputs of the independent NumPy/SciPy transcription
   (tests/np_transcription.py) on those inputs: labels / sizes / K after each of S teacher-forced
   sweeps, loglik, logprior, canonical labels, per-point candidate log-weights.

The Julia package itself cannot be run here (no julia binary), so these vectors pin the
build's reading of the reference, not the package's output ("parity unpinned" in DESIGN.md).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import np_transcription as T  # noqa: E402

H5 = "/root/reference/data/example_datasets.h5"
OFFSETS = [(12096, 92096), (215024, 295024), (385952, 465952)]  # (distance_matrix, cluster_labels)
S = 4  # sweeps per case


def extract():
    b = open(H5, "rb").read()
    out = {}
    for d, (od, ol) in enumerate(OFFSETS, start=1):
        D = np.frombuffer(b, dtype="<f8", count=100 * 100, offset=od).reshape(100, 100).copy()
        L = np.frombuffer(b, dtype="<i8", count=100, offset=ol).copy()
        assert np.array_equal(D, D.T) and np.all(np.diag(D) == 0) and L.min() == 1 and L.max() == 10
        out[f"D{d}"] = D
        out[f"labels{d}"] = L
    return out


def run_case(D, init, P, seed, rs, ps):
    logD = T.make_logD(D)
    clusts = init.copy()
    sizes, K = T.state_from_labels(clusts)
    rec = dict(labels=[], sizes=[], K=[], loglik=[], logprior=[], canon=[])
    # candidate log-weights of points 0, 37, 99 in the initial state
    sc = {}
    for i in (0, 37, 99):
        cands, lp = T.point_scores(D, logD, clusts, sizes, P, rs[0], ps[0], i)
        sc[i] = (cands, lp)
    for t in range(S):
        K = T.sweep(D, logD, clusts, sizes, P, rs[t], ps[t], seed, t)
        rec["labels"].append(clusts.copy())
        rec["sizes"].append(sizes.copy())
        rec["K"].append(K)
        rec["loglik"].append(T.loglik(D, logD, clusts, sizes, P))
        rec["logprior"].append(T.logprior(sizes, rs[t], ps[t], P))
        rec["canon"].append(T.sortlabels(clusts))
    return rec, sc


def main():
    data = extract()
    np.savez_compressed(os.path.join(HERE, "paper_datasets.npz"), **data)
    gold = {}
    rs = np.array([1.0, 1.3, 0.8, 2.1])
    ps = np.array([0.5, 0.42, 0.61, 0.3])
    gold["r_seq"], gold["p_seq"] = rs, ps
    cases = []
    for d in (1, 2, 3):
        D, truth = data[f"D{d}"], data[f"labels{d}"]
        P = T.likelihood_hyperparams(D, truth)
        rng = np.random.default_rng(100 + d)
        rand_init = rng.integers(1, 11, size=100).astype(np.int64)
        singles = np.arange(1, 101, dtype=np.int64)
        variants = [("truth", truth, P), ("random", rand_init, P)]
        if d == 1:
            variants += [("norep", rand_init, dict(P, repulsion=False)),
                         ("maxK6", rand_init, dict(P, maxK=6)),
                         ("singletons", singles, P)]
        for name, init, PP in variants:
            tag = f"d{d}_{name}"
            rec, sc = run_case(D, init, PP, seed=1234 + d, rs=rs, ps=ps)
            cases.append(tag)
            gold[f"{tag}_init"] = init
            gold[f"{tag}_params"] = np.array([PP[k] for k in ("delta1", "delta2", "alpha", "beta", "zeta",
                                                               "gamma", "eta", "sigma", "u", "v")])
            gold[f"{tag}_repulsion"] = np.array(int(PP["repulsion"]))
            gold[f"{tag}_maxK"] = np.array(int(PP["maxK"]))
            gold[f"{tag}_seed"] = np.array(1234 + d)
            for k, v in rec.items():
                gold[f"{tag}_{k}"] = np.array(v)
            for i, (cands, lp) in sc.items():
                gold[f"{tag}_cands_pt{i}"] = cands
                gold[f"{tag}_logprobs_pt{i}"] = lp
            print(tag, "K:", rec["K"], "loglik:", [round(x, 3) for x in rec["loglik"]])
    gold["cases"] = np.array(cases)
    # a few uniforms of the counter-based stream (known-answer for the uniform source)
    gold["uniform_kat"] = np.array([T.uniform(1, 0, 0, 0), T.uniform(1235, 3, 99, 10),
                                    T.uniform(2**40 + 7, 2**33 + 1, 8191, 50)])
    np.savez_compressed(os.path.join(HERE, "golden_sweeps.npz"), **gold)
    make_golden_mh(data)
    make_golden_pointestimate(data)
    make_golden_chain(data)
    print("wrote", HERE)


def make_golden_mh(data):
    """Split–merge step as written (sample_labels!, mcmc.jl:356-479): accept / split flags and the caller's labels
    after each of 20 iterations (numMH=3, numGibbs=5) from inits with merged and with split true clusters."""
    gold = {}
    cases = []
    for d, kind in ((1, "merged"), (3, "merged"), (3, "splitup"), (2, "truth")):
        D, truth = data[f"D{d}"], data[f"labels{d}"]
        P = T.likelihood_hyperparams(D, truth)
        init = truth.copy()
        if kind == "merged":
            init[init == 2] = 1; init[init == 4] = 3; init[init == 9] = 8
        elif kind == "splitup":
            m = np.flatnonzero(init == 5); init[m[: len(m) // 2]] = 11
            m = np.flatnonzero(init == 7); init[m[::2]] = 12
        logD = T.make_logD(D)
        cl = init.copy()
        sz, K = T.state_from_labels(cl)
        acc, spl, labs, Ks = [], [], [], []
        for it in range(20):
            a, s, K = T.sample_labels(D, logD, cl, sz, K, P, 1.0 + 0.05 * it, 0.5, 3, 5, 4321 + d, it)
            acc.append(a); spl.append(s); labs.append(cl.copy()); Ks.append(K)
        tag = f"d{d}_{kind}"
        cases.append(tag)
        gold[f"{tag}_init"] = init
        gold[f"{tag}_accept"] = np.array(acc)
        gold[f"{tag}_split"] = np.array(spl)
        gold[f"{tag}_labels"] = np.array(labs)
        gold[f"{tag}_K"] = np.array(Ks)
        gold[f"{tag}_seed"] = np.array(4321 + d)
        print(tag, "accepted", int(np.sum(acc)), "splits proposed", int(np.sum(spl)), "K", Ks[0], "->", Ks[-1])
    gold["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "golden_mh.npz"), **gold)


def make_golden_pointestimate(data):
    """MPEL loss matrices (pointestimate.jl:49-58) and evaluateclustering (summaries.jl:12-23) of the transcription on
    30 samples of a teacher-forced chain on paper dataset 1 (random init, so early samples differ a lot)."""
    D, truth = data["D1"], data["labels1"]
    P = T.likelihood_hyperparams(D, truth)
    logD = T.make_logD(D)
    rng = np.random.default_rng(2024)
    cl = rng.integers(1, 11, size=100).astype(np.int64)
    sz, K = T.state_from_labels(cl)
    samples = []
    for t in range(30):
        T.sweep(D, logD, cl, sz, P, 1.0, 0.5, 77, t)
        samples.append(cl.copy())
    gold = {"samples": np.array(samples), "truth": truth}
    for loss in ("binder", "omARI", "VI", "ID"):
        i, L, cs = T.getpointestimate_mpel(samples, loss)
        gold[f"lossmatrix_{loss}"] = L
        gold[f"colsum_{loss}"] = cs
        gold[f"argmin_{loss}"] = np.array(i)
        print("pointestimate", loss, "argmin", i, "min expected loss", cs[i] / len(samples))
    for k, v in T.evaluateclustering(samples[-1], truth).items():
        gold[f"eval_{k}"] = np.array(v)
    np.savez_compressed(os.path.join(HERE, "golden_pointestimate.npz"), **gold)


def make_golden_chain(data):
    """Free-running chains of the transcription's runsampler loop (mcmc.jl:533-556; scalar updates on the build's
    scalar stream): paper dataset 1 with numMH = 0 and dataset 1 (three merged pairs of true clusters as init) with
    numMH = 2, numGibbs = 5 (seed chosen so that proposals of both kinds are accepted)."""
    gold = {}
    for tag, d, numMH, iters, seed in (("d1_gibbs", 1, 0, 40, 100), ("d1_mh", 1, 2, 25, 103)):
        D, truth = data[f"D{d}"], data[f"labels{d}"]
        P = T.likelihood_hyperparams(D, truth)
        init = truth.copy()
        if numMH:
            init[init == 2] = 1; init[init == 4] = 3; init[init == 9] = 8
        else:
            init = np.random.default_rng(9).integers(1, 11, size=100).astype(np.int64)
        rec = T.run_chain(D, init, P, 1.0, 0.5, iters, 5, 2, 5, numMH, seed, eta=P["eta"], sigma=P["sigma"],
                          proposalsd_r=0.7, u=P["u"], v=P["v"])
        gold[f"{tag}_init"] = init
        gold[f"{tag}_seed"] = np.array(seed)
        gold[f"{tag}_numMH"] = np.array(numMH)
        gold[f"{tag}_iters"] = np.array(iters)
        for k, v in rec.items():
            gold[f"{tag}_{k}"] = np.array(v)
        print(tag, "K", rec["K"], "r accepted", int(np.sum(rec["r_acc"])), "sm accepted", int(np.sum(rec["sm_acc"])) if numMH else "-")
    np.savez_compressed(os.path.join(HERE, "golden_chain.npz"), **gold)


if __name__ == "__main__":
    main()
