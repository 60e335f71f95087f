import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, redclust_amd as rc
def soak(n, K, sigma, dim, sweeps, every, kcap, maxK=0):
    data = rc.generatemixture(n, K, seed=11, sigma=sigma, dim=dim)
    sh = np.random.default_rng(2).permutation(n)
    D = np.ascontiguousarray(data["distancematrix"][np.ix_(sh, sh)]); truth = data["clusts"][sh]
    P = dict(rc.likelihood_hyperparams(D, truth), maxK=maxK)
    A = rc.Context(D, kcap=kcap); A.set_params(**P); A.set_state(truth)
    B = rc.Context(D, kcap=kcap); B.set_params(**P); B.set_bulk_kernel("perm"); B.set_state(truth)
    t0 = time.perf_counter(); moved = 0
    for t in range(sweeps):
        A.gibbs_sweep(1.0, 0.5, 21, t, blocking=False); B.gibbs_sweep(1.0, 0.5, 21, t, blocking=False)
        if t % every == every - 1:
            a, b = A.get_state(), B.get_state()
            assert np.array_equal(a[0], b[0]) and a[2] == b[2], ("diverged", n, t)
            assert np.array_equal(a[1], np.bincount(a[0], minlength=n + 1)[1:]) and np.array_equal(b[1], np.bincount(b[0], minlength=n + 1)[1:]), ("inconsistent", n, t)
            moved += A.sweep_stats()["n_changes"]
    print(f"soak n={n}: {sweeps} sweeps ok, sampled changes {moved}, K={A.sweep_stats()['K']}, layouts {A.layout_info()}, {time.perf_counter()-t0:.1f} s")
    A.close(); B.close()
soak(8192, 50, 0.17, 50, 6000, 10, 256)
soak(5000, 12, 0.5, 12, 6000, 7, 128, maxK=40)
soak(700, 5, 0.6, 6, 20000, 11, 64, maxK=15)
