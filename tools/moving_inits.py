"""first sweeps from perturbed / random initial labels at the headline data: time, changes, resolver rounds per sweep"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, redclust_amd as rc
n, K = 8192, 50
d = rc.generatemixture(n, K, seed=1); D, t = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, t)
c = rc.Context(D, kcap=int(os.environ.get("KCAP", 256))); c.set_params(**P)
rng = np.random.default_rng(5)
for name, frac in (("1% re-drawn", 0.01), ("20% re-drawn", 0.2), ("uniform random", 1.0)):
    lab = t.copy(); idx = rng.choice(n, int(frac * n), replace=False); lab[idx] = rng.integers(1, K + 1, len(idx))
    c.set_state(lab); c.synchronize()
    for s in range(4):
        t0 = time.perf_counter(); c.gibbs_sweep(1.0, 0.5, 3, s); dt = time.perf_counter() - t0
        st = c.sweep_stats()
        print(f"{name:16s} sweep {s}: {dt*1e3:8.3f} ms  changes {st['n_changes']:5d} rounds {st['n_rounds']:4d} K {st['K']}")
