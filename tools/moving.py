"""Sweep time vs number of label changes: perturb a fraction of the generating labels, run one sweep.
usage: python tools/moving.py N K"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, redclust_amd as rc
N, K = int(sys.argv[1]), int(sys.argv[2])
d = rc.generatemixture(N, K, seed=1); D, truth = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, truth)
ctx = rc.Context(D, kcap=max(128, 4 * K)); ctx.set_params(**P)
rng = np.random.default_rng(0)
for frac in (0.0, 0.001, 0.01, 0.05, 0.2, 1.0):
    init = truth.copy()
    m = int(frac * N)
    if m:
        idx = rng.choice(N, m, replace=False)
        init[idx] = rng.integers(1, K + 1, size=m)
    ctx.set_state(init)
    ctx.gibbs_sweep(1.0, 0.5, 7, 0)   # warm (builds S)
    ctx.set_state(init)
    t0 = time.perf_counter(); ctx.gibbs_sweep(1.0, 0.5, 7, 0); dt = time.perf_counter() - t0
    st = ctx.sweep_stats()
    print(f"perturbed {m:6d} labels: sweep {dt*1e3:8.3f} ms  changes {st['n_changes']:6d}  rounds {st['n_rounds']:5d}  K {st['K']}")
