"""Plain sweep timing with the oldest API surface (works with experiment builds of older sources).
usage: python tools/time_sweeps.py n K bits steps"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import redclust_amd as rc
n, K, bits, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
d = rc.generatemixture(n, K, seed=1)
D, truth = d["distancematrix"], d["clusts"]
if os.environ.get("SHUFFLE"):
    sh = np.random.default_rng(3).permutation(n)
    D = np.ascontiguousarray(D[np.ix_(sh, sh)]); truth = truth[sh]
P = rc.likelihood_hyperparams(D, truth) if n <= 16384 else dict(delta1=20.0, delta2=30.0, alpha=1e6, beta=1e5, zeta=1e9, gamma=1e9, eta=1.0, sigma=1.0, u=1.0, v=1.0, repulsion=True, maxK=0)
L = np.log(np.where(np.eye(n, dtype=bool), 1.0, D)) if os.environ.get("STORED") else None   # STORED=1: the caller's logD (the Julia glue's exact_logD)
ctx = rc.Context(D, logD=L, storage_bits=bits); ctx.set_params(**P); ctx.set_state(truth)
for t in range(10): ctx.gibbs_sweep(1.0, 0.5, 1, t, blocking=False)
ctx.synchronize()
t0 = time.perf_counter()
for t in range(10, 10 + steps): ctx.gibbs_sweep(1.0, 0.5, 1, t, blocking=False)
t_enq = time.perf_counter() - t0
ctx.synchronize()
dt = time.perf_counter() - t0
print(f"n={n} K={K} bits={bits} {ctx.bulk_kernel_name()}: {steps / dt:.1f} sweeps/s ({dt / steps * 1e3:.3f} ms/sweep; host enqueue {t_enq / steps * 1e6:.1f} us/sweep) changes {ctx.sweep_stats()}")
