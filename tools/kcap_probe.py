import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, redclust_amd as rc
n, K = 8192, 50
d = rc.generatemixture(n, K, seed=1); D, t = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, t)
for kcap in (128, 256, 512, 0, 2048):
    c = rc.Context(D, kcap=kcap); c.set_params(**P); c.set_state(t)
    for s in range(20): c.gibbs_sweep(1.0, 0.5, 1, s, blocking=False)
    c.synchronize()
    t0 = time.perf_counter()
    for s in range(20, 320): c.gibbs_sweep(1.0, 0.5, 1, s, blocking=False)
    c.synchronize()
    dt = time.perf_counter() - t0
    print(f"kcap={kcap}: {300/dt:.0f} sweeps/s ({dt/300*1e3:.3f} ms)")
    c.close()
