"""iterations/s of rc_run_chain with the reference's default recording (thin = 1: every iteration after burn-in is recorded)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, redclust_amd as rc
n, K = 8192, 50
sig = float(os.environ.get("SIGMA", 0.1))
d = rc.generatemixture(n, K, seed=1 if sig == 0.1 else 2, sigma=sig); D, t = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, t)
L = np.log(D + np.eye(n))
for numMH in (0, 1):
    for mode in ("full", "incremental"):
        c = rc.Context(D, kcap=128); c.set_params(**P); c.set_state(t); c.cocluster_reset(); c.set_mode(mode)
        if numMH: c.attach_host_matrices(D, L)
        c.run_chain(int(os.environ.get("BURN", 50)), 0, int(os.environ.get("THIN", 1)), 5, numMH, 1, 1.0, 0.5, 1.0)
        iters = 600
        t0 = time.perf_counter()
        ch = c.run_chain(iters, 0, int(os.environ.get("THIN", 1)), 5, numMH, 2, 1.0, 0.5, 1.0, first_iter=int(os.environ.get("BURN", 50)))
        dt = time.perf_counter() - t0
        print(f"numMH={numMH} thin={os.environ.get('THIN', 1)} sigma={sig} mode={mode}: {iters/dt:8.1f} it/s  ({dt/iters*1e3:.3f} ms/it)  samples {ch['num_samples']}")
        c.close()
