"""iterations/s of the full iteration (sample_r, sample_p, numMH split-merge proposals, Gibbs sweep) with the reference's
default options (numMH = 1, numGibbs = 5) at the headline size; RC_SM_PROFILE=1 adds the phase profile of rc_splitmerge"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, redclust_amd as rc
n, K = int(os.environ.get("N", 8192)), int(os.environ.get("K", 50))
sig = float(os.environ.get("SIGMA", 0.1))
d = rc.generatemixture(n, K, seed=1 if sig == 0.1 else 2, sigma=sig); D, t = d["distancematrix"], d["clusts"]
P = rc.likelihood_hyperparams(D, t)
L = np.log(D + np.eye(n))
for numMH in (0, 1):
    c = rc.Context(D, kcap=max(128, 2 * K) if sig == 0.1 else 512); c.set_params(**P); c.set_state(t); c.cocluster_reset()
    if numMH: c.attach_host_matrices(D, L)
    if os.environ.get("MODE"): c.set_mode(os.environ["MODE"])
    c.run_chain(int(os.environ.get("BURN", 50)), 0, 10, 5, numMH, 1, 1.0, 0.5, 1.0)
    iters = 1000
    t0 = time.perf_counter()
    ch = c.run_chain(iters, 0, 10, 5, numMH, 2, 1.0, 0.5, 1.0, first_iter=int(os.environ.get("BURN", 50)))
    dt = time.perf_counter() - t0
    print(f"numMH={numMH} numGibbs=5: {iters/dt:8.1f} it/s  ({dt/iters*1e3:.3f} ms/it)  split-merge acceptances {int(ch['splitmerge_acceptances'].sum())} K_final {ch['K'][-1]}")
    c.close()
